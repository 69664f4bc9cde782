"""Development helper: random sweep of the RAW-PARAMETER rasteriser entry (render_rgbd_raw) against the activation kernels in front of the plain fused render
(fused_rendervar + render_rgbd) -- tests/parity_cases.check_raw_entry_random_draw on many seeds; the suite runs a handful (tests/test_gpu_parity.py).
GPU box: SEED0=0 SEED1=300 python scripts/exp/fuzz_raw.py     (ROWS=1: detail of the flagged gradient rows; DEVICE=cpu: the host-emulated kernels)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from activesplat_amd import rasterizer as R
from tests import parity_cases as pc

dev = os.environ.get("DEVICE", "cuda")
if dev == "cpu":                                            # the host-emulated kernels (tests/hipemu): a flagged seed traced without a GPU
    import subprocess
    from tests import util
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "hipemu"), "-j8"], stdout=subprocess.DEVNULL)
    util.use_emulated_kernels(os.path.join(ROOT, "tests", "hipemu", "libgsplat_emu.so"))
    R.use_frontend = False
bad, ties, rows = [], [], []
for seed in range(int(os.environ.get("SEED0", 0)), int(os.environ.get("SEED1", 300))):
    try:
        v = pc.check_raw_entry_random_draw(seed, dev, rows_detail=bool(os.environ.get("ROWS")))
        if v == "depth tie":
            ties.append(seed)
        elif v != "ok":
            rows.append((seed,) + v[1:])
    except Exception as e:
        bad.append(seed)
        print("FAIL seed", seed, repr(e)[:260], flush=True)
print("raw-parameter sweep: seeds %s..%s, %d failures %s" % (os.environ.get("SEED0", 0), os.environ.get("SEED1", 300), len(bad), bad[:20]))
print("   %d scenes where the two entries order a depth tie (view depths within four fp32 ulps) differently and every differing pixel lies in the pair's footprints: %s" % (len(ties), ties))
print("   %d scenes where two rows carry a gradient difference above 3e-3 of the norm while the rest agrees to 3e-4 (seed, key, row, relative difference): %s" % (len(rows), rows))
