"""Development helper: random sweep of the mapper's loss call -- get_loss as the reference runs it against the fully fused form of the same call
(tests/parity_cases.check_get_loss_random_draw on many seeds; the suite runs a handful, tests/test_gpu_parity.py).  profiles/r05_fuzz_get_loss.txt.
GPU box: SEED0=0 SEED1=300 python scripts/exp/fuzz_get_loss.py     (SEEDS=135,444: those draws; DEVICE=cpu: the host-emulated kernels)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from activesplat_amd import rasterizer as R
from tests import parity_cases as pc

dev = os.environ.get("DEVICE", "cuda")
if dev == "cpu":
    import subprocess
    from tests import util
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "hipemu"), "-j8"], stdout=subprocess.DEVNULL)
    util.use_emulated_kernels(os.path.join(ROOT, "tests", "hipemu", "libgsplat_emu.so"))
    R.use_frontend = False
bad, rows, events = [], [], []
seeds = [int(v) for v in os.environ["SEEDS"].split(",")] if os.environ.get("SEEDS") else range(int(os.environ.get("SEED0", 0)), int(os.environ.get("SEED1", 300)))
for seed in seeds:
    try:
        v = pc.check_get_loss_random_draw(seed, dev)
        if v != "ok":
            (rows if v[0] == "rows" else events).append((seed,) + tuple(v))
    except Exception as e:
        bad.append(seed)
        print("FAIL seed", seed, repr(e)[:300], flush=True)
print("get_loss sweep (reference call pattern against the fully fused call): seeds %s..%s, %d failures %s" % (seeds[0], seeds[-1] + 1, len(bad), bad[:30]))
print("   scenes whose gradients differ because the two renders differ by a discrete event (seed, class, first key above 3e-4, its relative difference, detail):")
for ev in events:
    print("     ", ev)
print("   rows above 3e-3 of the norm while the rest agrees to 3e-4 (seed, 'rows', key, row, relative difference):", rows)
