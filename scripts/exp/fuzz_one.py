"""Development helper (checker run): one seed of scripts/exp/fuzz_gpu.py in detail -- where the product's gradient differs from the
fp64 oracle, next to the fp32 oracle's own difference.  usage: SEED=50310 python scripts/exp/fuzz_one.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.gs_oracle import Oracle
from tests import util
from tests.fuzz_scenes import sweep_scene

seed = int(os.environ.get("SEED", 50310))
dev = os.environ.get("DEVICE", "cuda")
if dev == "cpu":                                            # the host-emulated kernels (tests/hipemu), for comparison
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "hipemu"), "-j8"], stdout=subprocess.DEVNULL)
    util.use_emulated_kernels(os.path.join(ROOT, "tests", "hipemu", "libgsplat_emu.so"))
o32, o64 = Oracle("f32"), Oracle("f64")
rs, rv = sweep_scene(seed, dev, os.environ.get("PLAIN"))
if os.environ.get("SEGS"):                                  # list segments of the few-tile backward (gs_set_backward_segments)
    from activesplat_amd import _lib
    _lib.get().gs_set_backward_segments(int(os.environ["SEGS"]))
H, W = int(rs.image_height), int(rs.image_width)
dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(0))
got = util.run_product(rs, rv, dL)
art = util.artefacts()
r64 = util.run_oracle(o64, rs, rv, dL); r32 = util.run_oracle(o32, rs, rv, dL)
print("P", rv["means3D"].shape[0], "WxH", W, H, "D", got["D"])
print("n_contrib equal to fp32 oracle:", np.array_equal(art["n_contrib"], r32["n_contrib"].reshape(H, W)) if "n_contrib" in r32 else "n/a")
for k, g in got["grads"].items():
    a = r64["grads"][k].reshape(g.shape); b = r32["grads"][k].reshape(g.shape).astype(np.float64)
    d = np.abs(g - a).reshape(g.shape[0], -1).max(1); d32 = np.abs(b - a).reshape(g.shape[0], -1).max(1)
    top = np.argsort(-d)[:4]
    print(k, "rel", np.linalg.norm(g - a) / np.linalg.norm(a), "rel32", np.linalg.norm(b - a) / np.linalg.norm(a), "|g|max", np.abs(a).max())
    for i in top:
        print("   gaussian", i, "err", d[i], "fp32-oracle err", d32[i], "radius", got["radii"][i])
    # the elements outside the stated tolerance (rtol 1e-3, atol 1e-6 |g|inf): how large are they, and how much cancellation is behind them?
    gmax = float(np.abs(a).max())
    out = np.abs(g - a) > 1e-6 * gmax + 1e-3 * np.abs(a)
    out32 = np.abs(b - a) > 1e-6 * gmax + 1e-3 * np.abs(a)
    print("   elements outside the tolerance: kernel %d, fp32 oracle %d of %d" % (int(out.sum()), int(out32.sum()), out.size))
    idx = np.argwhere(out)[:8]
    for ix in idx:
        ix = tuple(ix)
        print("      element", ix, "fp64 %.3e kernel %.3e fp32-oracle %.3e  |value| / |g|inf = %.1e  kernel err / |g|inf = %.1e" % (
            a[ix], g[ix], b[ix], abs(a[ix]) / gmax, abs(g[ix] - a[ix]) / gmax), "scales", rv["scales"][ix[0]].tolist() if "scales" in rv else None, "radius", got["radii"][ix[0]])
