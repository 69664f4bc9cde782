"""Development helper (checker run): a forward that misses the image tolerance against the fp32 oracle on a HARD sweep scene (strongly anisotropic splats) --
is the kernel further from the truth (fp64 oracle) than the fp32 oracle is, or are both simply fp32 evaluations of an ill-conditioned quadratic form?
GPU box: HARD=1.2 SEED=180297 python scripts/exp/fwd_hard.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.gs_oracle import Oracle
from tests import parity_cases as pc, util
from tests.fuzz_scenes import sweep_scene
seed = int(os.environ["SEED"])
rs, rv = sweep_scene(seed, "cuda", os.environ.get("PLAIN"))
if os.environ.get("HARD") and "scales" in rv:
    gen = torch.Generator().manual_seed(seed)
    rv["scales"] = rv["scales"] * torch.exp(float(os.environ["HARD"]) * torch.randn(rv["scales"].shape, generator=gen)).to(rv["scales"].device)
    if seed % 2:
        rv["means3D"] = rv["means3D"] * torch.tensor([1.0, 1.0, 0.35], device=rv["means3D"].device)
o32, o64 = Oracle("f32"), Oracle("f64")
got = util.run_product(rs, rv); art = util.artefacts()
r32 = util.run_oracle(o32, rs, rv); r64 = util.run_oracle(o64, rs, rv)
print("seed", seed, "P", rv["means3D"].shape[0], "image", got["color"].shape, "D", got["D"])
for name, kr in (("color", "color"), ("depth", "out_depth"), ("opacity", "opacity")):
    a, b, c = got[name].astype(np.float64), r32[kr].astype(np.float64), r64[kr]
    sc = max(1.0, float(np.abs(c).max()))
    tol = lambda x, y: np.abs(x - y) > pc.FWD_ATOL * sc + pc.FWD_RTOL * np.abs(y)
    print(f"{name:8s} values outside the tolerance: kernel vs fp32 oracle {int(tol(a, b).sum())}, kernel vs fp64 {int(tol(a, c).sum())}, fp32 oracle vs fp64 {int(tol(b, c).sum())} of {a.size};"
          f"  max |.|: {np.abs(a - b).max():.2e} / {np.abs(a - c).max():.2e} / {np.abs(b - c).max():.2e}")
nc = art["n_contrib"]
print("n_contrib: kernel vs fp32 oracle differ at", int((nc != r32["n_contrib"].reshape(nc.shape)).sum()), "pixels; fp32 oracle vs fp64 at", int((r32["n_contrib"] != r64["n_contrib"]).sum()))
sc_ = rv["scales"].cpu().numpy(); an = sc_.max(1) / sc_.min(1)
print("anisotropy (max / min scale): median %.1f, p90 %.1f, max %.1f" % (np.median(an), np.percentile(an, 90), an.max()))
