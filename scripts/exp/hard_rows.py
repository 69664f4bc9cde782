"""Development helper (checker run: uses the oracles): one scene of the s = HARD sweep on the device, per-key relative L2 error of the kernel and of the fp32
oracle against the fp64 oracle, the three worst rows of each, under several kernel configurations (few-tile segments, plain kernels).
usage: python scripts/exp/hard_rows.py 180333 1.2"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.gs_oracle import Oracle
from activesplat_amd import _lib
from tests import util
from tests.fuzz_scenes import sweep_scene
seed, hard = int(sys.argv[1]), float(sys.argv[2])
lib = _lib.get()
o32, o64 = Oracle("f32"), Oracle("f64"); o32.set_threads(16); o64.set_threads(16)
rs, rv = sweep_scene(seed, "cuda", None)
gen = torch.Generator().manual_seed(seed)
rv["scales"] = rv["scales"] * torch.exp(hard * torch.randn(rv["scales"].shape, generator=gen)).to(rv["scales"].device)
if seed % 2:
    rv["means3D"] = rv["means3D"] * torch.tensor([1.0, 1.0, 0.35], device=rv["means3D"].device)
H, W = int(rs.image_height), int(rs.image_width)
dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(0))
r64 = util.run_oracle(o64, rs, rv, dL); r32 = util.run_oracle(o32, rs, rv, dL)
print("seed", seed, "P", rv["means3D"].shape[0], f"{W}x{H}", "D", r32["D"])
def report(tag):
    got = util.run_product(rs, rv, dL)
    out = []
    for k, g in got["grads"].items():
        r = np.asarray(r64["grads"][k], np.float64).reshape(g.shape); o = np.asarray(r32["grads"][k], np.float64).reshape(g.shape)
        n = np.linalg.norm(r)
        if n == 0: continue
        ek = ((g - r) ** 2).reshape(g.shape[0], -1).sum(1); eo = ((o - r) ** 2).reshape(g.shape[0], -1).sum(1)
        out.append(f"{k} k {np.sqrt(ek.sum()) / n:.2e} (rows {np.argsort(ek)[-2:][::-1].tolist()} {np.round(np.sort(ek)[-2:][::-1] / ek.sum(), 2).tolist()}) o32 {np.sqrt(eo.sum()) / n:.2e}")
    print(tag + ": " + " | ".join(out), flush=True)
    return got
for rep in range(2):
    g = report("default")
_lib.check(lib.gs_set_backward_segments(1)); report("bwd segments 1"); _lib.check(lib.gs_set_backward_segments(3))
_lib.check(lib.gs_set_half_quadrants(0)); report("plain kernels"); _lib.check(lib.gs_set_backward_chain(3, 0)); report("plain + chained"); 
_lib.check(lib.gs_set_half_quadrants(256)); _lib.check(lib.gs_set_backward_chain(3, -1))
i = int(os.environ.get("ROW", -1))
if i >= 0:
    print("row", i, "radius", r64["radii"][i], "xy", r64["xy"][i], "conic", r64["conic_opacity"][i], "cov2d", r64["cov2d"][i], "depth", r64["depth"][i], "tiles", r64["tiles_touched"][i])
    for k in ("means2D", "means3D", "scales"):
        print("  ", k, "fp64", np.asarray(r64["grads"][k])[i], "kernel", g["grads"][k][i], "o32", np.asarray(r32["grads"][k])[i])
