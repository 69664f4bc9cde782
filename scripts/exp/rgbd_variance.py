"""Development helper: run-to-run spread of the fused RGB-D gradients (atomics order) next to the fused-vs-two-pass difference, per tensor, and where it sits.
GPU box: SEED=150586 python scripts/exp/rgbd_variance.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import rasterizer as R
from tests.fuzz_scenes import sweep_scene

seed = int(os.environ.get("SEED", 150586))
rs, rv = sweep_scene(seed, "cuda", os.environ.get("PLAIN"))
H, W = int(rs.image_height), int(rs.image_width); P = rv["means3D"].shape[0]
g = torch.Generator().manual_seed(seed)
dLc = torch.randn(3, H, W, generator=g).cuda(); dLd = torch.randn(1, H, W, generator=g).cuda()
print("seed", seed, "P", P, f"{W}x{H}")


def fused():
    inp = {k: v.detach().clone().requires_grad_(True) for k, v in rv.items()}
    m2d = torch.zeros(P, 3, device="cuda", requires_grad=True)
    color, radii, depth, sil, dsq = R.render_rgbd(rs, means2D=m2d, **inp)
    ((color * dLc).sum() + (depth * dLd).sum()).backward()
    return {k: v.grad.double() for k, v in inp.items()}


def two():
    inp = {k: v.detach().clone().requires_grad_(True) for k, v in rv.items()}
    m2a = torch.zeros(P, 3, device="cuda", requires_grad=True); m2b = torch.zeros(P, 3, device="cuda", requires_grad=True)
    c1 = R.GaussianRasterizer(raster_settings=rs)(means2D=m2a, **inp)[0]
    V = rs.viewmatrix.reshape(4, 4).cuda()
    z = inp["means3D"] @ V[:3, 2] + V[3, 2]
    second = {k: v for k, v in inp.items() if k not in ("colors_precomp", "shs")}
    c2 = R.GaussianRasterizer(raster_settings=rs._replace(bg=torch.zeros(3, device="cuda")))(means2D=m2b, colors_precomp=torch.stack([z, torch.ones_like(z), z * z], 1), **second)[0]
    ((c1 * dLc).sum() + (c2[0:1] * dLd).sum()).backward()
    return {k: v.grad.double() for k, v in inp.items()}


F = [fused() for _ in range(6)]
T = [two() for _ in range(3)]
for k in F[0]:
    nrm = float(T[0][k].norm())
    spread_f = max(float((F[i][k] - F[0][k]).norm()) for i in range(1, 6)) / nrm
    spread_t = max(float((T[i][k] - T[0][k]).norm()) for i in range(1, 3)) / nrm
    d = (F[0][k] - T[0][k]).reshape(P, -1)
    e = (d ** 2).sum(1)
    top = torch.argsort(-e)[:3].tolist()
    print(f"{k:15s} fused-vs-two {float(d.norm()) / nrm:.2e}   run-to-run fused {spread_f:.2e}  two-pass {spread_t:.2e}   share of top Gaussian {float(e[top[0]] / e.sum()):.3f}  top {top}")
