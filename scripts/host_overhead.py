"""Development helper: host-side cost of one forward+backward (tiny scene, GPU work negligible)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activesplat_amd import GaussianRasterizer, setup_camera  # noqa: E402
from activesplat_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda")
N, W, H = 2000, 120, 150
cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev)
rv = {k: v.to(dev).requires_grad_(True) for k, v in syn.activate(syn.make_params(N, W, H, seed=0)).items()}
dL = torch.randn(3, H, W, device=dev)


def fwd_only():
    with torch.no_grad():
        m2d = torch.zeros(N, 3, device=dev)
        GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)


def step():
    m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
    color, _, _, _ = GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)
    torch.autograd.grad(color, list(rv.values()) + [m2d], dL)


for f, name in ((fwd_only, "forward only (no_grad)"), (step, "forward+backward")):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        f()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per call (N={N}, {W}x{H})")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
