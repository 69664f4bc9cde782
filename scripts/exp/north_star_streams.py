"""2M-Gaussian SH-3 frames on 1..4 streams (GPU box)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import GaussianRasterizer, setup_camera, synthetic as syn, rasterizer as R  # noqa: E402
dev = torch.device("cuda"); W, H, N = 640, 480, 2_000_000
deg = int(os.environ.get("SH", 3))
p = syn.make_params(N, W, H, seed=0, sh_degree=deg if deg else None)
rv = {k: v.to(dev).requires_grad_(True) for k, v in syn.activate(p).items()}
cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev, sh_degree=deg)
dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
keys = list(rv.keys())
def step():
    m2 = torch.zeros(N, 3, device=dev, requires_grad=True)
    col = GaussianRasterizer(raster_settings=cam)(means2D=m2, **rv)[0]
    return torch.autograd.grad(col, [rv[k] for k in keys] + [m2], dL)
for ns in (1, 2, 3, 4):
    pool = [torch.cuda.Stream() for _ in range(ns)]
    torch.cuda.synchronize()
    for i in range(6):
        with torch.cuda.stream(pool[i % ns]): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(36):
        with torch.cuda.stream(pool[i % ns]): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 36
    fb = N * (832 if deg else 292) + R.last_stats["num_rendered"] * 160 + W * H * 48
    print(f"SH-{deg} streams={ns}: {1/dt:.1f} frames/s  {fb/dt/8e12*100:.1f} % of 8 TB/s")
