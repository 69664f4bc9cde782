"""Development helper: where the host time of one forward + backward through the Python boundary goes (tiny scene: the GPU work is
negligible).  GPU box: python scripts/exp/host_profile.py"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import GaussianRasterizer, setup_camera
from activesplat_amd import synthetic as syn

dev = torch.device("cuda")
N, W, H = int(os.environ.get("N", 1000)), int(os.environ.get("W", 64)), int(os.environ.get("H", 64))
cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev)
rv = {k: v.to(dev).requires_grad_(True) for k, v in syn.activate(syn.make_params(N, W, H, seed=0)).items()}
dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
m2d = torch.zeros(N, 3, device=dev, requires_grad=True)


def step():
    color = GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)[0]
    torch.autograd.grad(color, list(rv.values()) + [m2d], dL)


for _ in range(50):
    step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(500):
    step()
torch.cuda.synchronize(); print("%.1f us per frame" % ((time.perf_counter() - t) / 500 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(500):
    step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
