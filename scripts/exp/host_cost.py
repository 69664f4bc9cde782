"""Development helper: host cost of one forward+backward through the Python boundary (tiny scene: the GPU work is negligible),
and the configs[1] frame rate with the frames issued by 1..4 streams from one host thread / from one host thread per stream."""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import GaussianRasterizer, setup_camera
from activesplat_amd import synthetic as syn

dev = torch.device("cuda")

def make(N, W, H):
    cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev)
    rv = {k: v.to(dev).requires_grad_(True) for k, v in syn.activate(syn.make_params(N, W, H, seed=0)).items()}
    dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    def step():
        m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
        color = GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)[0]
        torch.autograd.grad(color, list(rv.values()) + [m2d], dL)
    return step

step = make(1000, 64, 64)
for _ in range(20): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print("tiny scene: %.1f us per frame (host-bound)" % ((time.perf_counter() - t) / 200 * 1e6))

step = make(500_000, 640, 480)
def run_single_thread(S, n):
    pool = [torch.cuda.Stream(device=dev) for _ in range(S)]
    for i in range(n):
        with torch.cuda.stream(pool[i % S]):
            step()
    torch.cuda.synchronize()
def run_threads(S, n):
    def work(k):
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            for _ in range(n // S): step()
    th = [threading.Thread(target=work, args=(k,)) for k in range(S)]
    for t_ in th: t_.start()
    for t_ in th: t_.join()
    torch.cuda.synchronize()
for S in (1, 2, 3, 4):
    for name, fn in (("one host thread", run_single_thread), ("thread per stream", run_threads)):
        fn(S, 12 * S); t = time.perf_counter(); fn(S, 60 * S if S != 0 else 60); dt = time.perf_counter() - t
        print("streams=%d %-18s %.0f frames/s" % (S, name, 60 * S / dt), flush=True)
