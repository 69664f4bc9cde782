"""Development helper: the fused mapping loss alone (gs_mapping_loss: loss_stats_kernel + loss_grad_kernel) on a 640x480 frame -- wall time per call over
back-to-back calls, for rocprofv3 kernel tables / counter passes of just these two kernels.   W=640 H=480 CALLS=200 python scripts/exp/loss_only.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import mapping as M
from activesplat_amd import synthetic as syn
W, H, n = int(os.environ.get("W", 640)), int(os.environ.get("H", 480)), int(os.environ.get("CALLS", 200))
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
im = torch.rand(3, H, W, generator=g).to(dev).requires_grad_(True)
depth = (torch.rand(1, H, W, generator=g) * 3 + 0.5).to(dev).requires_grad_(True)
dsq = (depth.detach() ** 2 + 0.01).contiguous()
tim, tdepth = (t.to(dev) for t in syn.make_targets(W, H))
for _ in range(10):
    loss, _ = M.fused_mapping_loss(im, depth, dsq, tim, tdepth, dict(im=0.5, depth=1.0))
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(n):
    loss, _ = M.fused_mapping_loss(im, depth, dsq, tim, tdepth, dict(im=0.5, depth=1.0))
torch.cuda.synchronize()
print("fused mapping loss %dx%d: %.2f us per call (wall, %d calls back to back), loss %.6f" % (W, H, (time.perf_counter() - t) / n * 1e6, n, float(loss)))
