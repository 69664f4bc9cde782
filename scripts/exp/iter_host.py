"""Development helper: where the host time of one fused mapping iteration goes (small map: the GPU work is short, the loop is host-bound)."""
import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import setup_camera, mapping as M, optim as O
from activesplat_amd import synthetic as syn

dev = torch.device("cuda")
N, W, H = int(os.environ.get("N", 200_000)), 256, 256
cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev)
params = syn.make_params(N, W, H, seed=0)
prm = {k: torch.nn.Parameter(v.to(dev)) for k, v in params.items()}
prm["cam_unnorm_rots"] = torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0], device=dev).reshape(1, 4, 1))
prm["cam_trans"] = torch.nn.Parameter(torch.zeros(1, 3, 1, device=dev))
var = {k: torch.zeros(N, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
tim, tdepth = syn.make_targets(W, H)
data = dict(cam=cam, im=tim.to(dev), depth=tdepth.to(dev), id=0, w2c=torch.eye(4, device=dev))
opt = O.initialize_optimizer(prm, dict(means3D=1e-4, rgb_colors=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05,
                                       log_scales=1e-3, cam_unnorm_rots=0.0, cam_trans=0.0))
flags = dict(fused=True, fused_loss=True, fused_inputs=True)
if os.environ.get("HALF"):
    from activesplat_amd import _lib
    _lib.get().gs_set_half_quadrants(int(os.environ["HALF"]))

def it():
    loss, _, _ = M.get_loss(prm, data, var, 0, dict(im=0.5, depth=1.0), pose7=[1.0, 0, 0, 0, 0, 0, 0], **flags)
    loss.backward()
    with torch.no_grad():
        opt.step(); opt.zero_grad(set_to_none=True)

for _ in range(20): it()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(200): it()
torch.cuda.synchronize(); wall = (time.perf_counter() - t) / 200
# GPU time of the same iterations: events around a batch issued ahead
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("N=%d %dx%d: %.1f us per iteration wall" % (N, W, H, wall * 1e6))
if os.environ.get("PROFILE"):
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): it()
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
