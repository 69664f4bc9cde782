"""Development helper: one fused mapping iteration (get_loss + backward + Adam) at a given size -- wall clock per iteration, the GPU's
busy time per iteration (sum of kernel times is not available here: hipEvents around the whole loop give wall = max(host, GPU)), and a
cProfile of the host side.  usage (GPU box): N=200000 W=256 H=256 python scripts/exp/map_iter.py"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import setup_camera, mapping as M, optim as O
from activesplat_amd import synthetic as syn

dev = torch.device("cuda")
N, W, H = int(os.environ.get("N", 200000)), int(os.environ.get("W", 256)), int(os.environ.get("H", 256))
params = syn.make_params(N, W, H, seed=0)
cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev)
prm = {k: torch.nn.Parameter(v.to(dev)) for k, v in params.items()}
prm["cam_unnorm_rots"] = torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0], device=dev).reshape(1, 4, 1))
prm["cam_trans"] = torch.nn.Parameter(torch.zeros(1, 3, 1, device=dev))
var = {k: torch.zeros(N, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
tim, tdepth = syn.make_targets(W, H)
data = dict(cam=cam, im=tim.to(dev), depth=tdepth.to(dev), id=0, w2c=torch.eye(4, device=dev))
opt = O.initialize_optimizer(prm, dict(means3D=1e-4, rgb_colors=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05, log_scales=1e-3,
                                       cam_unnorm_rots=0.0, cam_trans=0.0))
flags = dict(fused=True, fused_loss=True, fused_inputs=True, fused_preprocess=os.environ.get("RAW", "1") == "1")


def it():
    if os.environ.get("DIRECT") == "1":                       # the iteration's four library calls without autograd
        M.mapping_iteration(prm, data, var, 0, dict(im=0.5, depth=1.0), opt, pose7=[1.0, 0, 0, 0, 0, 0, 0])
        return
    loss, _, _ = M.get_loss(prm, data, var, 0, dict(im=0.5, depth=1.0), pose7=[1.0, 0, 0, 0, 0, 0, 0],
                            fused_adam=opt if os.environ.get("ADAM") == "1" else None, **flags)      # ADAM=1: the step inside the backward kernel
    loss.backward(M.unit_gradient(loss))
    with torch.no_grad():
        opt.step(); opt.zero_grad(set_to_none=True)


if os.environ.get("MT") == "0":                             # the backward on the calling thread instead of autograd's device thread (no hand-over per iteration)
    torch.autograd.set_multithreading_enabled(False)
for _ in range(30):
    it()
K = int(os.environ.get("K", 300))
ts = []
for _ in range(int(os.environ.get("BATCHES", 8))):          # host-side timing is noisy (shared hosts): best and median of several batches
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K):
        it()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) / K * 1e6)
print("N=%d %dx%d: %.1f us per mapping iteration (wall; best of %d batches, median %.1f)" % (N, W, H, min(ts), len(ts), sorted(ts)[len(ts) // 2]))
if os.environ.get("PROFILE", "1") == "1":
    pr = cProfile.Profile(); pr.enable()
    for _ in range(K):
        it()
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
