"""Time the parts of one densify event at 2M Gaussians / SH-3 (first and second event).  GPU box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activesplat_amd import optim as O, synthetic as syn  # noqa: E402
dev = torch.device("cuda")
N = int(os.environ.get("N", 2_000_000))
p = syn.make_params(N, 640, 480, seed=0, sh_degree=3)
params = {k: torch.nn.Parameter(p[k].to(dev)) for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales")}
params["shs"] = torch.nn.Parameter(p["shs"].to(dev))
lrs = dict(means3D=1e-4, shs=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05, log_scales=1e-3)
opt = O.initialize_optimizer(params, lrs)
for k, v in params.items():
    v.grad = torch.randn_like(v) * 1e-3
opt.step()
variables = {k: torch.zeros(N, device=dev) for k in ("max_2D_radius", "denom", "timestep")}
variables["means2D_gradient_accum"] = torch.rand(N, device=dev) * 4e-4
variables["denom"] += 1
variables["scene_radius"] = torch.tensor(4.0 / 3.0, device=dev)
ddict = dict(start_after=0, remove_big_after=0, stop_after=1000, densify_every=50, grad_thresh=0.0002, num_to_split_into=2,
             removal_opacity_threshold=0.005, final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=3000)
acc = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0) + time.perf_counter() - t
        return r
    setattr(mod, name, g)
for n in ("build_index", "gather_rows", "cat_params_to_optimizer", "remove_points", "accumulate_mean2d_gradient"):
    wrap(O, n)
for ev in range(3):
    acc.clear()
    variables["means2D_gradient_accum"] = torch.rand(params["means3D"].shape[0], device=dev) * 4e-4
    variables["denom"] = torch.ones(params["means3D"].shape[0], device=dev)
    variables["means2D"] = torch.zeros(params["means3D"].shape[0], 3, device=dev)          # .grad is None: nothing to accumulate
    torch.cuda.synchronize(); t0 = time.perf_counter()
    params, variables = O.densify(params, variables, opt, 50, ddict)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"event {ev}: {dt*1e3:.1f} ms  N={params['means3D'].shape[0]}  reserved={torch.cuda.memory_reserved()/2**30:.1f} GiB")
    for k, v in acc.items():
        print(f"    {k:32s} {v*1e3:.2f} ms")
