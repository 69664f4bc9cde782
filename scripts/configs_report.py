"""BASELINE configs[2] (optimise loop) and the configs[4] substitute (mapper harness on a synthetic RGB-D spin):
numbers for profiles/.  Run on the GPU box:  python scripts/configs_report.py > gpurun_out/configs.json"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activesplat_amd import mapping as M, optim as O, rasterizer as R, setup_camera  # noqa: E402
from activesplat_amd import synthetic as syn  # noqa: E402
from activesplat_amd.mapper import SplatMapper  # noqa: E402

dev = torch.device("cuda")
out = {}

# ---- configs[2]: 2M Gaussians, SH degree 3, 640x480, 100-iteration optimise loop, fused Adam + densify/prune ----
N, W, H, ITERS = int(os.environ.get("N", 2_000_000)), 640, 480, 100
p = syn.make_params(N, W, H, seed=0, sh_degree=3)
params = {k: torch.nn.Parameter(p[k].to(dev)) for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales")}
params["shs"] = torch.nn.Parameter(p["shs"].to(dev))
params["cam_unnorm_rots"] = torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0], device=dev).reshape(1, 4, 1))
params["cam_trans"] = torch.nn.Parameter(torch.zeros(1, 3, 1, device=dev))
lrs = dict(means3D=1e-4, shs=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05, log_scales=1e-3, cam_unnorm_rots=0.0, cam_trans=0.0)
opt = O.initialize_optimizer(params, lrs)
variables = {k: torch.zeros(N, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
variables["scene_radius"] = torch.tensor(4.0 / 3.0, device=dev)
cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev, sh_degree=3)
gt_im, gt_depth = (t.to(dev) for t in syn.make_targets(W, H))
ddict = dict(start_after=0, remove_big_after=0, stop_after=ITERS, densify_every=50, grad_thresh=0.0002, num_to_split_into=2,
             removal_opacity_threshold=0.005, final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=3000)
losses, counts = [], [N]


def one_iter(it):
    global params, variables
    rv = M.fused_rendervar(dict(params, rgb_colors=params["shs"]), 0, [1.0, 0, 0, 0, 0, 0, 0])
    rv.pop("colors_precomp")
    rv["means2D"].retain_grad()
    im, radius, depth, sil, dsq = R.render_rgbd(cam, shs=params["shs"], **rv)
    loss, _ = M.fused_mapping_loss(im, depth, dsq, gt_im, gt_depth, dict(im=0.5, depth=1.0))
    loss.backward()
    variables["means2D"], variables["seen"] = rv["means2D"], radius > 0
    variables["max_2D_radius"] = torch.maximum(variables["max_2D_radius"], radius.float())
    with torch.no_grad():
        if it > 0:
            params, variables = O.densify(params, variables, opt, it, ddict)
            if params["means3D"].shape[0] != counts[-1]:
                counts.append(int(params["means3D"].shape[0]))
        opt.step()
        opt.zero_grad(set_to_none=True)
    return loss


for it in range(3):
    one_iter(0)
# one densify event on a throw-away 4k-Gaussian copy: the first call of each torch op / library kernel pays a one-time
# code-object load (~0.2-0.4 s in total) that has nothing to do with the 100 timed iterations
_wp = {k: torch.nn.Parameter(v.detach()[:4096].clone()) for k, v in params.items() if k not in ("cam_unnorm_rots", "cam_trans")}
_wo = O.initialize_optimizer(_wp, {k: lrs[k] for k in _wp})
for v in _wp.values():
    v.grad = torch.zeros_like(v)
_wo.step()
_wv = {k: torch.ones(4096, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
_wv.update(scene_radius=variables["scene_radius"], means2D=torch.zeros(4096, 3, device=dev), seen=torch.ones(4096, dtype=torch.bool, device=dev))
O.densify(_wp, _wv, _wo, 50, ddict)
del _wp, _wo, _wv
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(ITERS):
    l = one_iter(it)
    if it in (0, ITERS - 1):
        losses.append(float(l))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out["configs2_optimise_loop"] = dict(gaussians_start=N, gaussians_after_densify=counts, sh_degree=3, iters=ITERS, seconds=round(dt, 4),
                                     iters_per_s=round(ITERS / dt, 2), loss_first=losses[0], loss_last=losses[-1],
                                     densify="every 50 iterations (grad_thresh 2e-4, split into 2, opacity cull 0.005), anisotropic rule of DESIGN.md",
                                     note="fused activations + single-pass RGB-D render (SH-3) + fused loss + fused Adam (5 tensors incl. shs)")
del params, opt, variables
torch.cuda.empty_cache()

# ---- configs[4] substitute: mapper harness on a synthetic RGB-D spin (no Habitat / Gibson / ROS here) ----
W, H, FR = 256, 256, 31
gt = syn.shell_scene(400_000, seed=2, W=W, H=H)
gt["logit_opacities"] = gt["logit_opacities"] + 3.0
seq = list(syn.orbit_sequence(gt, FR, W, H, dev))
rep = {}
for name, flags in (("reference_call_pattern", {}), ("fused", dict(fused_render=True, fused_loss=True, fused_inputs=True))):
    mp = SplatMapper(syn.intrinsics(W, H), W, H, config=dict(step_num=FR, mapping_iters=10, **flags), device=dev)   # high-res setting: 2 iters/frame
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for fr in seq:
        mp.run(fr)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ps = []
    for fr in seq[::5]:
        im, depth, opacity = mp.render_rgbd(fr["w2c"])
        seen = (fr["depth"] > 0)[0]
        mse = float(((im[:, seen] - fr["color"][:, seen]) ** 2).mean())
        ps.append(10 * np.log10(1.0 / mse))
    rep[name] = dict(frames=FR, seconds=round(dt, 3), iterations=mp.stats["iters"], ms_per_iteration=round(mp.stats["iter_time"] / max(mp.stats["iters"], 1) * 1e3, 3),
                     gaussians=int(mp.params["means3D"].shape[0]), psnr_db_vs_synthetic_gt=round(float(np.mean(ps)), 2))
out["configs4_substitute_mapper_harness"] = dict(rep, note="synthetic in-place spin (10 degree turns) inside a 400k-Gaussian ground-truth scene, 256x256; "
                                                 "Habitat-sim, Gibson data, ROS and a reference checkpoint are unavailable (SURVEY 8d)")
print(json.dumps(out, indent=1))
