"""Where one configs[2] iteration (2M Gaussians, SH-3) spends its time: steady iterations vs densify events, and the
stages inside a steady iteration.  GPU box: python scripts/configs2_breakdown.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activesplat_amd import mapping as M, optim as O, rasterizer as R, setup_camera, _lib  # noqa: E402
from activesplat_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda")
N, W, H = int(os.environ.get("N", 2_000_000)), 640, 480
p = syn.make_params(N, W, H, seed=0, sh_degree=3)
params = {k: torch.nn.Parameter(p[k].to(dev)) for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales")}
params["shs"] = torch.nn.Parameter(p["shs"].to(dev))
lrs = dict(means3D=1e-4, shs=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05, log_scales=1e-3)
opt = O.initialize_optimizer(params, lrs)
variables = {k: torch.zeros(N, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
variables["scene_radius"] = torch.tensor(4.0 / 3.0, device=dev)
cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev, sh_degree=3)
gt_im, gt_depth = (t.to(dev) for t in syn.make_targets(W, H))
ddict = dict(start_after=0, remove_big_after=0, stop_after=1000, densify_every=50, grad_thresh=0.0002, num_to_split_into=2,
             removal_opacity_threshold=0.005, final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=3000)


def T():
    torch.cuda.synchronize()
    return time.perf_counter()


acc = {}


def tick(name, t0):
    t1 = T()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


def one_iter(it, timed):
    global params, variables
    t = T() if timed else 0
    rv = M.fused_rendervar(dict(params, rgb_colors=params["shs"]), 0, [1.0, 0, 0, 0, 0, 0, 0])
    rv.pop("colors_precomp")
    rv["means2D"].retain_grad()
    if timed: t = tick("activate", t)
    im, radius, depth, sil, dsq = R.render_rgbd(cam, shs=params["shs"], **rv)
    if timed: t = tick("render_fwd", t)
    loss, _ = M.fused_mapping_loss(im, depth, dsq, gt_im, gt_depth, dict(im=0.5, depth=1.0))
    if timed: t = tick("loss", t)
    loss.backward()
    if timed: t = tick("backward(loss+render+activate)", t)
    variables["means2D"], variables["seen"] = rv["means2D"], radius > 0
    variables["max_2D_radius"] = torch.maximum(variables["max_2D_radius"], radius.float())
    with torch.no_grad():
        if it > 0:
            params, variables = O.densify(params, variables, opt, it, ddict)
        if timed: t = tick("densify_call(accumulate only)" if it % 50 else "densify_event", t)
        opt.step()
        opt.zero_grad(set_to_none=True)
        if timed: t = tick("adam", t)


for it in range(1, 4):
    one_iter(it, False)
t0 = T()
for it in range(1, 21):
    one_iter(it, False)
steady = (T() - t0) / 20
for it in range(21, 41):
    one_iter(it, True)
print("steady iteration (no per-stage syncs): %.3f ms" % (steady * 1e3))
for k, v in acc.items():
    print("  %-34s %.3f ms" % (k, v / 20 * 1e3))
acc.clear()
t0 = T(); one_iter(50, True); print("densify-event iteration: %.1f ms, N %d -> %d" % ((T() - t0) * 1e3, N, params["means3D"].shape[0]))
for k, v in acc.items():
    print("  %-34s %.3f ms" % (k, v * 1e3))
lib = _lib.get(); lib.gs_profile_enable(1)
for it in range(51, 61):
    one_iter(it, False)
torch.cuda.synchronize()
for k, (ms, c) in _lib.profile_collect().items():
    if c: print("  stage %-28s %.1f us x%d" % (k, ms / c * 1e3, c // 10))
