"""Synthetic WORKLOADS of the BASELINE configurations that more than one caller runs (bench.py legs, tests, scripts/)."""
from __future__ import annotations

import numpy as np
import torch

from . import rasterizer as R
from . import synthetic as syn
from .camera import setup_camera


def configs2_optimise_loop(N, iters, device, densify_fn=None, densify_every=50, sh_degree=3, W=640, H=480, seed=0, time_it=False, raw=True,
                           fused_adam=True):
    """BASELINE configs[2]: N Gaussians with SH coefficients, one 640x480 target, `iters` iterations of
    fused activations -> single-pass RGB-D render -> fused loss -> backward -> densify (every `densify_every`) -> fused Adam.
    fused_adam (with raw): on iterations without a densify event the Adam step rides in the backward's per-Gaussian kernel
    (render_rgbd_raw(adam=...)); same parameters afterwards, bit for bit.
    densify_fn: optim.densify unless a caller compares another implementation of the same call (the tests' step-by-step baseline).
    Returns dict(losses=[first, last], counts=[N after every densify event], seconds)."""
    import time
    from activesplat_amd import mapping as M, optim as O
    dev = torch.device(device)
    p = syn.make_params(N, W, H, seed=seed, sh_degree=sh_degree)
    params = {k: torch.nn.Parameter(p[k].to(dev)) for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales")}
    params["shs"] = torch.nn.Parameter(p["shs"].to(dev))
    params["cam_unnorm_rots"] = torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0], device=dev).reshape(1, 4, 1))
    params["cam_trans"] = torch.nn.Parameter(torch.zeros(1, 3, 1, device=dev))
    lrs = dict(means3D=1e-4, shs=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05, log_scales=1e-3, cam_unnorm_rots=0.0, cam_trans=0.0)
    opt = O.initialize_optimizer(params, lrs)
    variables = {k: torch.zeros(N, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
    variables["scene_radius"] = torch.tensor(4.0 / 3.0, device=dev)
    cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev, sh_degree=sh_degree)
    gt_im, gt_depth = (t.to(dev) for t in syn.make_targets(W, H))
    ddict = dict(start_after=0, remove_big_after=0, stop_after=iters, densify_every=densify_every, grad_thresh=0.0002, num_to_split_into=2,
                 removal_opacity_threshold=0.005, final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=3000)
    state = dict(params=params, variables=variables)
    counts, losses, densify_s = [], [], []

    def one_iter(it, densify=True):
        params, variables = state["params"], state["variables"]
        if raw:
            # the per-Gaussian kernels take the parameters themselves: frame transform + activations inside, no activation launches
            m2d = torch.empty_like(params["means3D"], requires_grad=True)
            rv = {"means2D": m2d}
            in_backward = fused_adam and not (it > 0 and densify and O.densify_event(it, ddict))
            im, radius, depth, sil, dsq = R.render_rgbd_raw(cam, params["means3D"], m2d, params["logit_opacities"], params["log_scales"],
                                                            params["unnorm_rotations"], [1.0, 0, 0, 0, 0, 0, 0], shs=params["shs"],
                                                            adam=opt if in_backward else None)
        else:
            rv = M.fused_rendervar(dict(params, rgb_colors=params["shs"]), 0, [1.0, 0, 0, 0, 0, 0, 0])
            rv.pop("colors_precomp")
            im, radius, depth, sil, dsq = R.render_rgbd(cam, shs=params["shs"], **rv)
        loss, _ = M.fused_mapping_loss(im, depth, dsq, gt_im, gt_depth, dict(im=0.5, depth=1.0))
        with M.backward_on_calling_thread():
            loss.backward(M.unit_gradient(loss))         # (cached dL/dloss = 1: no fill launch, and the fused loss skips its scaling launch)
        variables["means2D"] = rv["means2D"]
        variables["seen"] = O.visibility_stats(radius, variables["max_2D_radius"])      # one launch: seen + running max radius in place
        with torch.no_grad():
            if it > 0 and densify:
                n0 = params["means3D"].shape[0]
                event = it % densify_every == 0
                if event and time_it:
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                params, variables = (densify_fn or O.densify)(params, variables, opt, it, ddict)
                if event and time_it:
                    torch.cuda.synchronize(); densify_s.append(time.perf_counter() - t1)
                if params["means3D"].shape[0] != n0:
                    counts.append(int(params["means3D"].shape[0]))
            opt.step()
            opt.zero_grad(set_to_none=True)
        state["params"], state["variables"] = params, variables
        return loss

    if time_it:
        for _ in range(3):
            one_iter(0)
        if dev.type == "cuda":
            # the caching allocator's pool at the high-water mark of a densify event (new parameter + moment tensors: ~3 x 59 floats per
            # Gaussian), as a mapper that has densified before finds it: without it the timed event is a series of hipMalloc calls (1 -> 23 ms)
            pool = torch.empty(int(N * 1.05) * 3 * 59 * 4, dtype=torch.uint8, device=dev)
            del pool
            torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(iters):
        loss = one_iter(it)
        if it in (0, iters - 1):
            losses.append(float(loss.detach()))
    if dev.type == "cuda":
        torch.cuda.synchronize()
    return dict(losses=losses, counts=counts, seconds=time.perf_counter() - t0, densify_seconds=densify_s, final_N=int(state["params"]["means3D"].shape[0]))
