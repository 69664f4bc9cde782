"""Caller-side helpers of the rasteriser hot path: frame transforms, activations, loss, map growth.

Host-side mirror (PyTorch plumbing, explicit `device`, no hard-coded .cuda()) of the reference
functions that produce every tensor handed to / consumed from the rasteriser:

  build_rotation                          src/mapper/splatam/utils/slam_external.py:25-42
  quat_mult, l1_loss_v1                   src/mapper/splatam/utils/slam_helpers.py:5-6,21-28
  transform_to_frame                      slam_helpers.py:252-304
  transformed_params2rendervar            slam_helpers.py:124-139
  get_depth_and_silhouette,
  transformed_params2depthplussilhouette  slam_helpers.py:196-213,234-249
  get_rendervars, render                  src/mapper/splatam/splatam.py:436-468,413-434
  calc_ssim                               slam_external.py:54-97
  get_loss (mapping branch)               splatam.py:172-301
  get_pointcloud, initialize_params,
  initialize_new_params, add_new_gaussians splatam.py:25-115,304-379

Same names, argument meaning and results (pinned by tests/golden/*.npz generated from the reference);
the reference's unused per-iteration setup_camera call (splatam.py:205) is omitted.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from .camera import setup_camera
from .rasterizer import GaussianRasterizationSettings as Camera
from .rasterizer import GaussianRasterizer as Renderer
from .rasterizer import render_rgbd, render_rgbd_raw

_GAUSSIAN_KEYS = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")


def build_rotation(q: torch.Tensor) -> torch.Tensor:
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)


def quat_mult(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    w1, x1, y1, z1 = q1.unbind(-1)
    w2, x2, y2, z2 = q2.unbind(-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)


def l1_loss_v1(x, y):
    return (x - y).abs().mean()


def transform_to_frame(params, time_idx, gaussians_grad, camera_grad):
    """World -> camera frame `time_idx` for isotropic or anisotropic Gaussians (slam_helpers.py:252-304)."""
    cam_rot = params["cam_unnorm_rots"][..., time_idx]
    cam_tran = params["cam_trans"][..., time_idx]
    if not camera_grad:
        cam_rot, cam_tran = cam_rot.detach(), cam_tran.detach()
    cam_rot = F.normalize(cam_rot)
    dev = cam_rot.device
    rel_w2c = torch.eye(4, device=dev, dtype=torch.float32)
    rel_w2c[:3, :3] = build_rotation(cam_rot)
    rel_w2c[:3, 3] = cam_tran
    pts, rots = params["means3D"], params["unnorm_rotations"]
    if not gaussians_grad:
        pts, rots = pts.detach(), rots.detach()
    out = {"means3D": pts @ rel_w2c[:3, :3].T + rel_w2c[:3, 3]}
    if params["log_scales"].shape[1] == 1:          # isotropic: rotation is irrelevant
        out["unnorm_rotations"] = rots
    else:
        out["unnorm_rotations"] = quat_mult(cam_rot, F.normalize(rots))
    return out


def _log_scales3(params):
    ls = params["log_scales"]
    return torch.tile(ls, (1, 3)) if ls.shape[1] == 1 else ls


def transformed_params2rendervar(params, transformed_gaussians):
    m = transformed_gaussians["means3D"]
    return {"means3D": m, "colors_precomp": params["rgb_colors"],
            "rotations": F.normalize(transformed_gaussians["unnorm_rotations"]),
            "opacities": torch.sigmoid(params["logit_opacities"]), "scales": torch.exp(_log_scales3(params)),
            "means2D": torch.zeros_like(params["means3D"], requires_grad=True) + 0}


def get_depth_and_silhouette(pts_3D, w2c):
    """Per-Gaussian "colour" [z_cam, 1, z_cam^2] for the depth/silhouette pass (slam_helpers.py:196-213)."""
    w2c = torch.as_tensor(w2c, dtype=torch.float32, device=pts_3D.device)
    z = pts_3D @ w2c[2, :3] + w2c[2, 3]
    return torch.stack([z, torch.ones_like(z), z * z], dim=1)


def transformed_params2depthplussilhouette(params, w2c, transformed_gaussians):
    m = transformed_gaussians["means3D"]
    return {"means3D": m, "colors_precomp": get_depth_and_silhouette(m, w2c),
            "rotations": F.normalize(transformed_gaussians["unnorm_rotations"]),
            "opacities": torch.sigmoid(params["logit_opacities"]), "scales": torch.exp(_log_scales3(params)),
            "means2D": torch.zeros_like(params["means3D"], requires_grad=True) + 0}


def get_rendervars(params, w2c):
    """World-frame rendervars for the no-grad consumers (splatam.py:436-468)."""
    base = {"means3D": params["means3D"], "rotations": F.normalize(params["unnorm_rotations"]),
            "opacities": torch.sigmoid(params["logit_opacities"]), "scales": torch.exp(_log_scales3(params))}
    rv = dict(base, colors_precomp=params["rgb_colors"], means2D=torch.zeros_like(params["means3D"]))
    dv = dict(base, colors_precomp=get_depth_and_silhouette(params["means3D"], w2c),
              means2D=torch.zeros_like(params["means3D"]))
    return rv, dv


def render(w2c, k, rendervar, depth_rendervar, cfg, scale_modifier=1.0, device=None, with_silhouette=True):
    """No-grad render (splatam.py:413-434): im on white, built-in depth & opacity, silhouette channel.
    with_silhouette=False skips the second raster pass, whose only product every reference caller discards
    (SURVEY App. E5)."""
    with torch.no_grad():
        device = rendervar["means3D"].device if device is None else device
        cam = setup_camera(cfg["viz_w"], cfg["viz_h"], k, w2c, cfg["viz_near"], cfg["viz_far"],
                           scale_modifier=scale_modifier, device=device)
        white = cam._replace(bg=torch.ones(3, dtype=torch.float32, device=device))
        im, _, depth, opacity = Renderer(raster_settings=white)(**rendervar)
        sil = None
        if with_silhouette:
            depth_sil, _, _, _ = Renderer(raster_settings=cam)(**depth_rendervar)
            sil = depth_sil[1].unsqueeze(0)
        return im, depth, opacity, sil


# ---------------------------------------------------------------------------------------------------
# loss (splatam.py:172-301, mapping branch) -- SSIM: 11x11 Gaussian window sigma 1.5, C1 1e-4, C2 9e-4
# ---------------------------------------------------------------------------------------------------
_WINDOWS = {}


def _window(channel, device, dtype):
    key = (channel, str(device), dtype)
    if key not in _WINDOWS:
        g = torch.tensor([math.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
        g = (g / g.sum()).unsqueeze(1)
        _WINDOWS[key] = (g @ g.T).expand(channel, 1, 11, 11).contiguous().to(device=device, dtype=dtype)
    return _WINDOWS[key]


def calc_ssim(img1, img2):
    ch = img1.size(-3)
    w = _window(ch, img1.device, img1.dtype)
    conv = lambda x: F.conv2d(x, w, padding=5, groups=ch)  # noqa: E731
    mu1, mu2 = conv(img1), conv(img2)
    s11 = conv(img1 * img1) - mu1 * mu1
    s22 = conv(img2 * img2) - mu2 * mu2
    s12 = conv(img1 * img2) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))).mean()


class _FusedRendervars(torch.autograd.Function):
    """gs_activate_forward / gs_activate_backward (csrc/activate.hip): transform_to_frame +
    transformed_params2rendervar in one launch each way."""

    @staticmethod
    def forward(ctx, means3D, unnorm_rotations, logit_opacities, log_scales, pose7, accumulate=False):
        import ctypes as C
        from . import _lib
        lib = _lib.get()
        dev = means3D.device
        P = int(means3D.shape[0])
        iso = 1 if log_scales.shape[1] == 1 else 0
        c = lambda t: t.detach().contiguous().float()  # noqa: E731
        m, r, o, s = c(means3D), c(unnorm_rotations), c(logit_opacities), c(log_scales)
        om, orr = torch.empty(P, 3, device=dev), torch.empty(P, 4, device=dev)
        oo, os_ = torch.empty(P, 1, device=dev), torch.empty(P, 3, device=dev)
        pose = (C.c_float * 7)(*[float(v) for v in pose7])
        st = _lib.stream_ptr(dev)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        _lib.check(lib.gs_activate_forward(P, iso, pose, p(m), p(r), p(o), p(s), p(om), p(orr), p(oo), p(os_), st))
        ctx.save_for_backward(r, oo, os_)
        ctx.pose, ctx.iso, ctx.st = pose, iso, st
        # accumulate: the backward adds its four gradients to the leaves' .grad inside the kernel (and hands autograd nothing)
        ctx.leaves = (means3D, unnorm_rotations, logit_opacities, log_scales) if accumulate else None
        return om, orr, oo, os_

    @staticmethod
    def backward(ctx, gm, gr, go, gs_):
        import ctypes as C
        from . import _lib
        lib = _lib.get()
        r, oo, os_ = ctx.saved_tensors
        P, dev = int(r.shape[0]), r.device
        c = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        gm, gr, go, gs_ = c(gm), c(gr), c(go), c(gs_)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        st = _lib.stream_ptr(dev)
        leaves = ctx.leaves
        if leaves is not None and all(x.is_leaf and x.requires_grad for x in leaves):
            # a keyframe batch (parallel.sharded_keyframe_step): what autograd would do with the four results is `leaf.grad += result`,
            # 3 x 44 bytes per Gaussian and keyframe of extra passes -- the kernel adds in place instead
            have = [x.grad is not None and x.grad.is_contiguous() and x.grad.dtype == torch.float32 and x.grad.shape == x.shape for x in leaves]
            if all(have):
                g = [x.grad for x in leaves]
                _lib.check(lib.gs_activate_backward_accumulate(P, ctx.iso, ctx.pose, p(r), p(oo), p(os_), p(gm), p(gr), p(go), p(gs_),
                                                               p(g[0]), p(g[1]), p(g[2]), p(g[3]), st))
                return None, None, None, None, None, None
            if not any(x.grad is not None for x in leaves):
                g = [torch.empty_like(x, memory_format=torch.contiguous_format) for x in leaves]
                _lib.check(lib.gs_activate_backward(P, ctx.iso, ctx.pose, p(r), p(oo), p(os_), p(gm), p(gr), p(go), p(gs_),
                                                    p(g[0]), p(g[1]), p(g[2]), p(g[3]), st))
                for x, t in zip(leaves, g):
                    x.grad = t
                return None, None, None, None, None, None
        dm, dr = torch.empty(P, 3, device=dev), torch.empty(P, 4, device=dev)
        dl, ds = torch.empty(P, 1, device=dev), torch.empty(P, 1 if ctx.iso else 3, device=dev)
        _lib.check(lib.gs_activate_backward(P, ctx.iso, ctx.pose, p(r), p(oo), p(os_), p(gm), p(gr), p(go), p(gs_), p(dm), p(dr),
                                            p(dl), p(ds), st))
        return dm, dr, dl, ds, None, None


def fused_rendervar(params, time_idx, pose7=None, accumulate_grads=False):
    """rendervar dict of transform_to_frame(gaussians_grad=True, camera_grad=False) +
    transformed_params2rendervar, built by ONE HIP launch (and one more in the backward).  pose7 = host
    (qw,qx,qy,qz,tx,ty,tz) of the frame's relative w2c; read from params['cam_*'] (one small D2H) if omitted."""
    if pose7 is None:
        q = F.normalize(params["cam_unnorm_rots"][..., time_idx].detach()).reshape(4)
        pose7 = torch.cat([q, params["cam_trans"][..., time_idx].detach().reshape(3)]).cpu().tolist()
    m, r, o, s = _FusedRendervars.apply(params["means3D"], params["unnorm_rotations"], params["logit_opacities"],
                                        params["log_scales"], pose7, accumulate_grads)
    # means2D only exists to receive the screen-space gradient: a fresh LEAF (autograd hands it the rasteriser's gradient
    # tensor as .grad without a copy; the reference's `zeros + 0` non-leaf costs an add and a retain_grad clone).  Its VALUES are
    # never read -- the rasteriser takes the tensor as a gradient carrier only -- so it is not filled either (one launch less)
    return {"means3D": m, "colors_precomp": params["rgb_colors"], "rotations": r, "opacities": o, "scales": s,
            "means2D": torch.empty_like(params["means3D"], requires_grad=True)}


#: dL/dloss tensors that are known to hold exactly 1 (the callers' cached root gradients: mapper, parallel.sharded_keyframe_step).  Kept alive
#: here, so that an address can never come back as another tensor's; _FusedMappingLoss.backward skips its `g * grads` launch for them.
_UNIT_GRADS = {}


def backward_on_calling_thread():
    """Context manager for `loss.backward()` in a mapping loop: autograd runs the backward of CUDA tensors on a per-device worker thread and the calling
    thread waits for it -- two thread hand-overs per iteration, which on a many-core host cost nothing or ~100-190 us depending on where the scheduler
    has put the two threads (measured at the reference's 256 x 256 / 200 k frame: 315-390 us per mapping iteration in the slow placement, 199-223 us in
    the fast one, bimodal from run to run; `scripts/exp/ab_host_mt.sh`).  With multithreading off the backward runs on the calling thread: 199-223 us,
    median = best.  One device per process (this package's layout), so nothing is lost.  A reference checkout gets the same with ONE line around
    its loop."""
    return torch.autograd.set_multithreading_enabled(False)


def unit_gradient(like: torch.Tensor) -> torch.Tensor:
    """A cached scalar 1 of `like`'s device and dtype for `loss.backward(unit_gradient(loss))`."""
    key = (str(like.device), like.dtype)
    t = _UNIT_GRADS.get(key)
    if t is None:
        t = _UNIT_GRADS[key] = torch.ones((), device=like.device, dtype=like.dtype)
    return t


#: persistent scratch of the fused loss per (device, stream, W, H): [buffer, calls so far] -- zeroed ONCE here; every call's second kernel
#: clears the accumulator set of the next call (gs_mapping_loss(persistent_call = k)), so no memset is launched per loss
_LOSS_SCRATCH = {}


#: bound on the bytes the cache pins (a 640 x 480 entry is 11 MB; streams and sizes come and go): oldest entries leave first
_LOSS_SCRATCH_MAX_BYTES = 256 << 20


def _loss_scratch(lib, dev, W, H):
    """-> (key, scratch, persistent_call).  The entry's call counter is advanced by _loss_scratch_done() only AFTER the call succeeded: a call that
    raises drops its entry (the accumulator set it was to use may be half-written, and the next call's set was not cleared), so the next call starts
    from a freshly zeroed scratch.  While the stream is being captured into a graph the persistent protocol is not used at all (a replay would bake one
    parity in and never clear the set it accumulates into): persistent_call = 0, the library's memset form, on a private buffer."""
    from . import _lib
    if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
        return None, torch.empty(int(lib.gs_mapping_loss_scratch_bytes(W, H)), dtype=torch.uint8, device=dev), 0
    key = (dev.index, _lib.stream_handle(dev), W, H)
    ent = _LOSS_SCRATCH.get(key)
    if ent is None:
        nbytes = int(lib.gs_mapping_loss_scratch_bytes(W, H))
        while _LOSS_SCRATCH and (len(_LOSS_SCRATCH) >= 32 or sum(e[0].numel() for e in _LOSS_SCRATCH.values()) + nbytes > _LOSS_SCRATCH_MAX_BYTES):
            _LOSS_SCRATCH.pop(next(iter(_LOSS_SCRATCH)))
        ent = _LOSS_SCRATCH[key] = [torch.zeros(nbytes, dtype=torch.uint8, device=dev), 0]
    return key, ent[0], ent[1] + 1


def _loss_scratch_done(key, ok):
    if key is None:
        return
    if ok:
        ent = _LOSS_SCRATCH.get(key)
        if ent is not None:
            ent[1] += 1
    else:
        _LOSS_SCRATCH.pop(key, None)


class _FusedMappingLoss(torch.autograd.Function):
    """gs_mapping_loss: value and gradients in two HIP launches (csrc/loss.hip)."""

    @staticmethod
    def forward(ctx, im, depth, depth_sq, gt_im, gt_depth, w_im, w_depth):
        import ctypes as C
        from . import _lib
        lib = _lib.get()
        H, W = int(im.shape[1]), int(im.shape[2])
        dev = im.device
        c = lambda t: None if t is None else t.detach().contiguous().float()  # noqa: E731
        im_, depth_, dsq_, gt_, gtd_ = c(im), c(depth), c(depth_sq), c(gt_im), c(gt_depth)
        buf = torch.empty(4, dtype=torch.float32, device=dev)            # {loss, im term, depth term, loss}
        grads = torch.empty(4, H, W, dtype=torch.float32, device=dev)    # dL/dim [3,H,W] and dL/ddepth [1,H,W] in ONE buffer
        d_im, d_depth = grads[:3], grads[3:]
        st = _lib.stream_ptr(dev)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        key, scratch, call = _loss_scratch(lib, dev, W, H)
        try:
            _lib.check(lib.gs_mapping_loss(W, H, p(im_), p(gt_), p(depth_), p(dsq_), p(gtd_), float(w_im), float(w_depth), p(buf),
                                           p(d_im), p(d_depth), p(scratch), call, st))
        except Exception:
            _loss_scratch_done(key, False)
            raise
        _loss_scratch_done(key, True)
        ctx.save_for_backward(grads)
        ctx.set_materialize_grads(False)
        losses, loss = buf[:3], buf[3]          # two views of one small buffer: no clone kernel for the scalar
        ctx.mark_non_differentiable(losses)
        return loss, losses

    @staticmethod
    def backward(ctx, g, _gl=None):
        (grads,) = ctx.saved_tensors
        if g is None:
            return None, None, None, None, None, None, None
        u = _UNIT_GRADS.get((str(g.device), g.dtype))
        if u is not None and g.dim() == 0 and g.data_ptr() == u.data_ptr():
            gd = grads                           # the root gradient is the cached 1: no launch
        else:
            gd = g * grads                       # one launch for both gradients
        return gd[:3], gd[3:], None, None, None, None, None


def fused_mapping_loss(im, depth, depth_sq, gt_im, gt_depth, loss_weights):
    """(loss, {'im','depth','loss'}) with the reference's mapping-loss semantics, computed by the HIP library."""
    loss, parts = _FusedMappingLoss.apply(im, depth, depth_sq, gt_im, gt_depth, loss_weights["im"], loss_weights["depth"])
    return loss, {"im": parts[1], "depth": parts[2], "loss": loss}


def get_loss(params, curr_data, variables, iter_time_idx, loss_weights, use_sil_for_loss=True, sil_thres=0.99,
             use_l1=True, ignore_outlier_depth_loss=False, do_ba=False, fused=False, fused_loss=False, fused_inputs=False,
             pose7=None, accumulate_grads=False, fused_preprocess=False, fused_adam=None):
    """Mapping loss: masked depth L1 + 0.8 L1 + 0.2 (1 - SSIM) on colour; updates
    variables['means2D'|'seen'|'max_2D_radius'].
    fused=False: the reference's two raster passes on the same geometry (RGB, then [z,1,z^2]).
    fused=True : ONE pass (rasterizer.render_rgbd) -- valid because curr_data['w2c'] is the settings' view
                 matrix at every reference call site; means2D.grad then also carries the depth term (the
                 reference's densifier sees the colour pass only).
    fused_loss : the masked-L1 + L1 + SSIM loss and its gradients come from gs_mapping_loss (csrc/loss.hip) instead
                 of ~40 torch kernels.
    fused_inputs : transform_to_frame + activations by gs_activate_* (csrc/activate.hip).
    accumulate_grads (with fused_inputs, for `loss.backward()` over a batch of keyframes): the activation backward ADDS its gradients to
                 the four per-Gaussian parameters' .grad in the kernel instead of handing them to autograd's accumulation passes; gradients
                 taken with torch.autograd.grad are not delivered in this mode.
    fused_preprocess (with fused; isotropic or anisotropic scale / rotation parameters, `rgb_colors` or 16-coefficient `shs` rows): no activation launches at all -- the
                 rasteriser's per-Gaussian kernels take the PARAMETERS and do the frame transform + activations themselves, forward and
                 backward (rasterizer.render_rgbd_raw); with accumulate_grads the backward also adds into the parameters' .grad.
    fused_adam (with fused_preprocess; the GaussianAdam that owns the parameters): the backward kernel also applies this iteration's Adam
                 step to the five per-Gaussian tensors in place (no gradient tensors; a following optimizer.step() skips them)."""
    if fused_preprocess and fused and not do_ba:
        if pose7 is None:
            q = F.normalize(params["cam_unnorm_rots"][..., iter_time_idx].detach()).reshape(4)
            pose7 = torch.cat([q, params["cam_trans"][..., iter_time_idx].detach().reshape(3)]).cpu().tolist()
        m2d = torch.empty_like(params["means3D"], requires_grad=True)      # gradient carrier only (see fused_rendervar)
        # seen + the running max radius are written by the render's per-Gaussian kernel where the tensors allow it
        mx = variables["max_2D_radius"]
        stats_in_render = fused_loss and use_l1 and not ignore_outlier_depth_loss and mx.dtype == torch.float32 and mx.is_contiguous() \
            and mx.device == params["means3D"].device and mx.numel() == params["means3D"].shape[0]
        seen = torch.empty(mx.numel(), dtype=torch.bool, device=mx.device) if stats_in_render else None
        im, radius, depth, _sil, depth_sq = render_rgbd_raw(curr_data["cam"], params["means3D"], m2d, params["logit_opacities"],
                                                             params["log_scales"], params["unnorm_rotations"], pose7,
                                                             colors_precomp=None if "shs" in params else params["rgb_colors"],
                                                             shs=params.get("shs"), accumulate_grads=accumulate_grads, adam=fused_adam,
                                                             visibility=(mx, seen) if stats_in_render else None)
        variables["means2D"] = m2d
        if fused_loss and use_l1 and not ignore_outlier_depth_loss:
            loss, weighted = fused_mapping_loss(im, depth, depth_sq, curr_data["im"], curr_data["depth"], loss_weights)
            if stats_in_render:
                variables["seen"] = seen
            else:
                from . import optim as O
                variables["seen"] = O.visibility_stats(radius, variables["max_2D_radius"])
            return loss, variables, weighted
        rendervar = None
    elif fused_inputs and not do_ba:
        tg = None
        rendervar = fused_rendervar(params, iter_time_idx, pose7, accumulate_grads)
    else:
        tg = transform_to_frame(params, iter_time_idx, gaussians_grad=True, camera_grad=do_ba)
        rendervar = transformed_params2rendervar(params, tg)
    if rendervar is not None:
        rendervar["means2D"].retain_grad()
    if rendervar is None:
        pass                                              # (rendered above, raw-parameter mode; the torch loss below)
    elif fused:
        im, radius, depth, _sil, depth_sq = render_rgbd(curr_data["cam"], **rendervar)
    else:
        if tg is None:
            tg = transform_to_frame(params, iter_time_idx, gaussians_grad=True, camera_grad=do_ba)
        depth_sil_rendervar = transformed_params2depthplussilhouette(params, curr_data["w2c"], tg)
        im, radius, _, _ = Renderer(raster_settings=curr_data["cam"])(**rendervar)
        depth_sil, _, _, _ = Renderer(raster_settings=curr_data["cam"])(**depth_sil_rendervar)
        depth = depth_sil[0].unsqueeze(0)
        depth_sq = depth_sil[2].unsqueeze(0)
    if rendervar is not None:
        variables["means2D"] = rendervar["means2D"]      # densification reads the colour pass' gradient only
    if fused_loss and use_l1 and not ignore_outlier_depth_loss:
        loss, weighted = fused_mapping_loss(im, depth, depth_sq, curr_data["im"], curr_data["depth"], loss_weights)
        from . import optim as O
        variables["seen"] = O.visibility_stats(radius, variables["max_2D_radius"])      # one launch: seen + max radius in place
        return loss, variables, weighted
    uncertainty = (depth_sq - depth ** 2).detach()
    mask = curr_data["depth"] > 0
    if ignore_outlier_depth_loss:
        err = (curr_data["depth"] - depth).abs() * mask
        mask = mask & (err < 10 * err.median())
    mask = (mask & ~torch.isnan(depth) & ~torch.isnan(uncertainty)).detach()
    losses = {}
    if use_l1:
        losses["depth"] = (curr_data["depth"] - depth).abs()[mask].mean()
    losses["im"] = 0.8 * l1_loss_v1(im, curr_data["im"]) + 0.2 * (1.0 - calc_ssim(im, curr_data["im"]))
    weighted = {k: v * loss_weights[k] for k, v in losses.items()}
    loss = sum(weighted.values())
    seen = radius > 0
    variables["max_2D_radius"][seen] = torch.max(radius[seen].to(variables["max_2D_radius"].dtype),
                                                 variables["max_2D_radius"][seen])
    variables["seen"] = seen
    weighted["loss"] = loss
    return loss, variables, weighted


class _DirectCtx:
    """Stands in for an autograd context when the Functions' forward / backward are called directly (mapping_iteration)."""
    needs_input_grad = (True,) * 8 + (False,) * 3
    saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


@torch.no_grad()
def mapping_iteration(params, curr_data, variables, iter_time_idx, loss_weights, optimizer, pose7=None):
    """One whole mapping iteration of the reference's loop (src/mapper/splatam/__init__.py:470-480: get_loss, loss.backward(), optimizer.step(),
    optimizer.zero_grad) WITHOUT autograd: the four library calls of the fused path -- per-Gaussian forward on the parameters, render,
    fused loss, backward with the Adam step inside -- are issued one after the other by the very code get_loss(fused=True, fused_loss=True,
    fused_preprocess=True, fused_adam=optimizer) + loss.backward() runs, minus the graph, the engine's thread hand-over and the Function
    boundaries.  At the reference's 256 x 256 frames the iteration is bound by exactly that host time (GPU busy ~230 us, wall ~320 us through
    autograd).  Same parameters, moments and statistics afterwards.  For iterations without a prune / densify event (optim.densify_event /
    prune_event) and the default loss options (use_l1, no outlier rejection, no bundle adjustment).
    -> (loss, variables, {'im', 'depth', 'loss'}); variables['means2D'].grad, ['seen'] and ['max_2D_radius'] are updated as get_loss +
    backward leave them."""
    from . import rasterizer as R
    if pose7 is None:
        q = F.normalize(params["cam_unnorm_rots"][..., iter_time_idx].detach()).reshape(4)
        pose7 = torch.cat([q, params["cam_trans"][..., iter_time_idx].detach().reshape(3)]).cpu().tolist()
    mx = variables["max_2D_radius"]
    means = params["means3D"]
    if not (mx.dtype == torch.float32 and mx.is_contiguous() and mx.device == means.device and mx.numel() == means.shape[0]):
        raise RuntimeError("mapping_iteration: variables['max_2D_radius'] must be a contiguous float32 [N] tensor on the parameters' device")
    seen = torch.empty(mx.numel(), dtype=torch.bool, device=mx.device)
    m2d = torch.empty_like(means, requires_grad=True)          # gradient carrier only: its .grad is what the densifier reads
    shs = params.get("shs")
    colors = None if shs is not None else params["rgb_colors"]
    if shs is not None and int(shs.shape[1]) != 16:
        raise Exception("mapping_iteration: SH rows of 16 coefficients only")
    iso = int(params["log_scales"].shape[1]) == 1
    rctx = _DirectCtx()
    im, radius, depth, _sil, depth_sq = R._RasterizeGaussians.forward(
        rctx, means, m2d, shs, colors, params["logit_opacities"], params["log_scales"], params["unnorm_rotations"], None, curr_data["cam"], True,
        (pose7, iso, False, (mx, seen), optimizer))
    lctx = _DirectCtx()
    loss, parts = _FusedMappingLoss.forward(lctx, im, depth, depth_sq, curr_data["im"], curr_data["depth"], loss_weights["im"], loss_weights["depth"])
    grads = lctx.saved_tensors[0]                               # dL/dim [3,H,W] and dL/ddepth [1,H,W] for dL/dloss = 1
    out = R._RasterizeGaussians.backward(rctx, grads[:3], None, grads[3:], None, None)
    m2d.grad = out[1]
    variables["means2D"] = m2d
    variables["seen"] = seen
    return loss, variables, {"im": parts[1], "depth": parts[2], "loss": loss}


# ---------------------------------------------------------------------------------------------------
# map initialisation / growth (splatam.py:25-115,304-379)
# ---------------------------------------------------------------------------------------------------
def get_pointcloud(color, depth, intrinsics, w2c, transform_pts=True, mask=None, compute_mean_sq_dist=False):
    H, W = color.shape[1], color.shape[2]
    dev = color.device
    fx, fy, cx, cy = intrinsics[0][0], intrinsics[1][1], intrinsics[0][2], intrinsics[1][2]
    xs = (torch.arange(W, device=dev, dtype=torch.float32) - cx) / fx
    ys = (torch.arange(H, device=dev, dtype=torch.float32) - cy) / fy
    xx = xs[None, :].expand(H, W).reshape(-1)
    yy = ys[:, None].expand(H, W).reshape(-1)
    z = depth[0].reshape(-1)
    pts = torch.stack((xx * z, yy * z, z), dim=-1)
    if transform_pts:
        c2w = torch.inverse(torch.as_tensor(w2c, dtype=torch.float32, device=dev))
        pts = pts @ c2w[:3, :3].T + c2w[:3, 3]
    cld = torch.cat((pts, color.permute(1, 2, 0).reshape(-1, 3)), -1)
    msd = (z / ((fx + fy) / 2)) ** 2                         # "projective": farther -> larger radius
    if mask is not None:
        cld, msd = cld[mask], msd[mask]
    return (cld, msd) if compute_mean_sq_dist else cld


def _new_gaussians(pt_cld, mean3_sq_dist, gaussian_distribution):
    n, dev = pt_cld.shape[0], pt_cld.device
    if gaussian_distribution not in ("isotropic", "anisotropic"):
        raise ValueError(f"Unknown gaussian_distribution {gaussian_distribution}")
    ls = torch.log(torch.sqrt(mean3_sq_dist))[:, None]
    rots = torch.zeros(n, 4, device=dev)
    rots[:, 0] = 1.0
    return {"means3D": pt_cld[:, :3], "rgb_colors": pt_cld[:, 3:6], "unnorm_rotations": rots,
            "logit_opacities": torch.zeros(n, 1, device=dev),
            "log_scales": ls if gaussian_distribution == "isotropic" else torch.tile(ls, (1, 3))}


def _as_params(d):
    return {k: torch.nn.Parameter(v.float().contiguous().requires_grad_(True)) for k, v in d.items()}


def initialize_params(init_pt_cld, num_frames, mean3_sq_dist, gaussian_distribution):
    d = _new_gaussians(init_pt_cld, mean3_sq_dist, gaussian_distribution)
    dev = init_pt_cld.device
    cam_rots = torch.zeros(1, 4, num_frames, device=dev)
    cam_rots[:, 0, :] = 1.0
    d["cam_unnorm_rots"] = cam_rots
    d["cam_trans"] = torch.zeros(1, 3, num_frames, device=dev)
    params = _as_params(d)
    n = params["means3D"].shape[0]
    variables = {k: torch.zeros(n, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
    return params, variables


def initialize_new_params(new_pt_cld, mean3_sq_dist, gaussian_distribution):
    return _as_params(_new_gaussians(new_pt_cld, mean3_sq_dist, gaussian_distribution))


def _pose7(params, time_idx):
    q = F.normalize(params["cam_unnorm_rots"][..., time_idx].detach()).reshape(4)
    return torch.cat([q, params["cam_trans"][..., time_idx].detach().reshape(3)]).cpu().tolist()


def _c2w_from_pose7(pose7):
    """Host-side inverse of the frame's w2c = [R(q) | t] (double precision; R as build_rotation defines it)."""
    w, x, y, z = [float(v) for v in pose7[:4]]
    R = np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y],
                  [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                  [2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])
    w2c = np.eye(4)
    w2c[:3, :3], w2c[:3, 3] = R, [float(v) for v in pose7[4:7]]
    return np.linalg.inv(w2c)


def grow_rows(render_depth, silhouette, gt_depth, color, intrinsics, c2w, sil_thres, gaussian_distribution):
    """gs_grow_gaussians (csrc/grow.hip): -> (dict of new per-Gaussian rows, number of non-presence pixels before the
    valid-depth mask).  render_depth / silhouette / gt_depth: [H,W] or [1,H,W]; color [3,H,W]."""
    import ctypes as C
    from . import _lib
    if gaussian_distribution not in ("isotropic", "anisotropic"):
        raise ValueError(f"Unknown gaussian_distribution {gaussian_distribution}")
    lib = _lib.get()
    dev = gt_depth.device
    H, W = int(color.shape[1]), int(color.shape[2])
    c = lambda t: t.detach().contiguous().float()  # noqa: E731
    rd, sil, gt, col = c(render_depth), c(silhouette), c(gt_depth), c(color)
    n = H * W
    iso = gaussian_distribution == "isotropic"
    out = {"means3D": torch.empty(n, 3, device=dev), "rgb_colors": torch.empty(n, 3, device=dev),
           "unnorm_rotations": torch.empty(n, 4, device=dev), "logit_opacities": torch.empty(n, 1, device=dev),
           "log_scales": torch.empty(n, 1 if iso else 3, device=dev)}
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    scratch = torch.empty(int(lib.gs_grow_scratch_bytes(W, H)), dtype=torch.uint8, device=dev)
    K = np.asarray(intrinsics.detach().cpu() if torch.is_tensor(intrinsics) else intrinsics, dtype=np.float64)
    k4 = (C.c_float * 4)(K[0][0], K[1][1], K[0][2], K[1][2])
    c2w = np.asarray(c2w.detach().cpu() if torch.is_tensor(c2w) else c2w, dtype=np.float64)
    m12 = (C.c_float * 12)(*[float(v) for v in c2w[:3, :4].reshape(-1)])
    st = _lib.stream_ptr(dev)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _lib.check(lib.gs_grow_gaussians(W, H, p(rd), p(sil), p(gt), p(col), k4, m12, float(sil_thres), 1 if iso else 0, p(out["means3D"]),
                                     p(out["rgb_colors"]), p(out["unnorm_rotations"]), p(out["logit_opacities"]), p(out["log_scales"]),
                                     p(counts), p(scratch), st))
    n_cand, n_new = [int(v) for v in counts.cpu().tolist()]              # the one host sync of the growth step
    return {k: v[:n_new] for k, v in out.items()}, n_cand


def add_new_gaussians(params, variables, curr_data, sil_thres, time_idx, gaussian_distribution, fused=False, pose7=None):
    """Silhouette / depth-error driven growth (splatam.py:332-379).

    fused=False: the reference op for op (second-pass style [z,1,z^2] render, torch masks, median, boolean gathers).
    fused=True : ONE standard forward (its built-in depth / opacity outputs ARE the depth and silhouette channels,
                 SURVEY 8c-iii) on the fused activations, then gs_grow_gaussians; pose7 = host (qw,qx,qy,qz,tx,ty,tz) of
                 the frame (read from params with one small D2H when omitted)."""
    if fused:
        pose7 = _pose7(params, time_idx) if pose7 is None else pose7
        with torch.no_grad():
            rv = fused_rendervar(params, time_idx, pose7)
            rv["means2D"] = torch.zeros_like(rv["means3D"])
            _, _, render_depth, sil = Renderer(raster_settings=curr_data["cam"])(**rv)
            rows, n_cand = grow_rows(render_depth, sil, curr_data["depth"], curr_data["im"], curr_data["intrinsics"],
                                     _c2w_from_pose7(pose7), sil_thres, gaussian_distribution)
        if n_cand > 0:
            dev = curr_data["depth"].device
            for k, v in rows.items():
                params[k] = torch.nn.Parameter(torch.cat((params[k].detach(), v), dim=0).requires_grad_(True))
            n = params["means3D"].shape[0]
            for k in ("means2D_gradient_accum", "denom", "max_2D_radius"):
                variables[k] = torch.zeros(n, device=dev)
            variables["timestep"] = torch.cat((variables["timestep"], time_idx * torch.ones(rows["means3D"].shape[0], device=dev)), dim=0)
        return params, variables
    """Silhouette / depth-error driven growth (splatam.py:332-379)."""
    tg = transform_to_frame(params, time_idx, gaussians_grad=False, camera_grad=False)
    dsv = transformed_params2depthplussilhouette(params, curr_data["w2c"], tg)
    with torch.no_grad():
        depth_sil, _, _, _ = Renderer(raster_settings=curr_data["cam"])(**dsv)
    sil, render_depth = depth_sil[1], depth_sil[0]
    gt_depth = curr_data["depth"][0]
    depth_error = (gt_depth - render_depth).abs() * (gt_depth > 0)
    behind = (render_depth > gt_depth) & (depth_error > 2 * depth_error.median())
    non_presence = (sil < sil_thres) | (behind & (sil > sil_thres) & (gt_depth < 5))
    non_presence = non_presence.reshape(-1)
    if non_presence.sum() > 0:
        dev = gt_depth.device
        cam_rot = F.normalize(params["cam_unnorm_rots"][..., time_idx].detach())
        curr_w2c = torch.eye(4, device=dev)
        curr_w2c[:3, :3] = build_rotation(cam_rot)
        curr_w2c[:3, 3] = params["cam_trans"][..., time_idx].detach()
        non_presence = non_presence & (gt_depth > 0).reshape(-1)
        new_cld, msd = get_pointcloud(curr_data["im"], curr_data["depth"], curr_data["intrinsics"], curr_w2c,
                                      mask=non_presence, compute_mean_sq_dist=True)
        new_params = initialize_new_params(new_cld, msd, gaussian_distribution)
        for k, v in new_params.items():
            params[k] = torch.nn.Parameter(torch.cat((params[k], v), dim=0).requires_grad_(True))
        n = params["means3D"].shape[0]
        for k in ("means2D_gradient_accum", "denom", "max_2D_radius"):
            variables[k] = torch.zeros(n, device=dev)
        variables["timestep"] = torch.cat((variables["timestep"], time_idx * torch.ones(new_cld.shape[0], device=dev)), dim=0)
    return params, variables
