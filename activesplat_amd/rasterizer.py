"""Drop-in `GaussianRasterizationSettings` / `GaussianRasterizer` backed by the gfx950 HIP library.

Mirrors the Python API the reference imports from the (absent) `diff_gaussian_rasterization`
package -- imports at src/mapper/splatam/splatam.py:22-23, utils/recon_helpers.py:2,
utils/eval_helpers.py:18; settings construction utils/recon_helpers.py:14-27 and
splatam.py:416-429; calls splatam.py:208,212,338,430,431 with the keyword tensors of
utils/slam_helpers.py:131-138.  Contract:

    rasterizer = GaussianRasterizer(raster_settings=cam)            # cheap, constructed per call
    color, radii, depth, opacity = rasterizer(means3D=..., means2D=..., opacities=...,
                                              colors_precomp=... | shs=...,
                                              scales=..., rotations=... | cov3D_precomp=...)

`color` [3,H,W] is differentiable w.r.t. means3D, colors_precomp/shs, opacities, scales, rotations,
cov3D_precomp and the dummy `means2D` [P,3] (NDC-scaled screen-space gradient, read by the reference's
densifier at utils/slam_external.py:100-108).  `radii` [P] int32 (>0 <=> visible), `depth` [1,H,W]
(sum z*alpha*T) and `opacity` [1,H,W] (1 - T_final) are non-differentiable, which is all the reference
needs (they are only read under torch.no_grad(), splatam.py:414,430).

All compute is in libgsplat_hip.so through the C ABI of include/gsplat_hip.h; there is no fallback.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
from typing import NamedTuple

import torch
from torch import nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


#: statistics of the most recent forward on this process (read by bench.py / tests).  Process-global and unsynchronised: with forwards
#: in flight on several threads it holds the numbers of whichever finished last -- informational, nothing in the product path reads it
last_stats = {"num_rendered": 0, "P": 0}
#: per (P, W, H, device): (pair capacity, tile-list capacity) guessed for the next frame's optimistic launch
_capacity = {}
#: set False to always size the binning workspace from the exact counts (one mid-pipeline host sync per forward)
optimistic = True

#: pinned host counter pairs (D, largest tile list), one per (thread, device, stream): the scan kernel stores straight into
#: them, so two forwards in flight -- another stream of a keyframe batch, the planner / visualiser thread of the reference's
#: threaded layout (SURVEY section 8b "Threading") -- must never share one
_tls = threading.local()
_capacity_lock = threading.Lock()


@contextlib.contextmanager
def capture():
    """`with capture() as state:` -- the state buffers of the forwards run inside the block by THIS thread (the last one wins) are put
    into `state` (geom / image / binning workspaces, point_list, their layouts, D, P, W, H; layouts in include/gsplat_hip.h) so that
    tests and bench.py can decode the integer artefacts.  Nothing is retained outside a capture block: the reference's `debug=True`
    settings flag alone does not pin any buffer."""
    stack = getattr(_tls, "captures", None)
    if stack is None:
        stack = _tls.captures = []
    state = {}
    stack.append(state)
    try:
        yield state
    finally:
        stack.remove(state)


def _require_rocm(device):
    """The product path takes ROCm device tensors only (no CPU fallback)."""
    if device.type != "cuda":
        raise RuntimeError("activesplat_amd rasteriser needs ROCm device tensors (no CPU fallback)")


def _host_counters(device, stream_handle):
    pool = getattr(_tls, "pinned", None)
    if pool is None:
        pool = _tls.pinned = {}
    key = (device.index, stream_handle)
    buf = pool.get(key)
    if buf is None:
        if len(pool) >= 64:                              # short-lived streams: do not grow without bound
            pool.pop(next(iter(pool)))
        buf = pool[key] = torch.zeros(2, dtype=torch.int32).pin_memory()
    return buf


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32(t, device):
    if t is None:
        return None
    if t.dtype != torch.float32 or t.device != device or not t.is_contiguous():
        t = t.to(device=device, dtype=torch.float32).contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _stream(device):
    return _lib.stream_ptr(device)


#: the two size-only layouts of a (P, W, H) frame: asked from the library once, not per frame
_layouts = {}


def _frame_layouts(lib, P, W, H):
    key = (P, W, H)
    hit = _layouts.get(key)
    if hit is None:
        gl = _lib.GsGeomLayout(); _lib.check(lib.gs_geom_layout(P, W, H, C.byref(gl)))
        il = _lib.GsImageLayout(); _lib.check(lib.gs_image_layout(W, H, C.byref(il)))
        if len(_layouts) >= 64:                          # P changes with every densify / growth step
            _layouts.pop(next(iter(_layouts)))
        hit = _layouts[key] = (gl, il, int(lib.gs_backward_scratch_bytes(P)))
    return hit


def _camera(rs: GaussianRasterizationSettings, device, sh_coeffs: int):
    keep = dict(bg=_f32(rs.bg, device).reshape(-1), view=_f32(rs.viewmatrix, device).reshape(-1),
                proj=_f32(rs.projmatrix, device).reshape(-1), campos=_f32(rs.campos, device).reshape(-1))
    if keep["view"].numel() != 16 or keep["proj"].numel() != 16 or keep["bg"].numel() != 3:
        raise Exception("GaussianRasterizationSettings: viewmatrix/projmatrix must hold 16 values, bg 3")
    cam = _lib.GsCamera(int(rs.image_width), int(rs.image_height), int(rs.sh_degree), int(sh_coeffs),
                        float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier), 0,
                        keep["bg"].data_ptr(), keep["view"].data_ptr(), keep["proj"].data_ptr(),
                        keep["campos"].data_ptr())
    return cam, keep


class _RasterizeGaussians(torch.autograd.Function):
    """fused=False: the reference contract (color differentiable; radii, depth, opacity not).
    fused=True : one pass also yields the reference's SECOND raster pass -- depth (differentiable), silhouette
                 (= opacity) and depth^2 -- see render_rgbd()."""

    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs, fused=False, raw=None):
        """raw = (pose7, isotropic, accumulate): means3D / opacities / scales / rotations are the mapper's PARAMETERS (world-frame means, logit
        opacities, log scales, unnormalised quaternions); frame transform + activations run inside the per-Gaussian kernels (render_rgbd_raw)."""
        lib = _lib.get()
        device = means3D.device
        leaves = (means3D, opacities, scales, rotations, colors_precomp)     # (raw + accumulate: the backward adds into these tensors' .grad)
        shs_in = shs
        _require_rocm(device)
        _lib.poll_async_status()                 # (a plain host load: did a chained backward of an earlier launch give up its wait?)
        P = int(means3D.shape[0])
        means3D = _f32(means3D, device)
        shs, colors_precomp = _f32(shs, device), _f32(colors_precomp, device)
        opacities, scales = _f32(opacities, device), _f32(scales, device)
        rotations, cov3D_precomp = _f32(rotations, device), _f32(cov3D_precomp, device)
        M = 0 if shs is None else int(shs.shape[1])
        cam, keep = _camera(rs, device, M)
        W, H = int(rs.image_width), int(rs.image_height)
        cur = torch.cuda.current_stream(device) if device.type == "cuda" else None     # ONE lookup per call (~6 us each)
        st_handle = int(cur.cuda_stream) if cur is not None else 0
        st = C.c_void_p(st_handle)

        gl, il, scratch_bytes = _frame_layouts(lib, P, W, H)
        geom = torch.empty(gl.total_bytes, dtype=torch.uint8, device=device)
        image = torch.empty(il.total_bytes, dtype=torch.uint8, device=device)
        radii = torch.empty(P, dtype=torch.int32, device=device)
        d_num = torch.empty(2, dtype=torch.int32, device=device)
        h_num = _host_counters(device, st_handle) if device.type == "cuda" else torch.zeros(2, dtype=torch.int32)
        want_bwd = 1 if any(ctx.needs_input_grad[:8]) else 0
        pose = None
        if raw is not None:
            pose = (C.c_float * 7)(*[float(v) for v in raw[0]])
            vis_max, vis_seen = raw[3] if len(raw) > 3 and raw[3] is not None else (None, None)
            _lib.check(lib.gs_preprocess_forward_raw(C.byref(cam), P, _ptr(means3D), _ptr(shs), _ptr(colors_precomp), _ptr(opacities),
                                                     _ptr(scales), _ptr(rotations), pose, 1 if raw[1] else 0, _ptr(vis_max), _ptr(vis_seen),
                                                     _ptr(radii), _ptr(geom), _ptr(image), _ptr(d_num), _ptr(h_num), want_bwd, st))
        else:
            _lib.check(lib.gs_preprocess_forward(C.byref(cam), P, _ptr(means3D), _ptr(shs), _ptr(colors_precomp),
                                                 _ptr(opacities), _ptr(scales), _ptr(rotations), _ptr(cov3D_precomp),
                                                 _ptr(radii), _ptr(geom), _ptr(image), _ptr(d_num), _ptr(h_num), want_bwd, st))
        color = torch.empty(3, H, W, dtype=torch.float32, device=device)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=device)
        opacity = torch.empty(1, H, W, dtype=torch.float32, device=device)
        depth_sq = torch.empty(1, H, W, dtype=torch.float32, device=device) if fused else None
        # the backward's gradient records: allocated now (when a backward can follow) so that the blend kernel of this
        # forward zero-fills them as a side job instead of a fill launch in front of the backward
        scratch = None
        if any(ctx.needs_input_grad[:8]) and P > 0:
            scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=device)

        def render(cap_d, cap_tile):
            bl_ = _lib.GsBinLayout(); _lib.check(lib.gs_bin_layout(cap_d, cap_tile, W, H, C.byref(bl_)))
            binning_ = torch.empty(bl_.total_bytes, dtype=torch.uint8, device=device)
            plist_ = torch.empty(max(cap_d, 1), dtype=torch.int32, device=device)
            _lib.check(lib.gs_render_forward(C.byref(cam), P, cap_d, cap_tile, _ptr(geom), _ptr(binning_), _ptr(plist_), _ptr(image),
                                             _ptr(color), _ptr(depth), _ptr(opacity), _ptr(depth_sq), _ptr(scratch), st))
            return bl_, binning_, plist_

        # Optimistic launch: the binning workspace is sized from the previous frame of this (P, W, H) stream (+25 %), the
        # whole render is enqueued BEHIND the counting kernels, and only then does the host wait for the two counters --
        # the GPU keeps working through what used to be an idle gap (host wake-up + allocation + launch).  A frame whose
        # true counts exceed the guess is re-launched with exact sizes (gs_render_forward is capacity-safe and idempotent).
        done = None
        key = (P, W, H, device.index)
        with _capacity_lock:
            guess = _capacity.get(key) if optimistic else None
        if guess is not None:
            if device.type == "cuda":
                ev = torch.cuda.Event()
                ev.record(cur)
            bl_g = _lib.GsBinLayout(); _lib.check(lib.gs_bin_layout(guess[0], guess[1], W, H, C.byref(bl_g)))
            # (never optimistic where the capacities would CHOOSE the algorithm: the segmented compositing of few-tile images is switched on, and
            # its segment count set, by the tile-list bound -- an inflated guess would make the image depend on the call history; such frames
            # take the exact launch, whose choice follows the true counts)
            if bl_g.path == 1 and bl_g.segments <= 1:
                done = render(*guess)
            if device.type == "cuda":
                ev.synchronize()                                # counters are on the host; the render is still in flight
        elif device.type == "cuda":
            cur.synchronize()                                    # first frame of a stream: D sizes the binning buffers
        D = int(h_num[0].item()) & 0xFFFFFFFF
        max_tile = int(h_num[1].item()) & 0xFFFFFFFF
        if done is not None and D <= guess[0] and max_tile <= guess[1]:
            bl, binning, point_list = done
            last_stats["optimistic_hits"] = last_stats.get("optimistic_hits", 0) + 1
        else:
            bl, binning, point_list = render(D, max_tile)
            last_stats["optimistic_misses"] = last_stats.get("optimistic_misses", 0) + (1 if guess is not None else 0)
        with _capacity_lock:
            old = _capacity.get(key, (0, 0))                    # monotone: views that alternate settle on the largest
            if len(_capacity) >= 64 and key not in _capacity:   # P changes with every densify / growth step: keep the table small
                _capacity.pop(next(iter(_capacity)))
            # (the tile-list bound only selects kernels and LDS variants: a tight margin keeps a 2 M-Gaussian frame -- lists of up to 4.8 k keys --
            # inside the one-workgroup bucket sort's 5632-key class; a list that outgrows it costs one exact re-launch)
            _capacity[key] = (max(old[0], int(D * 1.25) + 4096), max(old[1], max_tile + max_tile // 16 + 64))
        last_stats["num_rendered"], last_stats["P"], last_stats["max_tile_instances"] = D, P, max_tile
        ctx.rs, ctx.D, ctx.keep, ctx.fused, ctx.cam = rs, D, keep, fused, cam      # the backward reuses the camera block
        ctx.scratch, ctx.scratch_clean, ctx.sh_jac = scratch, scratch is not None, want_bwd
        ctx.has = (shs is not None, colors_precomp is not None, scales is not None, rotations is not None,
                   cov3D_precomp is not None)
        ctx.raw = None if raw is None else (pose, 1 if raw[1] else 0, bool(raw[2]), opacities)
        # in-kernel accumulation only into the very tensors the caller passed (a converted copy has no .grad to add to)
        adam = raw[4] if raw is not None and len(raw) > 4 else None
        same = raw is not None and (raw[2] or adam is not None) and all(a is b for a, b in zip(leaves, (means3D, opacities, scales, rotations, colors_precomp)))
        ctx.leaves = leaves if same else None
        ctx.adam = None
        if adam is not None:
            # the optimiser step inside the backward kernel: it updates the caller's parameter tensors in place
            if raw[2]:
                raise Exception("render_rgbd_raw: adam= and accumulate_grads= exclude each other")
            if not same or (shs is not None and shs is not shs_in):
                raise Exception("render_rgbd_raw(adam=...): the parameters must be contiguous, 16-byte aligned fp32 tensors on the device (they are updated in place)")
            ctx.adam = (adam, leaves[:4] + (shs_in if shs_in is not None else leaves[4],))
        caps = getattr(_tls, "captures", None)
        if caps:
            caps[-1].update(geom=geom, image=image, binning=binning, point_list=point_list, gl=gl, il=il, bl=bl, D=D, P=P, W=W, H=H)
        e = torch.empty(0, device=device)
        ctx.save_for_backward(means3D, shs if shs is not None else e, colors_precomp if colors_precomp is not None else e,
                              scales if scales is not None else e, rotations if rotations is not None else e,
                              cov3D_precomp if cov3D_precomp is not None else e, radii, geom, point_list, image)
        ctx.set_materialize_grads(False)          # no zero-filled grad tensors for the non-differentiable outputs
        if fused:
            ctx.mark_non_differentiable(radii, opacity, depth_sq)
            return color, radii, depth, opacity, depth_sq
        ctx.mark_non_differentiable(radii, depth, opacity)
        return color, radii, depth, opacity

    @staticmethod
    def backward(ctx, grad_color, _gr=None, grad_depth=None, _go=None, _gq=None):
        lib = _lib.get()
        means3D, shs, colors, scales, rots, cov3Dp, radii, geom, point_list, image = ctx.saved_tensors
        has_sh, has_col, has_sc, has_rot, has_cov = ctx.has
        device = means3D.device
        P = int(means3D.shape[0])
        M = int(shs.shape[1]) if has_sh else 0
        cam = ctx.cam                                    # its device pointers are kept alive by ctx.keep
        if grad_color is None:
            grad_color = torch.zeros(3, int(ctx.rs.image_height), int(ctx.rs.image_width), device=device)
        grad_color = _f32(grad_color, device)
        grad_depth = _f32(grad_depth, device) if (ctx.fused and grad_depth is not None) else None
        scratch, clean = ctx.scratch, ctx.scratch_clean
        if scratch is None:
            scratch, clean = torch.empty(int(lib.gs_backward_scratch_bytes(P)), dtype=torch.uint8, device=device), False
        ctx.scratch, ctx.scratch_clean = None, False     # released with this launch (64 B x P); a second backward through the same graph
                                                         # takes a fresh buffer plus a memset
        if ctx.raw is not None and ctx.adam is not None:
            # single-keyframe step: Adam on the five per-Gaussian tensors rides in the per-Gaussian backward kernel (no gradient tensors)
            if ctx.adam == "stepped":
                # (retain_graph + a second backward through this render would apply the optimiser step twice, on already-stepped parameters)
                raise RuntimeError("render_rgbd_raw(adam=...): this render's backward has already applied its optimiser step; a second backward "
                                   "through the same graph is refused -- render again, or use accumulate_grads / a separate optimizer.step()")
            pose, iso, _acc, logit = ctx.raw
            d_m2d = torch.empty(P, 3, dtype=torch.float32, device=device)
            opt, tensors = ctx.adam
            desc = opt.backward_step_descriptors(tensors)                 # (validates, then advances the step counters ...; a refusal here
            ctx.adam = "stepped"                                          # leaves counters AND this graph as they were: the caller may retry)
            try:
                _lib.check(lib.gs_render_backward_raw_adam(
                    C.byref(cam), P, ctx.D, _ptr(means3D), _ptr(shs if has_sh else None), _ptr(colors if has_col else None), _ptr(logit),
                    _ptr(scales), _ptr(rots), pose, iso, _ptr(radii), _ptr(geom), _ptr(point_list), _ptr(image), _ptr(grad_color),
                    _ptr(grad_depth), _ptr(d_m2d), _ptr(scratch), 1 if clean else 0, int(ctx.sh_jac), desc, _stream(device)))
            except Exception:
                opt.rollback_backward_step(tensors)                       # (... which a launch that did not happen must not keep)
                ctx.adam = (opt, tensors)
                raise
            # the parameters changed in place behind autograd's back: bump their version counters, so that any other graph that saved them
            # (a regulariser on the same parameters, a second keyframe rendered before this backward) fails loudly in ITS backward instead of
            # differentiating through stepped values -- and note that such a branch gets no rasteriser gradient from this render
            torch.autograd.graph.increment_version([t for t in tensors if t is not None])
            return None, d_m2d, None, None, None, None, None, None, None, None, None
        z = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)  # noqa: E731  (kernel writes every row)
        d_m2d, d_m3d, d_op = z(P, 3), z(P, 3), z(P, 1)
        d_col = z(P, 3) if has_col else None
        d_shs = z(P, M, 3) if has_sh else None
        d_sc = z(P, 3) if has_sc else None
        d_rot = z(P, 4) if has_rot else None
        d_cov = z(P, 6) if has_cov else None
        if ctx.raw is not None:
            pose, iso, accumulate, logit = ctx.raw
            d_sc = z(P, 1 if iso else 3)                  # (gradients w.r.t. the parameters: log scales are [P,1] for an isotropic map)
            # accumulate: add into the leaves' .grad inside the kernel (what autograd would do with the results in separate passes);
            # leaves = (means3D, logit opacities, log scales, rotations, colours)
            leaves, acc, into = ctx.leaves, 0, None
            if leaves is not None:
                live = [x for x in leaves if x is not None]
                ok = all(x.is_leaf and x.requires_grad for x in live)
                have = [x.grad is not None and x.grad.is_contiguous() and x.grad.dtype == torch.float32 and x.grad.shape == x.shape for x in live]
                if ok and all(have):
                    acc, into = 1, [None if x is None else x.grad for x in leaves]
                elif ok and not any(x.grad is not None for x in live):
                    into = [None if x is None else torch.empty_like(x, memory_format=torch.contiguous_format) for x in leaves]
            if into is not None:
                d_m3d, d_op, d_sc, d_rot = into[0], into[1], into[2], into[3]
                if has_col:
                    d_col = into[4]
            _lib.check(lib.gs_render_backward_raw(
                C.byref(cam), P, ctx.D, _ptr(means3D), _ptr(shs if has_sh else None), _ptr(colors if has_col else None), _ptr(logit),
                _ptr(scales), _ptr(rots), pose, iso, acc, _ptr(radii), _ptr(geom), _ptr(point_list), _ptr(image), _ptr(grad_color),
                _ptr(grad_depth), _ptr(d_m2d), _ptr(d_m3d), _ptr(d_op), _ptr(d_col), _ptr(d_shs), _ptr(d_sc), _ptr(d_rot),
                _ptr(scratch), 1 if clean else 0, int(ctx.sh_jac), _stream(device)))
            if into is not None:
                if not acc:
                    for x, t in zip(leaves, into):
                        if x is not None:
                            x.grad = t
                return None, d_m2d, d_shs, None, None, None, None, None, None, None, None
            return d_m3d, d_m2d, d_shs, d_col, d_op, d_sc, d_rot, None, None, None, None
        _lib.check(lib.gs_render_backward(
            C.byref(cam), P, ctx.D, _ptr(means3D), _ptr(shs if has_sh else None), _ptr(colors if has_col else None),
            _ptr(scales if has_sc else None), _ptr(rots if has_rot else None), _ptr(cov3Dp if has_cov else None),
            _ptr(radii), _ptr(geom), _ptr(point_list), _ptr(image), _ptr(grad_color), _ptr(grad_depth),
            _ptr(d_m2d), _ptr(d_m3d), _ptr(d_op), _ptr(d_col), _ptr(d_shs), _ptr(d_sc), _ptr(d_rot), _ptr(d_cov),
            _ptr(scratch), 1 if clean else 0, int(ctx.sh_jac), _stream(device)))
        return d_m3d, d_m2d, d_shs, d_col, d_op, d_sc, d_rot, d_cov, None, None, None


#: the drop-in call goes through the C++ autograd front-end (csrc/torch_frontend.cpp: the same library calls in the same order as
#: _RasterizeGaussians above, without the interpreter in the way -- host time of a forward + backward ~240 -> ~90 us, which is what makes the reference's own
#: 256 x 256 frames GPU-bound).  False: the Python twin (A/B measurements, tests of the twin).  Inside a capture() block and on the
#: host-emulated test build the Python twin runs regardless.
use_frontend = True


def _frontend_apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs, fused):
    """The drop-in forward through the C++ front-end; None when this call is not one for it (then the Python twin takes it)."""
    if not use_frontend or _lib.emulated() or means3D.device.type != "cuda":
        return None
    caps = getattr(_tls, "captures", None)
    from . import _frontend
    _lib.get()                                          # (the HIP library itself missing: raises -- there is no CPU path)
    ext = _frontend.get_or_none()                       # (no g++ / torch headers on this box: ONE warning, then the Python twin -- the same library calls)
    if ext is None:
        return None
    device = means3D.device
    P, W, H = int(means3D.shape[0]), int(rs.image_width), int(rs.image_height)
    key = (P, W, H, device.index)
    with _capacity_lock:
        guess = _capacity.get(key) if optimistic else None
    out, D, max_tile, hit, state, off = ext.rasterize(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs.bg,
                                                      rs.viewmatrix, rs.projmatrix, rs.campos, W, H, float(rs.tanfovx), float(rs.tanfovy),
                                                      float(rs.scale_modifier), int(rs.sh_degree), bool(fused), guess[0] if guess else 0,
                                                      guess[1] if guess else 0, bool(caps))
    if caps:
        # capture(): views of the one workspace (and of the exact re-render's buffers after a miss), under the names the Python twin publishes
        lib = _lib.get()
        gl, il, _ = _frame_layouts(lib, P, W, H)
        ws, binning, plist2 = state
        cap_d, cap_tile = (guess if hit else (D, max_tile))
        bl = _lib.GsBinLayout(); _lib.check(lib.gs_bin_layout(cap_d, cap_tile, W, H, C.byref(bl)))
        caps[-1].update(geom=ws[off[0]:off[0] + off[1]], image=ws[off[2]:off[2] + off[3]], gl=gl, il=il, bl=bl, D=D, P=P, W=W, H=H, binning=binning,
                        point_list=ws[off[4]:off[4] + 4 * off[5]].view(torch.int32) if hit else plist2)
    if hit:
        last_stats["optimistic_hits"] = last_stats.get("optimistic_hits", 0) + 1
    else:
        last_stats["optimistic_misses"] = last_stats.get("optimistic_misses", 0) + (1 if guess is not None else 0)
    with _capacity_lock:
        old = _capacity.get(key, (0, 0))
        if len(_capacity) >= 64 and key not in _capacity:
            _capacity.pop(next(iter(_capacity)))
        _capacity[key] = (max(old[0], int(D * 1.25) + 4096), max(old[1], max_tile + max_tile // 16 + 64))
    last_stats["num_rendered"], last_stats["P"], last_stats["max_tile_instances"] = D, P, max_tile
    return tuple(out)


def rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                        raster_settings):
    out = _frontend_apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, raster_settings, False)
    if out is not None:
        return out
    return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                     cov3D_precomp, raster_settings)


def render_rgbd(raster_settings, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
    """Single-pass RGB + depth + silhouette render -- an ADDITIONAL entry point, the
    drop-in GaussianRasterizer is unchanged.

    The reference rasterises every training iteration twice on identical geometry: once for RGB and once with
    per-Gaussian "colours" [z_cam, 1, z_cam^2] (src/mapper/splatam/splatam.py:208,212 with
    utils/slam_helpers.py:196-249).  With the settings' viewmatrix equal to the w2c used for z_cam (true at every
    reference call site) those three channels are sum z*alpha*T, sum alpha*T and sum z^2*alpha*T of the SAME blend,
    so one pass returns
        color [3,H,W], radii [P], depth [1,H,W], silhouette [1,H,W], depth_sq [1,H,W]
    with `color` AND `depth` differentiable (the mapping loss back-propagates through both; silhouette and
    depth_sq are only used as masks / detached, splatam.py:213-231).  The depth gradient reaches means3D through
    the per-Gaussian view depth exactly as in the two-pass formulation."""
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    out = _frontend_apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, raster_settings, True)
    if out is not None:
        return out
    return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                     raster_settings, True)


def render_rgbd_raw(raster_settings, means3D, means2D, logit_opacities, log_scales, unnorm_rotations, pose7, shs=None, colors_precomp=None,
                    accumulate_grads=False, visibility=None, adam=None):
    """render_rgbd straight from the mapper's PARAMETERS: the frame transform + activations of transform_to_frame /
    transformed_params2rendervar (slam_helpers.py:252-304,124-139; `mapping.fused_rendervar` does them in two launches of their own) happen
    inside the per-Gaussian kernels of the rasteriser, forward and backward.  pose7 = host (qw,qx,qy,qz,tx,ty,tz) of the frame's relative
    w2c (camera quaternion normalised); log_scales [P,1] = isotropic map.  Colours given, or 16-coefficient SH rows.
    accumulate_grads (for `loss.backward()` over a batch of keyframes): the backward ADDS the gradients of means3D, logit_opacities,
    log_scales, unnorm_rotations and colors_precomp to those tensors' .grad inside its kernel and hands autograd nothing for them.
    visibility = (max_2D_radius [P] float32, seen [P] bool), both contiguous on the device: the mapper's statistics of this render
    (splatam.py:296-298: max_2D_radius = max(max_2D_radius, radius) in place, seen = radius > 0) written by the forward kernel itself.
    adam = the optim.GaussianAdam that owns the five tensors (single-keyframe steps -- the reference's loss.backward(); optimizer.step() on one
    frame, src/mapper/splatam/__init__.py:470-480): the backward kernel that forms a Gaussian's parameter gradients applies the Adam update
    to parameters and moments IN PLACE (gs_render_backward_raw_adam; same arithmetic as optimizer.step(), bit for bit) and writes no
    gradient tensors -- the five tensors keep .grad = None, a following optimizer.step() skips them; means2D.grad is delivered as usual.
    Not for gradient accumulation over several keyframes, and not on iterations whose densify / prune event replaces the tensors between
    backward and step (optim.densify_event)."""
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if shs is not None and int(shs.shape[1]) != 16:
        raise Exception("render_rgbd_raw: SH rows of 16 coefficients only")
    iso = int(log_scales.shape[1]) == 1
    if visibility is not None:
        mx, seen = visibility
        P = int(means3D.shape[0])
        if not (mx.dtype == torch.float32 and mx.is_contiguous() and mx.numel() == P and seen.dtype == torch.bool and seen.is_contiguous()
                and seen.numel() == P and mx.device == means3D.device and seen.device == means3D.device):
            raise Exception("render_rgbd_raw: visibility = (float32 [P], bool [P]) contiguous tensors on the parameters' device")
    return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, logit_opacities, log_scales, unnorm_rotations, None,
                                     raster_settings, True, (pose7, iso, accumulate_grads, visibility, adam))


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs)


@torch.no_grad()
def render_views(settings_list, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                 return_atlas=False):
    """Forward-only render of V views of ONE set of Gaussians in a single pass (the planner's look-around panoramas,
    src/mapper/splatam/__init__.py:707-736, 765-778: three 120 x 150 views per node).  The views share size, fov, background,
    scale modifier and SH degree and differ in their view / projection matrices and camera centre.  The per-Gaussian stage runs
    once over V x P virtual Gaussians; binning, sorting and blending see ONE atlas image (GsCamera.num_views, gs_atlas_layout) --
    one set of launches and one host read of the counters instead of V.
    -> list of (color [3,H,W], radii [P] int32, depth [1,H,W], opacity [1,H,W]) per view (views of the atlas tensors; `radii` is the
       view's first P rows of the padded virtual layout -- each view owns ceil(P/256)*256 rows, the padding rows are never returned);
    return_atlas=True: (color [3,H,AW], depth [1,H,AW], opacity [1,H,AW], view_stride) -- view v is columns [v stride, v stride + W).
    Settings whose tensors live on the host (setup_camera(device="cpu")) cost ONE host-to-device copy for all views."""
    lib = _lib.get()
    V = len(settings_list)
    rs0 = settings_list[0]
    if V == 1:
        one = GaussianRasterizer(rs0)(means3D=means3D, means2D=None, opacities=opacities, shs=shs, colors_precomp=colors_precomp,
                                      scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
        return (one[0], one[2], one[3], int(rs0.image_width)) if return_atlas else [one]
    bg0 = [float(x) for x in rs0.bg.reshape(-1).tolist()]
    for rs in settings_list[1:]:
        if (rs.image_width, rs.image_height, rs.tanfovx, rs.tanfovy, rs.scale_modifier, rs.sh_degree) != \
                (rs0.image_width, rs0.image_height, rs0.tanfovx, rs0.tanfovy, rs0.scale_modifier, rs0.sh_degree) or \
                (rs.bg is not rs0.bg and [float(x) for x in rs.bg.reshape(-1).tolist()] != bg0):
            raise Exception("render_views: the views must share size, field of view, scale modifier, SH degree and background")
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    device = means3D.device
    _require_rocm(device)
    P, W, H = int(means3D.shape[0]), int(rs0.image_width), int(rs0.image_height)
    means3D, shs, colors_precomp = _f32(means3D, device), _f32(shs, device), _f32(colors_precomp, device)
    opacities, scales, rotations, cov3D_precomp = _f32(opacities, device), _f32(scales, device), _f32(rotations, device), _f32(cov3D_precomp, device)
    M = 0 if shs is None else int(shs.shape[1])
    if all(not rs.viewmatrix.is_cuda for rs in settings_list) and device.type == "cuda":
        host = torch.cat([torch.stack([rs.viewmatrix.reshape(16) for rs in settings_list]).reshape(-1),
                          torch.stack([rs.projmatrix.reshape(16) for rs in settings_list]).reshape(-1),
                          torch.stack([rs.campos.reshape(3) for rs in settings_list]).reshape(-1), rs0.bg.reshape(3).cpu()]).float()
        pack = host.to(device, non_blocking=True)
        keep = dict(pack=pack, view=pack[:16 * V], proj=pack[16 * V:32 * V], campos=pack[32 * V:35 * V], bg=pack[35 * V:35 * V + 3])
    else:
        keep = dict(bg=_f32(rs0.bg, device).reshape(-1),
                    view=torch.stack([_f32(rs.viewmatrix, device).reshape(16) for rs in settings_list]).contiguous(),
                    proj=torch.stack([_f32(rs.projmatrix, device).reshape(16) for rs in settings_list]).contiguous(),
                    campos=torch.stack([_f32(rs.campos, device).reshape(3) for rs in settings_list]).contiguous())
    cam = _lib.GsCamera(W, H, int(rs0.sh_degree), M, float(rs0.tanfovx), float(rs0.tanfovy), float(rs0.scale_modifier), V,
                        keep["bg"].data_ptr(), keep["view"].data_ptr(), keep["proj"].data_ptr(), keep["campos"].data_ptr())
    pv, aw, stride = C.c_int32(), C.c_int32(), C.c_int32()
    _lib.check(lib.gs_atlas_layout(P, W, V, C.byref(pv), C.byref(aw), C.byref(stride)))
    Pv, AW, S = pv.value, aw.value, stride.value
    st_handle = _lib.stream_handle(device)
    st = C.c_void_p(st_handle)
    gl = _lib.GsGeomLayout(); _lib.check(lib.gs_geom_layout(Pv, AW, H, C.byref(gl)))
    il = _lib.GsImageLayout(); _lib.check(lib.gs_image_layout(AW, H, C.byref(il)))
    geom = torch.empty(gl.total_bytes, dtype=torch.uint8, device=device)
    image = torch.empty(il.total_bytes, dtype=torch.uint8, device=device)
    radii = torch.empty(max(Pv, 1), dtype=torch.int32, device=device)
    d_num = torch.empty(2, dtype=torch.int32, device=device)
    h_num = _host_counters(device, st_handle) if device.type == "cuda" else torch.zeros(2, dtype=torch.int32)
    _lib.check(lib.gs_preprocess_forward(C.byref(cam), P, _ptr(means3D), _ptr(shs), _ptr(colors_precomp), _ptr(opacities), _ptr(scales),
                                         _ptr(rotations), _ptr(cov3D_precomp), _ptr(radii), _ptr(geom), _ptr(image), _ptr(d_num), _ptr(h_num), 0, st))
    if device.type == "cuda":
        torch.cuda.current_stream(device).synchronize()      # D and the longest tile list size the binning workspace
    D, max_tile = int(h_num[0].item()) & 0xFFFFFFFF, int(h_num[1].item()) & 0xFFFFFFFF
    bl = _lib.GsBinLayout(); _lib.check(lib.gs_bin_layout(D, max_tile, AW, H, C.byref(bl)))
    binning = torch.empty(bl.total_bytes, dtype=torch.uint8, device=device)
    plist = torch.empty(max(D, 1), dtype=torch.int32, device=device)
    color = torch.empty(3, H, AW, dtype=torch.float32, device=device)
    depth = torch.empty(1, H, AW, dtype=torch.float32, device=device)
    opacity = torch.empty(1, H, AW, dtype=torch.float32, device=device)
    _lib.check(lib.gs_render_forward(C.byref(cam), P, D, max_tile, _ptr(geom), _ptr(binning), _ptr(plist), _ptr(image),
                                     _ptr(color), _ptr(depth), _ptr(opacity), None, None, st))
    last_stats["num_rendered"], last_stats["P"], last_stats["max_tile_instances"] = D, P, max_tile
    if return_atlas:
        return color, depth, opacity, S
    rows = Pv // V
    return [(color[:, :, v * S:v * S + W], radii[v * rows:v * rows + P], depth[:, :, v * S:v * S + W], opacity[:, :, v * S:v * S + W])
            for v in range(V)]
