"""Seeded synthetic scenes for benchmarks and parity tests.

Gaussians are generated on the CPU with torch.Generator().manual_seed(seed) exactly as section 8d
prescribes: pixel u~U[0,W), v~U[0,H), depth z~U[0.5,4]; means = ((u-cx)/fx z, (v-cy)/fy z, z);
log_scales = log(z/fx) + N(0,0.3^2); unnorm_rotations ~ N(0,1)^4; logit_opacities ~ N(0,1);
rgb ~ U[0,1]^3.  Activations are those of the reference's transformed_params2rendervar
(src/mapper/splatam/utils/slam_helpers.py:124-139): normalize / sigmoid / exp.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def intrinsics(W=640, H=480, fx=None, fy=None):
    fx = W / 2.0 if fx is None else fx          # hfov 90 deg
    fy = fx if fy is None else fy
    return np.array([[fx, 0.0, W / 2.0 - 1.0], [0.0, fy, H / 2.0 - 1.0], [0.0, 0.0, 1.0]], dtype=np.float64)


def make_params(N, W=640, H=480, seed=0, K=None, sh_degree=None, scale_jitter=0.3, zmin=0.5, zmax=4.0):
    """Raw (un-activated) parameter dict with the reference's key names (splatam.py:89-113)."""
    K = intrinsics(W, H) if K is None else K
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(N, generator=g) * W
    v = torch.rand(N, generator=g) * H
    z = zmin + torch.rand(N, generator=g) * (zmax - zmin)
    means = torch.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    log_scales = torch.log(z / fx)[:, None] + scale_jitter * torch.randn(N, 3, generator=g)
    rots = torch.randn(N, 4, generator=g)
    logit_op = torch.randn(N, 1, generator=g)
    rgb = torch.rand(N, 3, generator=g)
    p = dict(means3D=means.float(), rgb_colors=rgb.float(), unnorm_rotations=rots.float(),
             logit_opacities=logit_op.float(), log_scales=log_scales.float())
    if sh_degree is not None:
        M = 16
        shs = 0.2 * torch.randn(N, M, 3, generator=g)
        shs[:, 0, :] = (rgb - 0.5) / 0.28209479177387814
        p["shs"] = shs.float()
    return p


def activate(params):
    """rendervar dict for GaussianRasterizer(**rendervar) (slam_helpers.py:124-139)."""
    rv = dict(means3D=params["means3D"],
              rotations=torch.nn.functional.normalize(params["unnorm_rotations"]),
              opacities=torch.sigmoid(params["logit_opacities"]),
              scales=torch.exp(params["log_scales"]))
    if "shs" in params:
        rv["shs"] = params["shs"]
    else:
        rv["colors_precomp"] = params["rgb_colors"]
    return rv


def make_targets(W=640, H=480, seed=1):
    g = torch.Generator().manual_seed(seed)
    im = torch.rand(3, H, W, generator=g)
    depth = 0.5 + 3.5 * torch.rand(1, H, W, generator=g)
    return im.float(), depth.float()


def shell_scene(N, seed=0, W=640, H=480):
    """C4 scene: Gaussians on a shell around the origin; keyframe i looks along yaw 2*pi*i/K."""
    g = torch.Generator().manual_seed(seed)
    yaw = torch.rand(N, generator=g) * 2 * math.pi
    pitch = (torch.rand(N, generator=g) * 70.0 - 35.0) * math.pi / 180.0
    rng = 0.5 + 3.5 * torch.rand(N, generator=g)
    means = torch.stack([rng * torch.cos(pitch) * torch.sin(yaw), rng * torch.sin(pitch),
                         rng * torch.cos(pitch) * torch.cos(yaw)], 1)
    fx = W / 2.0
    log_scales = torch.log(rng / fx)[:, None] + 0.3 * torch.randn(N, 3, generator=g)
    return dict(means3D=means.float(), rgb_colors=torch.rand(N, 3, generator=g).float(),
                unnorm_rotations=torch.randn(N, 4, generator=g).float(),
                logit_opacities=torch.randn(N, 1, generator=g).float(), log_scales=log_scales.float())


def uneven_shell_scene(N, seed=0, W=640, H=480, dense_share=0.75):
    """shell_scene with `dense_share` of the Gaussians in HALF of the azimuth range: the keyframes that look into the dense half (yaw in (0, pi))
    see dense_share / (1 - dense_share) = 3 times the tile instances of the others -- the load imbalance of real keyframes (a wall at one metre
    against a view down a corridor), which contiguous keyframe blocks hand to the ranks as they come."""
    d = shell_scene(N, seed, W, H)
    g = torch.Generator().manual_seed(seed + 7919)
    dense = torch.rand(N, generator=g) < dense_share
    yaw = torch.rand(N, generator=g) * math.pi + torch.where(dense, torch.zeros(N), torch.full((N,), math.pi))
    m = d["means3D"]
    r_xz = torch.sqrt(m[:, 0] ** 2 + m[:, 2] ** 2)
    d["means3D"] = torch.stack([r_xz * torch.sin(yaw), m[:, 1], r_xz * torch.cos(yaw)], 1).float()
    return d


def keyframe_w2c(i, K):
    """In-place rotation about +y by yaw 2*pi*i/K (Habitat-like bootstrap spin)."""
    a = 2 * math.pi * i / K
    c, s = math.cos(a), math.sin(a)
    c2w = np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float64)
    return np.linalg.inv(c2w)


def quat_from_yaw(a):
    """(w, x, y, z) of a rotation by `a` radians about +y."""
    return np.array([math.cos(a / 2), 0.0, math.sin(a / 2), 0.0], dtype=np.float32)


def orbit_sequence(gt_params, num_frames, W, H, device, yaw_step_deg=10.0, K=None):
    """Synthetic RGB-D sequence for the mapper harness (substitute for Habitat/Gibson):
    an in-place spin (10 degree turns, like config/env/activesplat_pointnav.yaml) inside a fixed ground-truth
    Gaussian scene, rendered with this package's rasteriser.  Yields the dicts SplatMapper.run() takes; the
    pose is the w2c of the frame relative to frame 0 as quaternion + translation."""
    from .camera import setup_camera
    from .rasterizer import GaussianRasterizer
    K = intrinsics(W, H) if K is None else K
    rv = {k: v.to(device) for k, v in activate(gt_params).items()}
    for i in range(num_frames):
        a = math.radians(yaw_step_deg) * i
        c, s = math.cos(a), math.sin(a)
        w2c = np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float64)
        cam = setup_camera(W, H, K, w2c, device=device)
        with torch.no_grad():
            m2d = torch.zeros_like(rv["means3D"])
            color, _, depth, opacity = GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)
        d = torch.where(opacity > 0.5, depth / opacity.clamp_min(1e-6), torch.zeros_like(depth))
        yield dict(id=i, color=color.clamp(0, 1), depth=d, quat=quat_from_yaw(a), position=np.zeros(3, np.float32), w2c=w2c)
