"""The planner's look-around renders: three 120 x 150 views, 120 degrees of yaw apart,
hstacked into a 360-degree opacity / RGB / depth panorama whose `1 - opacity` is ActiveSplat's invisibility.

Reference: src/mapper/splatam/__init__.py:698-741 (get_global_invisibility) and :763-790
(get_local_invisibility); intrinsics src/dataloader/__init__.py:275-284; camera yaw src/utils/pose_utils.py:13-43.
The reference spends, per node, 3 x (get_rendervars [8 elementwise torch kernels, twice] + two raster passes
[the second one's only product, the silhouette, is discarded] + 3 blocking D2H copies).  `look_around(fused=True)`
activates the Gaussians once (gs_activate_forward with the identity pose), renders ALL views in one raster pass over a
multi-view atlas (rasterizer.render_views) and leaves the panorama on the device; `fused=False` reproduces the reference op for op (parity tests compare them).
Everything downstream of the arrays (DBSCAN clustering, convex hulls: src/mapper/__init__.py:8-80) is planner
code and out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

from . import mapping as M
from .camera import setup_camera
from .rasterizer import GaussianRasterizer, render_views

LOOK_HFOV_DEG, LOOK_VFOV_DEG, LOOK_W, LOOK_H = 120, 150, 120, 150        # 1 pixel = 1 degree of rotation
VIZ_NEAR, VIZ_FAR = 0.01, 100.0                                          # config/splatam/online_habitat_sim.py:99


def compute_intrinsics(width, height, hfov_rad, vfov_rad=None):
    """(fx, fy, cx, cy) with the reference's principal-point convention cx = W/2 - 1."""
    fx = 0.5 * width / np.tan(hfov_rad / 2.0)
    fy = fx if vfov_rad is None else 0.5 * height / np.tan(vfov_rad / 2.0)
    return fx, fy, width / 2 - 1, height / 2 - 1


def look_around_k():
    fx, fy, cx, cy = compute_intrinsics(LOOK_W, LOOK_H, np.deg2rad(LOOK_HFOV_DEG), np.deg2rad(LOOK_VFOV_DEG))
    return np.array([fx, 0, cx, 0, fy, cy, 0, 0, 1], dtype=np.float64).reshape(3, 3)


def rot_axis(view_c2w, axis, angle_rad):
    """Rotate a camera pose about one of its OWN axes (right-multiplication)."""
    c, s = np.cos(angle_rad), np.sin(angle_rad)
    R = {"x": [[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]],
         "y": [[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]],
         "z": [[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]}.get(axis)
    if R is None:
        raise ValueError("Axis must be 'x', 'y', or 'z'")
    return view_c2w @ np.array(R, dtype=np.float64)


_COLS = {}


def _panorama_columns(views, stride, device):
    key = (views, stride, str(device))
    if key not in _COLS:
        _COLS[key] = torch.cat([torch.arange(v * stride, v * stride + LOOK_W) for v in range(views)]).to(device)
    return _COLS[key]


def _world_rendervar(params):
    """World-frame rendervar through the fused activation kernel (identity pose)."""
    with torch.no_grad():
        m, r, o, s = M._FusedRendervars.apply(params["means3D"], params["unnorm_rotations"], params["logit_opacities"],
                                              params["log_scales"], [1.0, 0, 0, 0, 0, 0, 0])
    return {"means3D": m, "colors_precomp": params["rgb_colors"], "rotations": r, "opacities": o, "scales": s,
            "means2D": torch.zeros_like(m)}


@torch.no_grad()
def look_around(params, view_c2w, scale_modifier=1.0, fused=True, views=None, batched=True):
    """-> dict(opacity [150, 120 V], rgb uint8 [150, 120 V, 3], depth [150, 120 V, 1]) device tensors, V = 360/120 = 3.
    fused=True, batched=True : one activation, ONE raster pass for all V views (rasterizer.render_views: multi-view atlas);
    fused=True, batched=False: one activation, one raster pass per view;
    fused=False              : the reference op for op (two activations + two raster passes per view)."""
    views = int(360 / LOOK_HFOV_DEG) if views is None else views
    k = look_around_k()
    cfg = dict(viz_w=LOOK_W, viz_h=LOOK_H, viz_near=VIZ_NEAR, viz_far=VIZ_FAR)
    device = params["means3D"].device
    w2cs = [np.linalg.inv(rot_axis(np.asarray(view_c2w, dtype=np.float64), "y", np.deg2rad(LOOK_HFOV_DEG * i))) for i in range(views)]
    outs = []
    if fused:
        rv = _world_rendervar(params)
        if batched and views > 1:
            # cameras built on the host (one upload for all views inside render_views), panorama assembled from the atlas with one
            # column gather per output
            cams = [setup_camera(LOOK_W, LOOK_H, k, w2c, VIZ_NEAR, VIZ_FAR, scale_modifier=scale_modifier, device="cpu", bg=(1.0, 1.0, 1.0))
                    for w2c in w2cs]
            rv.pop("means2D")
            color, depth, opacity, stride = render_views(cams, return_atlas=True, **rv)
            cols = _panorama_columns(views, stride, device)
            im = color[:, :, cols]
            return {"opacity": opacity[0][:, cols], "rgb": (torch.clamp(im, min=0, max=1.0) * 255).byte().permute(1, 2, 0).contiguous(),
                    "depth": depth[0][:, cols].unsqueeze(-1)}
        cams = [setup_camera(LOOK_W, LOOK_H, k, w2c, VIZ_NEAR, VIZ_FAR, scale_modifier=scale_modifier, device=device, bg=(1.0, 1.0, 1.0))
                for w2c in w2cs]
        for cam in cams:
            im, _, depth, opacity = GaussianRasterizer(raster_settings=cam)(**rv)
            outs.append((im, depth, opacity))
    else:
        for w2c in w2cs:
            scene, scene_depth = M.get_rendervars(params, torch.tensor(w2c, dtype=torch.float32, device=device))
            im, depth, opacity, _ = M.render(w2c, k, scene, scene_depth, cfg, scale_modifier)
            outs.append((im, depth, opacity))
    rgbs = [(torch.clamp(im, min=0, max=1.0) * 255).byte().permute(1, 2, 0) for im, _, _ in outs]
    ops = [opacity[0] for _, _, opacity in outs]
    deps = [depth.float().permute(1, 2, 0) for _, depth, _ in outs]
    return {"opacity": torch.cat(ops, dim=1), "rgb": torch.cat(rgbs, dim=1).contiguous(), "depth": torch.cat(deps, dim=1)}


def global_invisibility_inputs(params, view_c2w, position, scale_modifier=1.0, fused=True):
    """Arrays get_global_invisibility hands to get_convexhull_volume: (depth_np [150,360,1], invisibility_np [150,360])
    for the agent moved to `position` (x, z replaced; camera height kept).  None for the all-zero position."""
    position = np.asarray(position)
    assert position.shape == (3,), f"Position must be a numpy array with shape (3,), but got {position.shape}"
    if (position == np.zeros(3)).all():
        return None
    c2w = np.array(view_c2w, dtype=np.float64, copy=True)
    c2w[0, 3], c2w[2, 3] = position[0], position[2]
    pano = look_around(params, c2w, scale_modifier, fused)
    return pano["depth"].cpu().numpy(), (1.0 - pano["opacity"]).cpu().numpy()


def local_invisibility(params, view_c2w, scale_modifier=1.0, fused=True):
    """sum(1 - opacity) over the panorama at the agent's pose -- reduced on the device, one scalar copied back."""
    pano = look_around(params, view_c2w, scale_modifier, fused)
    return float((1.0 - pano["opacity"]).sum())
