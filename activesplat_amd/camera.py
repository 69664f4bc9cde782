"""Camera -> rasteriser settings.

Mirror of the reference's `setup_camera` (src/mapper/splatam/utils/recon_helpers.py:4-28): same
argument meaning, same matrix conventions (viewmatrix = w2c^T, projmatrix = (P w2c)^T as [1,4,4]
tensors, OpenGL-style projection with off-centre principal point), plus an explicit `device`
(the reference hard-codes .cuda()) and `bg`.
"""
from __future__ import annotations

import numpy as np
import torch

from .rasterizer import GaussianRasterizationSettings


def setup_camera(w, h, k, w2c, near=0.01, far=100, scale_modifier=1.0, bg=(0.0, 0.0, 0.0), device=None,
                 sh_degree=0):
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    fx, fy, cx, cy = float(k[0][0]), float(k[1][1]), float(k[0][2]), float(k[1][2])
    # Host poses (the usual case: numpy / CPU tensors) are processed on the host in fp32 with the same torch ops and only
    # the three small results are copied over: on the device the 4x4 inverse and the 4x4 product would drag in rocSOLVER /
    # rocBLAS (hundreds of ms of one-time initialisation, ~100 us per call) for 2 x 64 flops.
    work = torch.device(device) if (torch.is_tensor(w2c) and w2c.device.type != "cpu") else torch.device("cpu")
    w2c_t = torch.as_tensor(np.asarray(w2c, dtype=np.float64) if not torch.is_tensor(w2c) else w2c).to(
        device=work, dtype=torch.float32)
    cam_center = torch.inverse(w2c_t)[:3, 3]
    view = w2c_t.unsqueeze(0).transpose(1, 2)
    opengl_proj = torch.tensor([[2 * fx / w, 0.0, -(w - 2 * cx) / w, 0.0],
                                [0.0, 2 * fy / h, -(h - 2 * cy) / h, 0.0],
                                [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
                                [0.0, 0.0, 1.0, 0.0]], dtype=torch.float32, device=work).unsqueeze(0).transpose(1, 2)
    full_proj = view.bmm(opengl_proj)
    # values as the reference builds them; stored contiguous and 16-byte aligned so that the rasteriser passes the
    # pointers through instead of re-packing a transposed view / an offset slice on every call.  Host-built blocks travel in ONE
    # buffer (one host-to-device copy instead of four): view at float 0, projection at 16, camera centre at 32, background at 36
    bg_t = torch.tensor(list(bg), dtype=torch.float32)
    if work.type == "cpu":
        pack = torch.empty(40, dtype=torch.float32)
        pack[0:16] = view.reshape(16); pack[16:32] = full_proj.reshape(16); pack[32:35] = cam_center; pack[36:39] = bg_t
        pack = pack.to(device)
        view, full_proj, cam_center, bg_t = pack[0:16].view(1, 4, 4), pack[16:32].view(1, 4, 4), pack[32:35], pack[36:39]
    else:
        view, full_proj, cam_center, bg_t = view.contiguous().to(device), full_proj.contiguous().to(device), cam_center.clone().to(device), bg_t.to(device)
    return GaussianRasterizationSettings(
        image_height=int(h), image_width=int(w), tanfovx=w / (2 * fx), tanfovy=h / (2 * fy),
        bg=bg_t, scale_modifier=scale_modifier,
        viewmatrix=view, projmatrix=full_proj, sh_degree=sh_degree, campos=cam_center,
        prefiltered=False, debug=False)
