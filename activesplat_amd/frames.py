"""Frame pre-processing of the mapper: what happens to a sensor frame before it reaches the
optimise loop.  Reference: src/mapper/splatam/__init__.py:332-378 --

* pose: X_WV is conjugated with OPENCV_TO_OPENGL (src/utils/__init__.py:10-17), made relative to the first frame and
  inverted to a world-to-camera matrix (:333-338);
* colour: cv2.resize(INTER_LINEAR) to the mapping resolution, then [3,H,W] / 255 (:341-348);
* depth : cv2.resize(INTER_NEAREST), then [1,H,W] (:343-349);
* the same pair again at the densification resolution with intrinsics / densify_downscale_factor (:362-376).

cv2 is not part of this stack, so the two resizes are restated with cv2's sampling conventions: INTER_NEAREST takes
source index floor(dst * src/dst_size); INTER_LINEAR samples at pixel centres ((dst + 0.5) * scale - 0.5, clamped to
the border) and rounds half up -- identical to cv2 except that cv2 quantises the two interpolation weights to 11 bits
for uint8 input, which can move a result by one grey level.
"""
from __future__ import annotations

import numpy as np
import torch

OPENCV_TO_OPENGL = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], dtype=np.float64)


def resize_nearest(img: np.ndarray, width: int, height: int) -> np.ndarray:
    h, w = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(height) * (h / height)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(width) * (w / width)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def resize_linear(img: np.ndarray, width: int, height: int) -> np.ndarray:
    """uint8 in -> uint8 out (rounded half up); float in -> float out."""
    h, w = img.shape[:2]
    if (h, w) == (height, width):
        return img.copy()
    src = img.astype(np.float64)

    def axis(n_dst, n_src):
        c = (np.arange(n_dst) + 0.5) * (n_src / n_dst) - 0.5
        i0 = np.floor(c).astype(np.int64)
        f = c - i0
        return np.clip(i0, 0, n_src - 1), np.clip(i0 + 1, 0, n_src - 1), f
    y0, y1, fy = axis(height, h)
    x0, x1, fx = axis(width, w)
    fx = fx.reshape((1, -1) + (1,) * (src.ndim - 2))
    fy = fy.reshape((-1, 1) + (1,) * (src.ndim - 2))
    top = src[y0][:, x0] * (1 - fx) + src[y0][:, x1] * fx
    bot = src[y1][:, x0] * (1 - fx) + src[y1][:, x1] * fx
    out = top * (1 - fy) + bot * fy
    if img.dtype == np.uint8:
        return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)
    return out.astype(img.dtype)


def to_mapping_tensors(image: np.ndarray, depth: np.ndarray, width: int, height: int, device):
    """image [h,w,3] (uint8 or float 0..255), depth [h,w] metres -> (color [3,H,W] in 0..1, depth [1,H,W]) on `device`."""
    color = torch.from_numpy(np.ascontiguousarray(resize_linear(image, width, height))).to(device).float().permute(2, 0, 1) / 255
    d = np.expand_dims(resize_nearest(depth, width, height), -1)
    return color, torch.from_numpy(np.ascontiguousarray(d)).to(device).float().permute(2, 0, 1)


def gt_w2c_from_pose(X_WV: np.ndarray, first_abs_pose: np.ndarray | None):
    """-> (gt_w2c [4,4] float32 numpy, first_abs_pose).  The pose is re-expressed in the OpenGL camera convention and
    made relative to the first frame, so frame 0 has the identity."""
    pose = (OPENCV_TO_OPENGL @ np.asarray(X_WV, dtype=np.float64) @ OPENCV_TO_OPENGL).astype(np.float32)
    if first_abs_pose is None:
        first_abs_pose = pose
    rel = np.linalg.inv(first_abs_pose.astype(np.float64)) @ pose.astype(np.float64)
    return np.linalg.inv(rel).astype(np.float32), first_abs_pose


def densify_intrinsics(intrinsics, factor: float):
    k = torch.as_tensor(intrinsics, dtype=torch.float32).clone() / factor
    k[2, 2] = 1.0
    return k
