"""Keyframe selection by re-projection overlap.

Host-side mirror of the reference's `keyframe_selection_overlap`
(src/mapper/splatam/utils/keyframe_selection.py:40-95, helper get_pointcloud :10-37): sample `pixels`
valid-depth pixels of the current frame, back-project them to the world, and rank the keyframes by the
fraction that lands inside their image (20-px border).  `sampled` optionally injects the pixel sample
(indices into the valid-depth pixel list) so that a run can be replayed; the reference's final
np.random.permutation of the ranked list is reproduced with `shuffle=True`.
"""
from __future__ import annotations

import numpy as np
import torch


def _backproject(depth, intrinsics, w2c, sampled_indices):
    fx, fy, cx, cy = intrinsics[0][0], intrinsics[1][1], intrinsics[0][2], intrinsics[1][2]
    xx = (sampled_indices[:, 1] - cx) / fx
    yy = (sampled_indices[:, 0] - cy) / fy
    z = depth[0, sampled_indices[:, 0], sampled_indices[:, 1]]
    pts_cam = torch.stack((xx * z, yy * z, z), dim=-1)
    c2w = torch.inverse(w2c)
    pts = pts_cam @ c2w[:3, :3].T + c2w[:3, 3]
    # drop points that coincide with the world origin (round to 4 decimals, as the reference does)
    A = torch.abs(torch.round(pts, decimals=4))
    B = torch.zeros((1, 3), device=pts.device, dtype=pts.dtype)
    _, idx, counts = torch.cat([A, B], dim=0).unique(dim=0, return_inverse=True, return_counts=True)
    invalid = torch.isin(idx, torch.where(counts.gt(1))[0])[: len(A)]
    return pts[~invalid]


def overlap_counts(pts, keyframe_list, intrinsics, width, height, edge=20):
    """gs_keyframe_overlap (csrc/grow.hip): how many of the world points land inside each keyframe's image (shrunk by
    `edge` px, positive depth) -- one launch and one copy back for ALL keyframes."""
    import ctypes as C
    from . import _lib
    lib = _lib.get()
    dev = pts.device
    n_kf = len(keyframe_list)
    if n_kf == 0:
        return []
    w2c = torch.stack([kf["est_w2c"].detach().to(dev).float() for kf in keyframe_list]).contiguous()
    p = pts.detach().contiguous().float()
    counts = torch.zeros(n_kf, dtype=torch.int32, device=dev)
    K = np.asarray(intrinsics.detach().cpu() if torch.is_tensor(intrinsics) else intrinsics, dtype=np.float64).reshape(-1)
    k9 = (C.c_float * 9)(*[float(v) for v in K])
    st = _lib.stream_ptr(dev)
    _lib.check(lib.gs_keyframe_overlap(int(p.shape[0]), C.c_void_p(p.data_ptr()), n_kf, C.c_void_p(w2c.data_ptr()), k9, int(width),
                                       int(height), int(edge), C.c_void_p(counts.data_ptr()), st))
    return counts.cpu().tolist()


def keyframe_selection_overlap(gt_depth, w2c, intrinsics, keyframe_list, k, pixels=1600, sampled=None, shuffle=True,
                               return_percent=False, fused=False):
    width, height = gt_depth.shape[2], gt_depth.shape[1]
    valid = torch.stack(torch.where(gt_depth[0] > 0), dim=1)
    if sampled is None:
        sampled = torch.randint(valid.shape[0], (pixels,))
    pts = _backproject(gt_depth, intrinsics, w2c, valid[sampled.to(valid.device)])
    ranked = []
    edge = 20
    if fused:
        n = max(int(pts.shape[0]), 1)
        ranked = [{"id": kid, "percent_inside": torch.tensor(c / n)} for kid, c in
                  enumerate(overlap_counts(pts, keyframe_list, intrinsics, width, height, edge))]
    for kid, kf in enumerate([] if fused else keyframe_list):
        est = kf["est_w2c"]
        tp = pts @ est[:3, :3].T + est[:3, 3]
        p2 = tp @ intrinsics.T
        z = p2[:, 2:] + 1e-5
        uv = (p2 / z)[:, :2]
        inside = (uv[:, 0] < width - edge) & (uv[:, 0] > edge) & (uv[:, 1] < height - edge) & (uv[:, 1] > edge) & (z[:, 0] > 0)
        ranked.append({"id": kid, "percent_inside": inside.sum() / uv.shape[0]})
    ranked = sorted(ranked, key=lambda d: d["percent_inside"], reverse=True)
    sel = [d["id"] for d in ranked if d["percent_inside"] > 0.0]
    if shuffle:
        sel = list(np.random.permutation(np.array(sel))[:k])
    else:
        sel = sel[:k]
    return (sel, ranked) if return_percent else sel
