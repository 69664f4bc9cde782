"""Keyframe selection by re-projection overlap.

Same call as the reference's `keyframe_selection_overlap`
(src/mapper/splatam/utils/keyframe_selection.py:40-95, helper get_pointcloud :10-37): sample `pixels`
valid-depth pixels of the current frame, back-project them to the world, and rank the keyframes by the
fraction that lands inside their image (20-px border) -- the ranking of all keyframes in ONE kernel launch.  `sampled` optionally injects the pixel sample
(indices into the valid-depth pixel list) so that a run can be replayed; the reference's final
np.random.permutation of the ranked list is reproduced with `shuffle=True`.
"""
from __future__ import annotations

import numpy as np
import torch


def world_points(depth, intrinsics, w2c, pixels_yx):
    """The sampled pixels (rows of (y, x)) of the depth image [1,H,W] as world points [n,3]: pinhole back-projection, then the camera-to-world
    transform (keyframe_selection.py:10-26)."""
    K = intrinsics
    y, x = pixels_yx[:, 0], pixels_yx[:, 1]
    z = depth[0, y, x]
    cam = torch.stack(((x - K[0][2]) / K[0][0] * z, (y - K[1][2]) / K[1][1] * z, z), dim=-1)
    c2w = torch.inverse(w2c)
    return cam @ c2w[:3, :3].T + c2w[:3, 3]


def drop_repeated_points(pts):
    """The reference's "remove points at camera origin" filter as it actually behaves (keyframe_selection.py:28-36): a point is dropped when its
    coordinate magnitudes, rounded to four decimals, occur MORE THAN ONCE among all the points plus the origin -- so a pixel drawn twice by the
    sampler (torch.randint draws with replacement) loses both copies, and so does a point at the origin.
    Exact integer formulation: the rounded magnitudes as integers (units of 1e-4), three 21-bit fields of one int64 key, one sort."""
    q = torch.round(pts.abs().float() * 1e4).to(torch.int64)        # (torch.round: half to even, as round(decimals=4) does)
    if q.numel() and int(q.max()) >= (1 << 21):                    # beyond 209 m: rows compared as triples
        rows = torch.cat([q, torch.zeros(1, 3, dtype=torch.int64, device=q.device)])
        _, inverse, counts = rows.unique(dim=0, return_inverse=True, return_counts=True)
        return pts[(counts[inverse] == 1)[: q.shape[0]]]
    key = (q[:, 0] << 42) | (q[:, 1] << 21) | q[:, 2]
    order = torch.argsort(key)
    ks = key[order]
    same_as_next = torch.zeros_like(ks, dtype=torch.bool)
    if ks.numel() > 1:
        same_as_next[:-1] = ks[1:] == ks[:-1]
    repeated = same_as_next.clone()
    repeated[1:] |= same_as_next[:-1]
    repeated |= ks == 0                                             # coincides with the appended origin row
    keep = torch.ones_like(repeated)
    keep[order] = ~repeated
    return pts[keep]


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


def keyframe_selection_overlap(gt_depth, w2c, intrinsics, keyframe_list, k, pixels=1600, sampled=None, shuffle=True, return_percent=False):
    """-> ids of up to `k` keyframes whose images see the current frame's surface (keyframe_selection.py:40-95).  The scoring of ALL keyframes
    is one launch of gs_keyframe_overlap (the reference's per-keyframe torch loop is the tests' comparison baseline, tests/reference_pattern.py)."""
    width, height = gt_depth.shape[2], gt_depth.shape[1]
    valid = torch.stack(torch.where(gt_depth[0] > 0), dim=1)
    if sampled is None:
        sampled = torch.randint(valid.shape[0], (pixels,))
    pts = drop_repeated_points(world_points(gt_depth, intrinsics, w2c, valid[sampled.to(valid.device)]))
    n = max(int(pts.shape[0]), 1)
    ranked = [{"id": kid, "percent_inside": torch.tensor(c / n)} for kid, c in
              enumerate(overlap_counts(pts, keyframe_list, intrinsics, width, height, 20))]
    ranked = sorted(ranked, key=lambda d: d["percent_inside"], reverse=True)
    sel = [d["id"] for d in ranked if d["percent_inside"] > 0.0]
    if shuffle:
        sel = list(np.random.permutation(np.array(sel))[:k])
    else:
        sel = sel[:k]
    return (sel, ranked) if return_percent else sel
