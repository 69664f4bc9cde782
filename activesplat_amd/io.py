"""Hand-off formats either side of the mapping path.

* `params.npz` -- the artefact the reference's offline tools read.  Writer: src/mapper/splatam/utils/common_utils.py:25-44
  (`params2cpu`, `save_params`), checkpoint variant :61-68; the 14 keys and the trimming of the per-frame camera
  tensors are assembled in src/mapper/splatam/__init__.py:555-572.
* `GaussianPacket` -- what the mapper thread posts to the visualiser queue (src/utils/gui_utils.py:76-88,
  src/mapper/splatam/__init__.py:536-542).
* `cut_gaussian_by_height` -- the visualiser's height filter (src/visualizer/visualizer.py:2277-2286), here on the
  library's stream-compaction kernels (gs_compact_index + gs_gather_rows) instead of five boolean-mask gathers.
* `load_params` -- NEW (the reference has a `load_checkpoint` flag but no loading code).
"""
from __future__ import annotations

import os

import numpy as np
import torch

PARAM_KEYS = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales", "cam_unnorm_rots", "cam_trans")
FILE_KEYS = PARAM_KEYS + ("timestep", "intrinsics", "w2c", "org_width", "org_height", "gt_w2c_all_frames",
                          "keyframe_time_indices")
GAUSSIAN_ROW_KEYS = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")


def params2cpu(params):
    return {k: (v.detach().cpu().contiguous().numpy() if isinstance(v, torch.Tensor) else v) for k, v in params.items()}


def finalize_params(params, variables, intrinsics, first_frame_w2c, org_width, org_height, gt_w2c_all_frames,
                    keyframe_time_indices):
    """The dict `save_params` receives at the end of a run: parameters + camera bookkeeping, `cam_*` trimmed to the
    number of frames actually mapped (the tensors are pre-allocated for the maximum)."""
    out = dict(params)
    out["timestep"] = variables["timestep"]
    out["intrinsics"] = _np(intrinsics)
    out["w2c"] = _np(first_frame_w2c)
    out["org_width"], out["org_height"] = org_width, org_height
    out["gt_w2c_all_frames"] = np.stack([_np(w) for w in gt_w2c_all_frames], axis=0)
    out["keyframe_time_indices"] = np.array(keyframe_time_indices)
    frames = out["gt_w2c_all_frames"].shape[0]
    for k in ("cam_trans", "cam_unnorm_rots"):
        if out[k].shape[-1] > frames:
            out[k] = out[k][..., :frames]
    return out


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save_params(output_params, output_dir):
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, "params.npz")
    np.savez(path, **params2cpu(output_params))
    return path


def save_params_ckpt(output_params, output_dir, time_idx):
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, "params" + str(time_idx) + ".npz")
    np.savez(path, **params2cpu(output_params))
    return path


def load_params(path, device, requires_grad=True):
    """-> (params dict of torch.nn.Parameter for PARAM_KEYS, extras dict of numpy arrays for the rest)."""
    z = np.load(path, allow_pickle=False)
    params = {k: torch.nn.Parameter(torch.from_numpy(np.array(z[k])).float().to(device).contiguous(), requires_grad=requires_grad)
              for k in PARAM_KEYS if k in z.files}
    extras = {k: np.array(z[k]) for k in z.files if k not in PARAM_KEYS}
    return params, extras


class GaussianPacket:
    def __init__(self, params=None, current_frame=None, finish=False):
        self.has_gaussians = False
        if params is not None:
            self.has_gaussians = True
            self.params = params
            self.current_frame = current_frame


def cut_gaussian_by_height(gaussian_params, upper_limit, lower_limit):
    """Drops rows with  -y < upper_limit  or  -y > lower_limit  (argument names as in the reference: "upper" is the
    smaller bound of -y).  The five per-Gaussian tensors are compacted with one index build + five row gathers."""
    from . import optim as O
    y = -gaussian_params["means3D"].detach()[:, 1]
    keep = ~torch.logical_or(y < upper_limit, y > lower_limit)
    index = O.build_index(keep)
    for k in GAUSSIAN_ROW_KEYS:
        gaussian_params[k] = O.gather_rows(gaussian_params[k], index)
    return gaussian_params
