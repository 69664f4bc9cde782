"""Mapping-loop harness: this build's counterpart of the reference's `SplaTAM.__mapping`
(src/mapper/splatam/__init__.py:254-542), which cannot be imported or shipped (it pulls in cv2, open3d, rospy,
habitat).  It reproduces the CALL PATTERN and SCHEDULE of the hot loop, nothing of the ROS/GUI shell:

  frame 0                       : back-project the RGB-D frame -> initialize_params; scene_radius = max depth / 3
                                  (__init__.py:382-386, config/splatam/online_habitat_sim.py:6)
  every frame                   : write the ground-truth pose into params['cam_*'][..., id] (tracking is skipped,
                                  __init__.py:400-405)
  frames with id == 0 or (id+1) % map_every == 0
                                : add_new_gaussians (id > 0), keyframe_selection_overlap over keyframe_list[:-1]
                                  (window = mapping_window_size - 2, + last keyframe + current frame), FRESH Adam
                                  (__init__.py:408-440)
  iterations                    : iter_per_frame = mapping_iters // map_every, or mapping_iters when that is 0 and
                                  id % map_every == 0 (__init__.py:395-397); each: np.random.randint keyframe pick,
                                  get_loss (two raster passes), backward, optional prune / densify, Adam step,
                                  zero_grad(set_to_none) (__init__.py:447-480)
  keyframes                     : appended when id == 0, (id+1) % keyframe_every == 0 or id == step_num - 2, unless the
                                  frame's ground-truth pose holds inf / nan (__init__.py:514-524)
  every frame once a map exists : get_high_loss_samples' render of the map at the frame's pose (two raster passes in the
                                  reference, one with fused_render) and its "rendered surface in front of a measured one" mask
                                  (__init__.py:184-214, 256-258); the mask's clustering (cv2 resize, DBSCAN) is planner code

Inputs are already-resized frames (`color [3,H,W]` in 0..1, `depth [1,H,W]` metres, pose relative to frame 0 as
quaternion (w,x,y,z) + translation of the w2c) -- the cv2 resize / PNG / manifest work of the reference is I/O
outside the hot path.  Defaults are the shipped configuration (config/splatam/online_habitat_sim.py,
config/datasets/gibson.json).
"""
from __future__ import annotations

import copy
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import mapping as M
from . import optim as O
from .camera import setup_camera
from . import frames as FR
from .keyframes import keyframe_selection_overlap

DEFAULT_CONFIG = dict(
    seed=0, gaussian_distribution="anisotropic", scene_radius_depth_ratio=3, mean_sq_dist_method="projective",
    map_every=5, keyframe_every=5, mapping_window_size=12, mapping_iters=2, step_num=1000,
    # False = the reference's call pattern; True = the fused HIP paths of this build (same loss, see mapping.get_loss)
    fused_render=False,      # one raster pass per iteration (rasterizer.render_rgbd) instead of two
    fused_loss=False,        # masked-L1 + L1 + SSIM value and gradients by gs_mapping_loss
    fused_inputs=False,      # transform_to_frame + activations by gs_activate_*
    fused_preprocess=False,  # ... or inside the rasteriser's per-Gaussian kernels (rasterizer.render_rgbd_raw; needs fused_render): no activation launches
    fused_adam=False,        # ... and the Adam step of an iteration inside that render's backward kernel (needs fused_preprocess; iterations whose
                             # prune / densify event replaces the parameter tensors take the separate step, as the reference's loop effectively does)
    fused_iteration=False,   # ... and the whole iteration (get_loss + backward + step + zero_grad) as four library calls without autograd
                             # (mapping.mapping_iteration; needs fused_render, fused_loss, fused_preprocess; event iterations take the usual path)
    fused_growth=False,      # add_new_gaussians: one forward + gs_grow_gaussians
    fused_keyframes=True,    # (accepted for older configs; no effect: the keyframe overlap scores always come from gs_keyframe_overlap, one launch)
    high_loss_samples=True,  # the per-frame no-grad render of get_high_loss_samples (__init__.py:184-258) before mapping a frame
    mapping=dict(
        loss_weights=dict(im=0.5, depth=1.0), sil_thres=0.98, use_sil_for_loss=False, use_l1=True,
        ignore_outlier_depth_loss=False, add_new_gaussians=True, prune_gaussians=False,
        use_gaussian_splatting_densification=False,
        lrs=dict(means3D=0.0001, rgb_colors=0.0025, unnorm_rotations=0.001, logit_opacities=0.05, log_scales=0.001,
                 cam_unnorm_rots=0.0, cam_trans=0.0),
        pruning_dict=dict(start_after=0, remove_big_after=0, stop_after=20, prune_every=20, removal_opacity_threshold=0.005,
                          final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=500),
        densify_dict=dict(start_after=500, remove_big_after=3000, stop_after=5000, densify_every=100, grad_thresh=0.0002,
                          num_to_split_into=2, removal_opacity_threshold=0.005, final_removal_opacity_threshold=0.005,
                          reset_opacities=False, reset_opacities_every=3000)),
    viz=dict(viz_near=0.01, viz_far=100.0),
)


class SplatMapper:
    def __init__(self, intrinsics, width, height, config=None, device=None):
        self.cfg = copy.deepcopy(DEFAULT_CONFIG)
        if config:
            for k, v in config.items():
                if isinstance(v, dict) and isinstance(self.cfg.get(k), dict):
                    self.cfg[k].update(v)
                else:
                    self.cfg[k] = v
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self.W, self.H = int(width), int(height)
        self.intrinsics = torch.as_tensor(np.asarray(intrinsics), dtype=torch.float32, device=self.device)
        self._k_host = np.asarray(intrinsics, dtype=np.float64)
        self._pose_host = {}               # frame id -> (quaternion, translation) as this mapper wrote them into the camera parameters (host tensors)
        lrs = self.cfg["mapping"]["lrs"]
        self._poses_fixed = float(lrs.get("cam_unnorm_rots", 0.0)) == 0.0 and float(lrs.get("cam_trans", 0.0)) == 0.0
        self.first_frame_w2c = torch.eye(4, device=self.device)
        self.cam = setup_camera(self.W, self.H, np.asarray(intrinsics), np.eye(4), device=self.device)
        self.densify_cam, self.densify_intrinsics = self.cam, self.intrinsics      # replaced when frames carry a densify copy
        self.first_abs_pose = None
        self._one = M.unit_gradient(torch.empty((), dtype=torch.float32, device=self.device))   # (the fused loss knows it: no scaling launch)
        self.params = self.variables = self.optimizer = None
        self.keyframe_list, self.selected_keyframes, self.gt_w2c_all_frames = [], [], []
        self.rng = np.random.RandomState(self.cfg["seed"])
        self.stats = dict(iters=0, iter_time=0.0, frames=0, frame_time=0.0)
        self._last_losses = None           # device scalars of the most recent iteration (read through `last_losses`)
        self.high_loss_mask = None          # bool [H,W] of the most recent frame (None before the first map exists)

    @property
    def last_losses(self):
        """{name: float} of the most recent mapping iteration (None before the first); synchronises with the device."""
        if self._last_losses is None:
            return None
        return {k: float(v.detach()) for k, v in self._last_losses.items()}

    # -- helpers ---------------------------------------------------------------------------------
    @staticmethod
    def _w2c_host(quat, pos):
        """4x4 w2c of a (w,x,y,z) quaternion + translation, on the HOST (the same torch ops as on the device, ~15 of them: issued as
        device launches they were 0.3 ms of host time per call, several times per frame)."""
        w2c = torch.eye(4)
        w2c[:3, :3] = M.build_rotation(F.normalize(quat.view(1, 4)))
        w2c[:3, 3] = pos
        return w2c

    def _w2c(self, frame_id):
        # the poses this mapper wrote into the camera parameters are still on the host (tracking is skipped and their learning rates are
        # zero in the shipped configuration); a pose that is optimised is read from the parameters
        host = self._pose_host.get(int(frame_id)) if self._poses_fixed else None
        if host is not None:
            return self._w2c_host(*host).to(self.device)
        rot = F.normalize(self.params["cam_unnorm_rots"][..., frame_id].detach())
        w2c = torch.eye(4, device=self.device)
        w2c[:3, :3] = M.build_rotation(rot)
        w2c[:3, 3] = self.params["cam_trans"][..., frame_id].detach()
        return w2c

    def _data(self, color, depth, frame_id):
        return {"cam": self.cam, "im": color, "depth": depth, "id": frame_id, "intrinsics": self.intrinsics,
                "w2c": self.first_frame_w2c}

    # -- one frame of the mapper (reference: SplaTAM.__mapping) ----------------------------------------
    def run(self, frame):
        cfg, mc = self.cfg, self.cfg["mapping"]
        fid = int(frame["id"])
        color = frame["color"].to(self.device).float()
        depth = frame["depth"].to(self.device).float()
        quat_h = torch.as_tensor(np.asarray(frame["quat"], dtype=np.float32)).reshape(4).clone()
        pos_h = torch.as_tensor(np.asarray(frame["position"], dtype=np.float32)).reshape(3).clone()
        self._pose_host[fid] = (quat_h, pos_h)
        quat, pos = quat_h.to(self.device), pos_h.to(self.device)
        if self.params is not None and cfg.get("high_loss_samples", True):
            self.high_loss_mask = self.high_loss_samples_mask(self._w2c_host(quat_h, pos_h), depth)       # (host pose: the camera block is built on the host)
        # densification-resolution copy of the frame (reference :362-376); defaults to the mapping resolution
        d_color = frame["densify_color"].to(self.device).float() if "densify_color" in frame else color
        d_depth = frame["densify_depth"].to(self.device).float() if "densify_depth" in frame else depth
        if fid == 0 and d_color.shape[1:] != color.shape[1:]:
            fac = color.shape[2] / d_color.shape[2]
            self.densify_intrinsics = FR.densify_intrinsics(self.intrinsics.cpu(), fac).to(self.device)
            self.densify_cam = setup_camera(d_color.shape[2], d_color.shape[1], self.densify_intrinsics.cpu().numpy(), np.eye(4),
                                            device=self.device)
        if fid == 0:
            init_w2c = torch.eye(4)
            init_w2c[:3, :3] = M.build_rotation(quat_h.view(1, 4))
            init_w2c[:3, 3] = pos_h
            init_w2c = init_w2c.to(self.device)
            mask = (d_depth > 0).reshape(-1)
            cld, msd = M.get_pointcloud(d_color, d_depth, self.densify_intrinsics, init_w2c, mask=mask, compute_mean_sq_dist=True)
            self.params, self.variables = M.initialize_params(cld, cfg["step_num"], msd, cfg["gaussian_distribution"])
            self.variables["scene_radius"] = torch.max(d_depth) / cfg["scene_radius_depth_ratio"]
        map_every = cfg["map_every"]
        iter_per_frame = int(cfg["mapping_iters"] // map_every)
        if iter_per_frame == 0 and fid % map_every == 0:
            iter_per_frame = cfg["mapping_iters"]
        with torch.no_grad():                                    # tracking skip: ground-truth pose
            self.params["cam_unnorm_rots"][..., fid] = quat
            self.params["cam_trans"][..., fid] = pos
        if fid == 0 or (fid + 1) % map_every == 0:
            if mc["add_new_gaussians"] and fid > 0:
                d_data = {"cam": self.densify_cam, "im": d_color, "depth": d_depth, "id": fid, "intrinsics": self.densify_intrinsics,
                          "w2c": self.first_frame_w2c}
                self.params, self.variables = M.add_new_gaussians(self.params, self.variables, d_data,
                                                                  mc["sil_thres"], fid, cfg["gaussian_distribution"],
                                                                  fused=cfg.get("fused_growth", False),
                                                                  pose7=[float(v) for v in F.normalize(quat_h.view(1, 4)).view(4).tolist()]
                                                                  + [float(v) for v in pos_h.tolist()] if cfg.get("fused_growth", False) else None)
            with torch.no_grad():
                sel = keyframe_selection_overlap(depth, self._w2c(fid), self.intrinsics, self.keyframe_list[:-1],
                                                 cfg["mapping_window_size"] - 2)
                self.selected_keyframes = [int(s) for s in sel]
                if len(self.keyframe_list) > 0:
                    self.selected_keyframes.append(len(self.keyframe_list) - 1)
                self.selected_keyframes.append(-1)
            self.optimizer = O.initialize_optimizer(self.params, mc["lrs"], tracking=False)
        t_frame = time.perf_counter()
        for it in range(iter_per_frame):
            t0 = time.perf_counter()
            pick = self.selected_keyframes[self.rng.randint(0, len(self.selected_keyframes))]
            if pick == -1:
                it_id, it_color, it_depth = fid, color, depth
            else:
                kf = self.keyframe_list[pick]
                it_id, it_color, it_depth = kf["id"], kf["color"], kf["depth"]
            # ONE predicate for both fused forms: does this iteration's prune / densify call move rows or reset parameters (then the tensors are
            # replaced between backward and step, and the separate step is taken, as the reference's loop effectively does)?
            event = (mc["prune_gaussians"] and O.prune_event(it, mc["pruning_dict"])) \
                or (mc["use_gaussian_splatting_densification"] and O.densify_event(it, mc["densify_dict"]))
            fused_path = cfg.get("fused_preprocess", False) and cfg["fused_render"] and not event
            in_backward = cfg.get("fused_adam", False) and fused_path
            # (fused_iteration IS the Adam-inside-the-backward form without autograd: it implies fused_adam whatever the config says)
            direct = cfg.get("fused_iteration", False) and fused_path and cfg["fused_loss"] and mc["use_l1"] and not mc["ignore_outlier_depth_loss"]
            if direct:
                # the whole iteration without autograd.  No event here, so prune_gaussians is a no-op by its own predicate (prune_event) and
                # densify only accumulates this iteration's statistics; nothing holds a gradient afterwards, so step() / zero_grad() have
                # nothing to do -- asserted, so that a parameter that starts receiving gradients (camera learning rates > 0) cannot be skipped
                loss, self.variables, losses = M.mapping_iteration(self.params, self._data(it_color, it_depth, it_id), self.variables, it_id,
                                                                   mc["loss_weights"], self.optimizer)
                if mc["use_gaussian_splatting_densification"]:
                    self.params, self.variables = O.densify(self.params, self.variables, self.optimizer, it, mc["densify_dict"])
                if any(p.grad is not None for p in self.params.values()):
                    raise RuntimeError("SplatMapper(fused_iteration=True): a parameter holds a gradient after the autograd-free iteration -- "
                                       "this loop would never step it; use fused_iteration=False")
                self._last_losses = {k: v.detach() for k, v in losses.items()}
                self.stats["iters"] += 1
                self.stats["iter_time"] += time.perf_counter() - t0
                continue
            loss, self.variables, losses = M.get_loss(self.params, self._data(it_color, it_depth, it_id), self.variables, it_id,
                                                      mc["loss_weights"], mc["use_sil_for_loss"], mc["sil_thres"], mc["use_l1"],
                                                      mc["ignore_outlier_depth_loss"], fused=cfg["fused_render"], fused_loss=cfg["fused_loss"],
                                                      fused_inputs=cfg["fused_inputs"], fused_preprocess=cfg.get("fused_preprocess", False),
                                                      fused_adam=self.optimizer if in_backward else None)
            with M.backward_on_calling_thread():        # (no hand-over to autograd's device thread: mapping.backward_on_calling_thread)
                loss.backward(gradient=self._one)       # cached dL/dloss = 1: saves autograd's ones_like launch per iteration
            with torch.no_grad():
                if mc["prune_gaussians"]:
                    self.params, self.variables = O.prune_gaussians(self.params, self.variables, self.optimizer, it, mc["pruning_dict"])
                if mc["use_gaussian_splatting_densification"]:
                    self.params, self.variables = O.densify(self.params, self.variables, self.optimizer, it, mc["densify_dict"])
                self.optimizer.step()
                self.optimizer.zero_grad(set_to_none=True)
            self._last_losses = {k: v.detach() for k, v in losses.items()}   # (no graph kept alive) converted on access: a float() here would stall the host every iteration
            self.stats["iters"] += 1
            self.stats["iter_time"] += time.perf_counter() - t0
        if iter_per_frame > 0:
            self.stats["frames"] += 1
            self.stats["frame_time"] += time.perf_counter() - t_frame
        with torch.no_grad():
            # the simulator's pose when the caller hands one over (run_raw), the pose written into the camera parameters otherwise
            gt_w2c = torch.as_tensor(np.asarray(frame["gt_w2c"]), dtype=torch.float32, device=self.device) if "gt_w2c" in frame \
                else self._w2c(fid)
            pose_ok = bool(torch.isfinite(gt_w2c).all())
            if (fid == 0 or (fid + 1) % cfg["keyframe_every"] == 0 or fid == cfg["step_num"] - 2) and pose_ok:
                self.keyframe_list.append({"id": fid, "est_w2c": self._w2c(fid), "color": color, "depth": depth})
            self.gt_w2c_all_frames.append(gt_w2c)
        return self.params

    @torch.no_grad()
    def high_loss_samples_mask(self, view_w2c, gt_depth):
        """The render + mask half of get_high_loss_samples (__init__.py:184-214): the map rendered at the frame's pose, and
        `rendered depth > measured depth  and  |error| > 0.3 m (measured pixels only)  and  opacity > 0.8` -- regions where the map
        shows a surface in front of free space the sensor saw through.  Stays on the device; the caller's clustering is not ours."""
        im, depth, opacity = self.render_rgbd(view_w2c)
        gt = gt_depth if gt_depth.dim() == 3 else gt_depth.unsqueeze(0)
        err = (depth - gt).abs() * (gt > 0)
        return ((depth > gt) & (err > 0.3) & (opacity > 0.8))[0]

    def run_raw(self, image, depth, X_WV, frame_id, quat, position):
        """Sensor frame in (uint8 [h,w,3] image, [h,w] metric depth, 4x4 pose X_WV): pre-processing of :332-378 (pose to the
        relative OpenGL-convention w2c, resize to the mapping resolution and to the densification resolution), then run().
        quat (w,x,y,z) / position are the tracker's camera parameters for this frame (the reference skips tracking and
        writes the simulator's, :400-405)."""
        gt_w2c, self.first_abs_pose = FR.gt_w2c_from_pose(X_WV, self.first_abs_pose)
        color, d = FR.to_mapping_tensors(image, depth, self.W, self.H, self.device)
        fac = float(self.cfg.get("densify_downscale_factor", 1))
        frame = {"id": frame_id, "color": color, "depth": d, "quat": quat, "position": position, "gt_w2c": gt_w2c}
        if fac != 1:
            frame["densify_color"], frame["densify_depth"] = FR.to_mapping_tensors(image, depth, int(self.W / fac), int(self.H / fac),
                                                                                   self.device)
        return self.run(frame)

    # -- hand-off (reference: q_main2vis.put(GaussianPacket(...)) __init__.py:536-542; post_processing :544-578) ----
    def packet(self, c2w=None):
        from .io import GaussianPacket
        return GaussianPacket(self.params, c2w)

    def post_processing(self, output_dir):
        """Writes params.npz with the reference's 14 keys (cam_* trimmed to the mapped frames)."""
        from . import io as IO
        kf_ids = [int(k["id"]) for k in self.keyframe_list]
        out = IO.finalize_params(self.params, self.variables, self.intrinsics, self.first_frame_w2c, self.W, self.H,
                                 self.gt_w2c_all_frames, kf_ids)
        return IO.save_params(out, output_dir)

    @torch.no_grad()
    def look_around(self, view_c2w, scale_modifier=1.0, fused=True):
        """360-degree opacity / RGB / depth panorama of the planner (lookaround.py)."""
        from . import lookaround as LA
        return LA.look_around(self.params, view_c2w, scale_modifier, fused)

    # -- no-grad consumers (reference: render_rgbd / get_*_invisibility, __init__.py:604-838) ----------------
    @torch.no_grad()
    def render_rgbd(self, w2c, scale_modifier=1.0, width=None, height=None, intrinsics=None):
        k = self._k_host if intrinsics is None else np.asarray(intrinsics)
        cfgv = dict(self.cfg["viz"], viz_w=width or self.W, viz_h=height or self.H)
        if self.cfg.get("fused_preprocess", False) and "rgb_colors" in self.params:
            # the map's PARAMETERS straight into the rasteriser (activations inside its per-Gaussian kernel, identity frame transform: the
            # view is the camera's): no activation launches, and none of the depth / silhouette "colours" every caller discards
            from . import rasterizer as R
            cam = setup_camera(cfgv["viz_w"], cfgv["viz_h"], k, w2c.cpu() if torch.is_tensor(w2c) else w2c, cfgv["viz_near"], cfgv["viz_far"],
                               scale_modifier=scale_modifier, device=self.device, bg=(1.0, 1.0, 1.0))
            p = self.params
            im, _radii, depth, opacity, _dsq = R.render_rgbd_raw(cam, p["means3D"].detach(), torch.empty(0, device=self.device), p["logit_opacities"].detach(),
                                                                 p["log_scales"].detach(), p["unnorm_rotations"].detach(), [1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
                                                                 colors_precomp=p["rgb_colors"].detach())
            return im, depth, opacity
        rv, dv = M.get_rendervars(self.params, w2c)
        im, depth, opacity, _ = M.render(w2c, k, rv, dv, cfgv, scale_modifier=scale_modifier, device=self.device,
                                         with_silhouette=False)
        return im, depth, opacity

    @torch.no_grad()
    def invisibility(self, w2c, width=120, height=150, intrinsics=None):
        """1 - opacity, the quantity the planner scores view points with (__init__.py:739,795)."""
        _, _, opacity = self.render_rgbd(w2c, width=width, height=height, intrinsics=intrinsics)
        return 1.0 - opacity
