"""Generate the golden vectors under tests/golden/ by IMPORTING the reference's Python on CPU.

Run only in the build container (needs /root/reference):   python tests/golden/make_golden.py
The reference's files never travel: only the small .npz inputs/outputs written here are committed.
Import recipe: SURVEY.md Appendix C (namespace shells for the packages whose __init__ needs
cv2/open3d/rospy, `diff_gaussian_rasterization` resolved to a stub renderer that returns preset tensors,
.cuda() -> identity, device='cuda' -> 'cpu').

What is pinned:
  camera.npz     setup_camera                       (recon_helpers.py:4-28)
  transform.npz  transform_to_frame, transformed_params2rendervar, transformed_params2depthplussilhouette,
                 get_rendervars                      (slam_helpers.py:124-139,196-249,252-304; splatam.py:436-468)
  rot.npz        build_rotation, quat_mult           (slam_external.py:25-42; slam_helpers.py:21-28)
  loss.npz       l1_loss_v1, calc_ssim, get_loss     (slam_helpers.py:5-6; slam_external.py:54-97; splatam.py:172-301)
  adam.npz       initialize_optimizer + Adam steps   (splatam.py:118-124)
  prune.npz      prune_gaussians, remove_points, cat_params_to_optimizer, update_params_and_optimizer,
                 accumulate_mean2d_gradient, densify (isotropic, no timestep)  (slam_external.py:100-247)
  densify_t.npz  the second executable densify variant: gs_external.densify, isotropic WITH `timestep` inherited by clones and
                 split children, [N,3] means2D through its accumulate_mean2d_gradient (no early return)   (gs_external.py:100-105,191-253)
  pointcloud.npz get_pointcloud, initialize_params, initialize_new_params, add_new_gaussians
                                                     (splatam.py:25-115,304-379)
  keyframe.npz   keyframe_selection_overlap          (keyframe_selection.py:40-95)
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

HERE = os.path.dirname(os.path.abspath(__file__))
R = "/root/reference/src"


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, f, t, a=(), k=None):
        k = dict(k or {})
        if str(k.get("device", "")).startswith("cuda"):
            k["device"] = "cpu"
        return f(*a, **k)


class StubSettings(tuple):
    pass


def install_stub():
    """A `diff_gaussian_rasterization` whose Renderer returns whatever STUB['outputs'] holds."""
    from typing import NamedTuple

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

    stub = {"outputs": [], "calls": []}

    class GaussianRasterizer(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, **kw):
            stub["calls"].append({k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in kw.items()})
            return stub["outputs"].pop(0)

    m = types.ModuleType("diff_gaussian_rasterization")
    m.GaussianRasterizationSettings = GaussianRasterizationSettings
    m.GaussianRasterizer = GaussianRasterizer
    sys.modules["diff_gaussian_rasterization"] = m
    return stub


def n(t):
    """snapshot (a COPY: the optimiser updates its tensors in place after we record them)"""
    return t.detach().cpu().numpy().copy() if torch.is_tensor(t) else np.array(t)


def main():
    for name, p in [("mapper", R + "/mapper"), ("mapper.splatam", R + "/mapper/splatam"),
                    ("mapper.splatam.utils", R + "/mapper/splatam/utils")]:
        m = types.ModuleType(name)
        m.__path__ = [p]
        sys.modules[name] = m
    sys.modules["cv2"] = types.ModuleType("cv2")
    stub = install_stub()
    torch.Tensor.cuda = lambda self, *a, **k: self
    with CudaToCpu():
        sp = importlib.import_module("mapper.splatam.splatam")
        from mapper.splatam.utils import keyframe_selection, recon_helpers, slam_external, slam_helpers

        # ---------------- camera ----------------
        out = {}
        cases = [(640, 480, 320.0, 320.0, 319.0, 239.0, np.eye(4), 0.01, 100.0, 1.0),
                 (256, 256, 128.0, 128.0, 127.0, 127.0, None, 0.01, 100.0, 1.0),
                 (120, 150, 120 / (2 * np.tan(np.deg2rad(60))), 150 / (2 * np.tan(np.deg2rad(75))), 59.0, 74.0, None, 0.01, 100.0, 0.01)]
        rng = np.random.RandomState(0)
        for i, (W, H, fx, fy, cx, cy, w2c, near, far, mod) in enumerate(cases):
            if w2c is None:
                a = rng.randn(3); a /= np.linalg.norm(a); th = 0.7 * (i + 1)
                Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
                Rm = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
                w2c = np.eye(4); w2c[:3, :3] = Rm; w2c[:3, 3] = rng.randn(3)
            K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32 if i == 1 else np.float64)
            cam = recon_helpers.setup_camera(W, H, K, w2c, near, far, scale_modifier=mod)
            out.update({f"c{i}_WH": np.array([W, H]), f"c{i}_K": K, f"c{i}_w2c": w2c, f"c{i}_nearfar": np.array([near, far]),
                        f"c{i}_mod": np.array(mod), f"c{i}_view": n(cam.viewmatrix), f"c{i}_proj": n(cam.projmatrix),
                        f"c{i}_tanfov": np.array([float(cam.tanfovx), float(cam.tanfovy)]), f"c{i}_campos": n(cam.campos),
                        f"c{i}_bg": n(cam.bg)})
        np.savez_compressed(os.path.join(HERE, "camera.npz"), **out)

        # ---------------- rot ----------------
        g = torch.Generator().manual_seed(0)
        q1, q2 = torch.randn(32, 4, generator=g), torch.randn(32, 4, generator=g)
        np.savez_compressed(os.path.join(HERE, "rot.npz"), q1=n(q1), q2=n(q2), build_rotation=n(slam_external.build_rotation(q1)),
                 quat_mult=n(slam_helpers.quat_mult(q1, q2)))

        # ---------------- transform ----------------
        out = {}
        for tag, ncol in (("aniso", 3), ("iso", 1)):
            g = torch.Generator().manual_seed(1)
            N, T = 64, 5
            params = dict(means3D=torch.randn(N, 3, generator=g), rgb_colors=torch.rand(N, 3, generator=g),
                          unnorm_rotations=torch.randn(N, 4, generator=g), logit_opacities=torch.randn(N, 1, generator=g),
                          log_scales=torch.randn(N, ncol, generator=g) * 0.3 - 3.0,
                          cam_unnorm_rots=torch.randn(1, 4, T, generator=g), cam_trans=torch.randn(1, 3, T, generator=g) * 0.2)
            w2c = torch.eye(4); w2c[:3, 3] = torch.tensor([0.1, -0.2, 0.3])
            tg = slam_helpers.transform_to_frame(params, 3, gaussians_grad=True, camera_grad=False)
            rv = slam_helpers.transformed_params2rendervar(params, tg)
            dv = slam_helpers.transformed_params2depthplussilhouette(params, w2c, tg)
            rv2, dv2 = sp.get_rendervars(params, n(w2c))
            out.update({f"{tag}_{k}": n(v) for k, v in params.items()})
            out.update({f"{tag}_w2c": n(w2c), f"{tag}_tg_means3D": n(tg["means3D"]), f"{tag}_tg_rots": n(tg["unnorm_rotations"])})
            out.update({f"{tag}_rv_{k}": n(v) for k, v in rv.items()})
            out.update({f"{tag}_dv_colors": n(dv["colors_precomp"]), f"{tag}_grv_scales": n(rv2["scales"]),
                        f"{tag}_grv_dcolors": n(dv2["colors_precomp"]), f"{tag}_grv_rot": n(rv2["rotations"])})
        np.savez_compressed(os.path.join(HERE, "transform.npz"), **out)

        # ---------------- loss ----------------
        g = torch.Generator().manual_seed(2)
        H, W, N = 48, 64, 40
        im_r = torch.rand(3, H, W, generator=g).requires_grad_(True)
        ds_r = torch.rand(3, H, W, generator=g) * torch.tensor([3.0, 1.0, 9.0]).view(3, 1, 1)
        ds_r = ds_r.clone().requires_grad_(True)
        gt_im = torch.rand(3, H, W, generator=g)
        gt_depth = torch.rand(1, H, W, generator=g) * 3.0
        gt_depth[0, :4, :7] = 0.0                                    # invalid-depth pixels
        radius = (torch.rand(N, generator=g) * 6).floor()
        params = dict(means3D=torch.randn(N, 3, generator=g), rgb_colors=torch.rand(N, 3, generator=g),
                      unnorm_rotations=torch.randn(N, 4, generator=g), logit_opacities=torch.randn(N, 1, generator=g),
                      log_scales=torch.randn(N, 3, generator=g) * 0.3 - 3.0, cam_unnorm_rots=torch.randn(1, 4, 3, generator=g),
                      cam_trans=torch.randn(1, 3, 3, generator=g) * 0.1)
        K = torch.tensor([[32.0, 0, 31.0], [0, 32.0, 23.0], [0, 0, 1.0]])
        cam = recon_helpers.setup_camera(W, H, n(K), np.eye(4))
        curr = dict(cam=cam, im=gt_im, depth=gt_depth, id=1, intrinsics=K, w2c=torch.eye(4))
        variables = dict(max_2D_radius=torch.rand(N, generator=g) * 4, means2D_gradient_accum=torch.zeros(N), denom=torch.zeros(N),
                         timestep=torch.zeros(N))
        max2d_before = variables["max_2D_radius"].clone()
        stub["outputs"] = [(im_r, radius, None, None), (ds_r, None, None, None)]
        loss, variables, wl = sp.get_loss(params, curr, variables, 1, dict(im=0.5, depth=1.0), True, 0.99, True, False, mapping=True)
        loss.backward()
        np.savez_compressed(os.path.join(HERE, "loss.npz"), im_r=n(im_r), ds_r=n(ds_r), gt_im=n(gt_im), gt_depth=n(gt_depth), radius=n(radius),
                 l1=n(slam_helpers.l1_loss_v1(im_r, gt_im)), ssim=n(slam_external.calc_ssim(im_r, gt_im)), loss=n(loss),
                 loss_im=n(wl["im"]), loss_depth=n(wl["depth"]), d_im=n(im_r.grad), d_ds=n(ds_r.grad),
                 max2d_before=n(max2d_before), max2d_after=n(variables["max_2D_radius"]), seen=n(variables["seen"]))
        stub["calls"].clear()

        # ---------------- adam ----------------
        g = torch.Generator().manual_seed(3)
        N, T = 50, 4
        raw = dict(means3D=torch.randn(N, 3, generator=g), rgb_colors=torch.rand(N, 3, generator=g),
                   unnorm_rotations=torch.randn(N, 4, generator=g), logit_opacities=torch.randn(N, 1, generator=g),
                   log_scales=torch.randn(N, 3, generator=g), cam_unnorm_rots=torch.randn(1, 4, T, generator=g),
                   cam_trans=torch.randn(1, 3, T, generator=g))
        lrs = dict(means3D=0.0001, rgb_colors=0.0025, unnorm_rotations=0.001, logit_opacities=0.05, log_scales=0.001,
                   cam_unnorm_rots=0.0000, cam_trans=0.0000)
        params = {k: torch.nn.Parameter(v.clone()) for k, v in raw.items()}
        opt = sp.initialize_optimizer(params, lrs, tracking=False)
        out = {f"p0_{k}": n(v) for k, v in raw.items()}
        out.update({f"lr_{k}": np.array(v) for k, v in lrs.items()})
        dflt = opt.defaults
        out.update(betas=np.array(dflt["betas"]), eps=np.array(dflt["eps"]), weight_decay=np.array(dflt["weight_decay"]),
                   amsgrad=np.array(dflt["amsgrad"]))
        for s in range(1, 4):
            for k, p in params.items():
                if k.startswith("cam_"):
                    p.grad = None                                    # camera_grad=False: skipped entirely
                else:
                    p.grad = torch.randn(p.shape, generator=g) * (10.0 ** float(torch.randint(-3, 2, (1,), generator=g)))
                    out[f"g{s}_{k}"] = n(p.grad)
            opt.step()
            for k, p in params.items():
                out[f"p{s}_{k}"] = n(p)
                st = opt.state.get(p, None)
                if st:
                    out[f"m{s}_{k}"] = n(st["exp_avg"]); out[f"v{s}_{k}"] = n(st["exp_avg_sq"]); out[f"t{s}_{k}"] = n(st["step"])
        np.savez_compressed(os.path.join(HERE, "adam.npz"), **out)

        # ---------------- prune / densify ----------------
        out = {}
        for tag, ncol in (("aniso", 3), ("iso", 1)):
            g = torch.Generator().manual_seed(4)
            N = 200
            raw = dict(means3D=torch.randn(N, 3, generator=g), rgb_colors=torch.rand(N, 3, generator=g),
                       unnorm_rotations=torch.randn(N, 4, generator=g), logit_opacities=torch.randn(N, 1, generator=g) * 3,
                       log_scales=torch.randn(N, ncol, generator=g) - 2.0, cam_unnorm_rots=torch.randn(1, 4, 2, generator=g),
                       cam_trans=torch.randn(1, 3, 2, generator=g))
            params = {k: torch.nn.Parameter(v.clone()) for k, v in raw.items()}
            opt = sp.initialize_optimizer(params, lrs, tracking=False)
            for k, p in params.items():
                if not k.startswith("cam_"):
                    p.grad = torch.randn(p.shape, generator=g)
            opt.step()
            variables = dict(means2D_gradient_accum=torch.rand(N, generator=g), denom=torch.rand(N, generator=g).round() + 1,
                             max_2D_radius=torch.rand(N, generator=g) * 5, timestep=torch.arange(N).float(), scene_radius=torch.tensor(1.5))
            out.update({f"{tag}_p0_{k}": n(v) for k, v in params.items()})
            for k, p in params.items():
                st = opt.state.get(p, None)
                if st:
                    out[f"{tag}_m0_{k}"] = n(st["exp_avg"]); out[f"{tag}_v0_{k}"] = n(st["exp_avg_sq"])
            out.update({f"{tag}_var0_{k}": n(v) for k, v in variables.items()})
            pdict = dict(start_after=0, remove_big_after=0, stop_after=20, prune_every=20, removal_opacity_threshold=0.005,
                         final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=500)
            out.update({f"{tag}_pdict_{k}": np.array(v) for k, v in pdict.items()})
            params, variables = slam_external.prune_gaussians(params, variables, opt, 0, pdict)
            out.update({f"{tag}_p1_{k}": n(v) for k, v in params.items()})
            for k, p in params.items():
                st = opt.state.get(p, None)
                if st:
                    out[f"{tag}_m1_{k}"] = n(st["exp_avg"]); out[f"{tag}_v1_{k}"] = n(st["exp_avg_sq"]); out[f"{tag}_t1_{k}"] = n(st["step"])
            out.update({f"{tag}_var1_{k}": n(v) for k, v in variables.items()})
            # cat_params_to_optimizer: append 7 new rows
            M = 7
            newp = dict(means3D=torch.randn(M, 3, generator=g), rgb_colors=torch.rand(M, 3, generator=g),
                        unnorm_rotations=torch.randn(M, 4, generator=g), logit_opacities=torch.zeros(M, 1),
                        log_scales=torch.randn(M, ncol, generator=g))
            out.update({f"{tag}_new_{k}": n(v) for k, v in newp.items()})
            params = slam_external.cat_params_to_optimizer(newp, params, opt)
            out.update({f"{tag}_p2_{k}": n(v) for k, v in params.items()})
            for k, p in params.items():
                st = opt.state.get(p, None)
                if st:
                    out[f"{tag}_m2_{k}"] = n(st["exp_avg"]); out[f"{tag}_v2_{k}"] = n(st["exp_avg_sq"]); out[f"{tag}_t2_{k}"] = n(st["step"])
            # opacity reset (update_params_and_optimizer)
            newo = {"logit_opacities": slam_external.inverse_sigmoid(torch.ones_like(params["logit_opacities"]) * 0.01)}
            params = slam_external.update_params_and_optimizer(newo, params, opt)
            out[f"{tag}_p3_logit_opacities"] = n(params["logit_opacities"])
            st = opt.state[params["logit_opacities"]]
            out[f"{tag}_m3_logit_opacities"] = n(st["exp_avg"]); out[f"{tag}_t3_logit_opacities"] = n(st["step"])
        # densify: the only variant of slam_external.densify that executes as shipped:
        # isotropic scales, variables WITHOUT 'timestep'
        g = torch.Generator().manual_seed(5)
        N = 300
        raw = dict(means3D=torch.randn(N, 3, generator=g), rgb_colors=torch.rand(N, 3, generator=g),
                   unnorm_rotations=torch.randn(N, 4, generator=g), logit_opacities=torch.randn(N, 1, generator=g) * 2,
                   log_scales=torch.randn(N, 1, generator=g) * 0.7 - 4.0, cam_unnorm_rots=torch.randn(1, 4, 2, generator=g),
                   cam_trans=torch.randn(1, 3, 2, generator=g))
        params = {k: torch.nn.Parameter(v.clone()) for k, v in raw.items()}
        opt = sp.initialize_optimizer(params, lrs, tracking=False)
        for k, p in params.items():
            if not k.startswith("cam_"):
                p.grad = torch.randn(p.shape, generator=g)
        opt.step()
        m2d = torch.zeros(N, 2, requires_grad=True)
        m2d.grad = torch.randn(N, 2, generator=g) * 3e-4
        seen = torch.rand(N, generator=g) > 0.3
        variables = dict(means2D=m2d, seen=seen, means2D_gradient_accum=torch.rand(N, generator=g) * 4e-4,
                         denom=(torch.rand(N, generator=g) * 3).floor(), max_2D_radius=torch.rand(N, generator=g) * 5,
                         scene_radius=torch.tensor(2.0))
        ddict = dict(start_after=0, remove_big_after=0, stop_after=100, densify_every=10, grad_thresh=0.0002, num_to_split_into=2,
                     removal_opacity_threshold=0.005, final_removal_opacity_threshold=0.005, reset_opacities=False,
                     reset_opacities_every=3000)
        out.update({f"den_p0_{k}": n(v) for k, v in params.items()})
        for k, p in params.items():
            st = opt.state.get(p, None)
            if st:
                out[f"den_m0_{k}"] = n(st["exp_avg"]); out[f"den_v0_{k}"] = n(st["exp_avg_sq"])
        out.update(den_m2d_grad=n(m2d.grad), den_seen=n(seen), den_accum0=n(variables["means2D_gradient_accum"]),
                   den_denom0=n(variables["denom"]), den_scene_radius=np.array(2.0))
        out.update({f"den_ddict_{k}": np.array(v) for k, v in ddict.items()})
        # record the normal samples torch.normal draws inside densify
        captured = {}
        real_normal = torch.normal

        def spy_normal(*a, **k):
            r = real_normal(*a, **k)
            captured["samples"] = r.detach().clone()
            return r
        torch.normal = spy_normal
        torch.manual_seed(123)
        params, variables = slam_external.densify(params, variables, opt, 10, ddict)
        torch.normal = real_normal
        out["den_samples"] = n(captured["samples"])
        out.update({f"den_p1_{k}": n(v) for k, v in params.items()})
        for k, p in params.items():
            st = opt.state.get(p, None)
            if st:
                out[f"den_m1_{k}"] = n(st["exp_avg"]); out[f"den_v1_{k}"] = n(st["exp_avg_sq"]); out[f"den_t1_{k}"] = n(st["step"])
        out.update(den_accum1=n(variables["means2D_gradient_accum"]), den_denom1=n(variables["denom"]),
                   den_max2d1=n(variables["max_2D_radius"]))
        np.savez_compressed(os.path.join(HERE, "prune.npz"), **out)

        # ---------------- densify, second executable variant: gs_external.densify (isotropic, WITH timestep) ----------------
        from mapper.splatam.utils import gs_external
        g = torch.Generator().manual_seed(15)
        N = 320
        raw = dict(means3D=torch.randn(N, 3, generator=g), rgb_colors=torch.rand(N, 3, generator=g),
                   unnorm_rotations=torch.randn(N, 4, generator=g), logit_opacities=torch.randn(N, 1, generator=g) * 2,
                   log_scales=torch.randn(N, 1, generator=g) * 0.7 - 4.0, cam_unnorm_rots=torch.randn(1, 4, 2, generator=g),
                   cam_trans=torch.randn(1, 3, 2, generator=g))
        params = {k: torch.nn.Parameter(v.clone()) for k, v in raw.items()}
        opt = sp.initialize_optimizer(params, lrs, tracking=False)
        for k, p in params.items():
            if not k.startswith("cam_"):
                p.grad = torch.randn(p.shape, generator=g)
        opt.step()
        m2d = torch.zeros(N, 3, requires_grad=True)                       # the [P,3] carrier the rasteriser's callers pass
        m2d.grad = torch.randn(N, 3, generator=g) * 3e-4
        seen = torch.rand(N, generator=g) > 0.3
        variables = dict(means2D=m2d, seen=seen, means2D_gradient_accum=torch.rand(N, generator=g) * 4e-4,
                         denom=(torch.rand(N, generator=g) * 3).floor(), max_2D_radius=torch.rand(N, generator=g) * 5,
                         timestep=torch.randint(0, 40, (N,), generator=g).float(), scene_radius=torch.tensor(2.0))
        ddict = dict(start_after=0, remove_big_after=0, stop_after=100, densify_every=10, grad_thresh=0.0002, num_to_split_into=2,
                     removal_opacity_threshold=0.005, final_removal_opacity_threshold=0.005, reset_opacities=False,
                     reset_opacities_every=3000)
        out = {f"p0_{k}": n(v) for k, v in params.items()}
        for k, p in params.items():
            st = opt.state.get(p, None)
            if st:
                out[f"m0_{k}"] = n(st["exp_avg"]); out[f"v0_{k}"] = n(st["exp_avg_sq"])
        out.update(m2d_grad=n(m2d.grad), seen=n(seen), accum0=n(variables["means2D_gradient_accum"]), denom0=n(variables["denom"]),
                   max2d0=n(variables["max_2D_radius"]), timestep0=n(variables["timestep"]), scene_radius=np.array(2.0))
        out.update({f"ddict_{k}": np.array(v) for k, v in ddict.items()})
        captured = {}

        def spy_normal_t(*a, **k):
            r = real_normal(*a, **k)
            captured["samples"] = r.detach().clone()
            return r
        torch.normal = spy_normal_t
        torch.manual_seed(321)
        params, variables = gs_external.densify(params, variables, opt, 10, ddict)
        torch.normal = real_normal
        out["samples"] = n(captured["samples"])
        out.update({f"p1_{k}": n(v) for k, v in params.items()})
        for k, p in params.items():
            st = opt.state.get(p, None)
            if st:
                out[f"m1_{k}"] = n(st["exp_avg"]); out[f"v1_{k}"] = n(st["exp_avg_sq"]); out[f"t1_{k}"] = n(st["step"])
        out.update(accum1=n(variables["means2D_gradient_accum"]), denom1=n(variables["denom"]), max2d1=n(variables["max_2D_radius"]),
                   timestep1=n(variables["timestep"]))
        np.savez_compressed(os.path.join(HERE, "densify_t.npz"), **out)

        # ---------------- pointcloud / new gaussians ----------------
        g = torch.Generator().manual_seed(6)
        H, W = 24, 32
        color = torch.rand(3, H, W, generator=g)
        depth = torch.rand(1, H, W, generator=g) * 3 + 0.2
        depth[0, 2:5, 3:9] = 0.0
        K = torch.tensor([[16.0, 0, 15.0], [0, 16.0, 11.0], [0, 0, 1.0]])
        w2c = torch.eye(4); w2c[:3, :3] = slam_external.build_rotation(torch.tensor([[0.9, 0.1, -0.2, 0.05]]))[0]; w2c[:3, 3] = torch.tensor([0.2, 0.1, -0.3])
        mask = (depth > 0).reshape(-1)
        pc, msd = sp.get_pointcloud(color, depth, K, w2c, mask=mask, compute_mean_sq_dist=True, mean_sq_dist_method="projective")
        out = dict(color=n(color), depth=n(depth), K=n(K), w2c=n(w2c), mask=n(mask), pc=n(pc), msd=n(msd))
        for tag in ("anisotropic", "isotropic"):
            p, v = sp.initialize_params(pc, 3, msd, tag)
            out.update({f"init_{tag}_{k}": n(x) for k, x in p.items()})
            out.update({f"initvar_{tag}_{k}": n(x) for k, x in v.items()})
        # add_new_gaussians with a stub silhouette render
        p, v = sp.initialize_params(pc[:100], 3, msd[:100], "anisotropic")
        with torch.no_grad():
            p["cam_unnorm_rots"][..., 1] = torch.tensor([[0.95, 0.05, 0.1, -0.02]])
            p["cam_trans"][..., 1] = torch.tensor([[0.05, -0.02, 0.1]])
        ds = torch.rand(3, H, W, generator=g)
        ds[0] = ds[0] * 3.5
        cam = recon_helpers.setup_camera(W, H, n(K), np.eye(4))
        curr = dict(cam=cam, im=color, depth=depth, id=1, intrinsics=K, w2c=torch.eye(4))
        stub["outputs"] = [(ds, None, None, torch.rand(1, H, W, generator=g))]
        out.update({f"add_p0_{k}": n(x) for k, x in p.items()})
        out.update(add_depth_sil=n(ds))
        p2, v2 = sp.add_new_gaussians(p, v, curr, 0.5, 1, "projective", "anisotropic")
        out.update({f"add_p1_{k}": n(x) for k, x in p2.items()})
        out.update({f"add_var1_{k}": n(x) for k, x in v2.items()})
        out["add_call_colors"] = n(stub["calls"][-1]["colors_precomp"])
        np.savez_compressed(os.path.join(HERE, "pointcloud.npz"), **out)

        # ---------------- keyframe selection ----------------
        g = torch.Generator().manual_seed(7)
        H, W = 60, 80
        gt_depth = torch.rand(1, H, W, generator=g) * 3 + 0.5
        gt_depth[0, :5, :] = 0
        K = torch.tensor([[40.0, 0, 39.0], [0, 40.0, 29.0], [0, 0, 1.0]])
        w2c = torch.eye(4)
        kfs = []
        for i in range(6):
            kw = torch.eye(4)
            kw[:3, :3] = slam_external.build_rotation(torch.tensor([[1.0, 0.0, 0.15 * i, 0.0]]))[0]
            kw[:3, 3] = torch.tensor([0.1 * i, 0.0, -0.05 * i])
            kfs.append({"id": i, "est_w2c": kw})
        real_randint = torch.randint
        cap = {}

        def spy_randint(*a, **k):
            r = real_randint(*a, **k)
            cap["idx"] = r.clone()
            return r
        torch.randint = spy_randint
        torch.manual_seed(11); np.random.seed(11)
        real_perm = np.random.permutation
        np.random.permutation = lambda x: x                              # keep the sorted order (the shuffle is RNG only)
        sel = keyframe_selection.keyframe_selection_overlap(gt_depth, w2c, K, kfs, 4, pixels=200)
        np.random.permutation = real_perm
        torch.randint = real_randint
        np.savez_compressed(os.path.join(HERE, "keyframe.npz"), gt_depth=n(gt_depth), K=n(K), w2c=n(w2c),
                 kf_w2c=np.stack([n(k["est_w2c"]) for k in kfs]), sampled=n(cap["idx"]), selected=np.array(sel))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
