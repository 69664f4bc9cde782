"""CPU, build container only (skipped where /root/reference is absent, e.g. on the GPU box): the reference's
OWN caller code -- splatam.get_pointcloud -> initialize_params -> initialize_optimizer -> get_loss -> backward ->
optimizer.step, and the no-grad `render()` -- runs unmodified against this repository's
`diff_gaussian_rasterization` (import recipe of SURVEY.md App. C; kernels on the host emulator), and agrees
with the host-side mirrors in activesplat_amd.mapping on identical inputs."""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch
from torch.overrides import TorchFunctionMode

REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, f, t, a=(), k=None):
        k = dict(k or {})
        if str(k.get("device", "")).startswith("cuda"):
            k["device"] = "cpu"
        return f(*a, **k)


@pytest.fixture()
def reference_modules(emu, monkeypatch):
    for name, p in [("mapper", REF + "/mapper"), ("mapper.splatam", REF + "/mapper/splatam"),
                    ("mapper.splatam.utils", REF + "/mapper/splatam/utils")]:
        m = types.ModuleType(name)
        m.__path__ = [p]
        monkeypatch.setitem(sys.modules, name, m)
    monkeypatch.setitem(sys.modules, "cv2", types.ModuleType("cv2"))
    for k in [k for k in sys.modules if k.startswith("mapper.splatam.")]:
        monkeypatch.delitem(sys.modules, k, raising=False)
    import diff_gaussian_rasterization  # noqa: F401  -- THIS repository's drop-in
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    with CudaToCpu():
        sp = importlib.import_module("mapper.splatam.splatam")
        yield sp


def test_reference_get_loss_backward_step_runs_on_the_dropin(reference_modules):
    sp = reference_modules
    from activesplat_amd import mapping as M
    from activesplat_amd import synthetic as syn
    H, W = 48, 64
    g = torch.Generator().manual_seed(0)
    color = torch.rand(3, H, W, generator=g)
    depth = torch.rand(1, H, W, generator=g) * 2 + 1.0
    K = torch.tensor(syn.intrinsics(W, H), dtype=torch.float32)
    w2c = torch.eye(4)
    with CudaToCpu():
        pc, msd = sp.get_pointcloud(color, depth, K, w2c, compute_mean_sq_dist=True, mean_sq_dist_method="projective")
        params, variables = sp.initialize_params(pc, 2, msd, "anisotropic")
        lrs = dict(means3D=1e-4, rgb_colors=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05, log_scales=1e-3, cam_unnorm_rots=0.0, cam_trans=0.0)
        opt = sp.initialize_optimizer(params, lrs, tracking=False)
        from mapper.splatam.utils.recon_helpers import setup_camera as ref_setup_camera
        cam = ref_setup_camera(W, H, K.numpy(), np.eye(4))
        gt = dict(cam=cam, im=torch.rand(3, H, W, generator=g), depth=depth * 1.05, id=0, intrinsics=K, w2c=w2c)
        loss, variables, wl = sp.get_loss(params, gt, variables, 0, dict(im=0.5, depth=1.0), True, 0.99, True, False, mapping=True)
        loss.backward()
        ref_grads = {k: v.grad.clone() for k, v in params.items() if v.grad is not None}
        m2d_grad = variables["means2D"].grad.clone()
        before = params["means3D"].detach().clone()
        opt.step()
    assert torch.isfinite(loss) and loss > 0 and variables["seen"].sum() > 0.9 * pc.shape[0]
    assert m2d_grad.shape == (pc.shape[0], 3) and m2d_grad[:, :2].abs().sum() > 0 and (m2d_grad[:, 2] == 0).all()
    assert (params["means3D"].detach() - before).abs().max() > 0
    # the mirror on the same inputs gives the same loss and gradients (same rasteriser underneath)
    pc2, msd2 = M.get_pointcloud(color, depth, K, w2c, compute_mean_sq_dist=True)
    mp, mv = M.initialize_params(pc2, 2, msd2, "anisotropic")
    gt2 = dict(gt, cam=cam)
    loss2, mv, _ = M.get_loss(mp, gt2, mv, 0, dict(im=0.5, depth=1.0))
    loss2.backward()
    np.testing.assert_allclose(loss2.item(), loss.item(), rtol=1e-6)
    for k in ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales"):
        np.testing.assert_allclose(mp[k].grad.numpy(), ref_grads[k].numpy(), rtol=1e-4, atol=max(1e-9, 1e-6 * float(ref_grads[k].abs().max())), err_msg=k)


def test_reference_nograd_render_runs_on_the_dropin(reference_modules):
    sp = reference_modules
    from activesplat_amd import synthetic as syn
    W, H, N = 80, 60, 1500
    p = syn.make_params(N, W, H, seed=4)
    params = dict(means3D=p["means3D"], rgb_colors=p["rgb_colors"], unnorm_rotations=p["unnorm_rotations"],
                  logit_opacities=p["logit_opacities"], log_scales=p["log_scales"])
    w2c = np.eye(4)
    with CudaToCpu():
        rv, dv = sp.get_rendervars(params, w2c)
        cfg = dict(viz_w=W, viz_h=H, viz_near=0.01, viz_far=100.0)
        im, depth, opacity, sil = sp.render(w2c, syn.intrinsics(W, H), rv, dv, cfg)
    assert im.shape == (3, H, W) and depth.shape == (1, H, W) and opacity.shape == (1, H, W) and sil.shape == (1, H, W)
    # call-site invariants: silhouette channel == opacity output; white background where empty
    np.testing.assert_allclose(sil.numpy(), opacity.numpy(), atol=3e-6)
    assert float(opacity.min()) >= 0 and float(opacity.max()) <= 1
    empty = opacity[0] < 1e-6
    if empty.any():
        assert torch.allclose(im[:, empty], torch.ones_like(im[:, empty]))
