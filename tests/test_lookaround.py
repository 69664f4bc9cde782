"""Look-around panorama of the planner (activesplat_amd/lookaround.py; reference src/mapper/splatam/__init__.py:698-790):
intrinsics / yaw conventions as closed forms, and the fused path (one activation + one raster pass per view) against
the reference call pattern (get_rendervars + two-pass render per view) on the emulated kernels."""
import numpy as np
import pytest
import torch

from activesplat_amd import lookaround as LA
from activesplat_amd import synthetic as syn


def test_look_around_intrinsics_one_pixel_per_degree():
    fx, fy, cx, cy = LA.compute_intrinsics(120, 150, np.deg2rad(120), np.deg2rad(150))
    assert np.isclose(fx, 60 / np.tan(np.deg2rad(60))) and np.isclose(fy, 75 / np.tan(np.deg2rad(75)))
    assert (cx, cy) == (59, 74)
    fx2, fy2, _, _ = LA.compute_intrinsics(256, 256, np.deg2rad(90))
    assert np.isclose(fx2, 128.0) and fy2 == fx2              # habitat convention: fy = fx when no vfov is given
    k = LA.look_around_k()
    assert k.shape == (3, 3) and k[2, 2] == 1 and k[0, 2] == 59 and k[1, 2] == 74


def test_rot_axis_is_a_rotation_about_the_cameras_own_axis():
    c2w = np.eye(4); c2w[:3, 3] = [1.0, 2.0, 3.0]
    r = LA.rot_axis(c2w, "y", np.deg2rad(120))
    assert np.allclose(r[:3, 3], [1, 2, 3])                    # position untouched
    assert np.allclose(r[:3, :3] @ r[:3, :3].T, np.eye(3)) and np.isclose(np.linalg.det(r[:3, :3]), 1.0)
    assert np.allclose(r[:3, 1], [0, 1, 0])                    # yaw keeps the camera's y axis
    full = LA.rot_axis(LA.rot_axis(r, "y", np.deg2rad(120)), "y", np.deg2rad(120))
    assert np.allclose(full, c2w, atol=1e-12)                  # three views close the circle
    for ax in "xz":
        assert np.allclose(LA.rot_axis(c2w, ax, 0.0), c2w)
    with pytest.raises(ValueError):
        LA.rot_axis(c2w, "w", 0.1)


def _params(n, device):
    p = syn.shell_scene(n, seed=2, W=LA.LOOK_W, H=LA.LOOK_H)
    return {k: v.to(device) for k, v in p.items()}


def test_fused_look_around_equals_reference_call_pattern(emu):
    params = _params(1500, emu)
    c2w = np.eye(4); c2w[:3, 3] = [0.1, 0.0, -0.2]
    a = LA.look_around(params, c2w, fused=True)
    b = LA.look_around(params, c2w, fused=False)
    assert a["opacity"].shape == (150, 360) and a["rgb"].shape == (150, 360, 3) and a["depth"].shape == (150, 360, 1)
    assert a["rgb"].dtype == torch.uint8
    assert float(a["opacity"].max()) > 0.5                     # the shell surrounds the camera: every view sees it
    for v in range(3):
        assert float(a["opacity"][:, 120 * v:120 * (v + 1)].max()) > 0.5
    assert torch.allclose(a["opacity"], b["opacity"], atol=2e-5)
    assert torch.allclose(a["depth"], b["depth"], atol=2e-5, rtol=1e-5)
    assert int((a["rgb"].int() - b["rgb"].int()).abs().max()) <= 1
    c = LA.look_around(params, c2w, fused=True, batched=False)          # one raster pass per view vs the multi-view atlas (a view's
    assert torch.allclose(a["opacity"], c["opacity"], atol=1e-5)        # pixel x is offset by its slot: fp32 rounding of dx differs)
    assert torch.allclose(a["depth"], c["depth"], atol=2e-5, rtol=1e-5)
    assert int((a["rgb"].int() - c["rgb"].int()).abs().max()) <= 1
    inv = LA.local_invisibility(params, c2w)
    assert np.isclose(inv, float((1 - b["opacity"]).sum()), rtol=1e-4)
    assert LA.global_invisibility_inputs(params, c2w, np.zeros(3)) is None
    depth_np, inv_np = LA.global_invisibility_inputs(params, c2w, np.array([0.3, 9.0, 0.1]))
    assert depth_np.shape == (150, 360, 1) and inv_np.shape == (150, 360)
    moved = np.array(c2w); moved[0, 3], moved[2, 3] = 0.3, 0.1           # y (camera height) is kept
    assert np.allclose(inv_np, (1 - LA.look_around(params, moved)["opacity"]).cpu().numpy())
