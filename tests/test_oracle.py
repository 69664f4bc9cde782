"""CPU tests of the oracle itself (no GPU, no product code):
  * the hand-derived backward of oracle/gs_oracle.c  ==  fp64 autograd of oracle/dense_torch.py,
  * closed-form known answers,
  * call-site invariants the reference implies (section 8c-iii),
  * metamorphic properties (section 8c-v)."""
import math

import numpy as np
import pytest
import torch

from oracle.dense_torch import build_cov3d, render_dense
from tests import util


def _cam(W, H, **kw):
    rs, _ = util.scene(1, W, H, **kw)
    return rs


@pytest.mark.parametrize("mode", ["rgb", "sh", "cov"])
def test_c_oracle_backward_equals_fp64_autograd(oracle64, mode):
    torch.manual_seed(0)
    W, H, N = 48, 40, 300
    rs, rv = util.scene(N, W, H, seed=3, w2c=util.pose(0.2, (0.1, -0.05, 0.2)), bg=(0.1, 0.2, 0.3), scale_modifier=1.3,
                        sh_degree=3 if mode == "sh" else None)
    cd = util.cam_dict(rs)
    inp = {k: v.double().clone().requires_grad_(True) for k, v in rv.items()}
    if mode == "cov":
        S = build_cov3d(inp["scales"].detach(), inp["rotations"].detach(), 1.0)
        cov = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)
        inp.pop("scales"); inp.pop("rotations")
        inp["cov3D_precomp"] = cov.clone().requires_grad_(True)
    m2d = torch.zeros(N, 3, dtype=torch.float64, requires_grad=True)
    d = render_dense(cd, inp["means3D"], inp["opacities"], colors=inp.get("colors_precomp"), shs=inp.get("shs"),
                     scales=inp.get("scales"), rotations=inp.get("rotations"), cov3D_precomp=inp.get("cov3D_precomp"),
                     means2D=m2d)
    dL = torch.randn(3, H, W, dtype=torch.float64)
    (d["color"] * dL).sum().backward()
    f = util.run_oracle(oracle64, rs, {k: v.detach() for k, v in inp.items()}, dL)
    assert np.array_equal(f["radii"], d["radii"].numpy())
    assert np.array_equal(f["n_contrib"], d["n_contrib"].numpy())
    for a, b in ((f["color"], d["color"]), (f["out_depth"], d["depth"]), (f["opacity"], d["opacity"])):
        np.testing.assert_allclose(a, b.detach().numpy(), atol=1e-12)
    for k, t in inp.items():
        g = t.grad.numpy()
        np.testing.assert_allclose(f["grads"][k].reshape(g.shape), g, atol=1e-10 * max(1.0, np.abs(g).max()))
    np.testing.assert_allclose(f["grads"]["means2D"], m2d.grad.numpy(), atol=1e-10 * np.abs(m2d.grad.numpy()).max())


def _single(oracle, W=64, H=48, z=2.0, s=0.05, o=0.8, px=20, py=17, bg=(0.0, 0.0, 0.0), col=(0.2, 0.5, 0.9)):
    rs = _cam(W, H, bg=bg)
    K = util.syn.intrinsics(W, H)
    mean = np.array([[(px - K[0, 2]) / K[0, 0] * z, (py - K[1, 2]) / K[1, 1] * z, z]], np.float32)
    rv = dict(means3D=torch.tensor(mean), opacities=torch.tensor([[o]]), colors_precomp=torch.tensor([col]),
              scales=torch.full((1, 3), s), rotations=torch.tensor([[1.0, 0, 0, 0]]))
    return rs, rv, util.run_oracle(oracle, rs, rv), K


def test_closed_form_single_isotropic_gaussian(oracle64):
    """On the optical axis (perspective Jacobian has no shear): alpha(d) = min(0.99, o exp(-|d|^2 / (2 sigma'^2))),
    sigma'^2 = (f s / z)^2 + 0.3; the projected mean sits at u - 0.5 (pixel centres at integers)."""
    W, H, z, s, o = 64, 48, 2.0, 0.05, 0.8
    px, py = W // 2 - 1, H // 2 - 1                      # principal point of synthetic.intrinsics
    rs, rv, f, K = _single(oracle64, W=W, H=H, z=z, s=s, o=o, px=px, py=py)
    sig2 = (K[0, 0] * s / z) ** 2 + 0.3
    assert f["radii"][0] == math.ceil(3 * math.sqrt(sig2 + math.sqrt(0.1)))   # lambda floor max(0.1, .)
    for dx, dy in ((0, 0), (1, 0), (0, 2), (-2, 1), (-1, -1)):
        d2 = (dx + 0.5) ** 2 + (dy + 0.5) ** 2
        a = min(0.99, o * math.exp(-d2 / (2 * sig2)))
        assert abs(f["opacity"][0, py + dy, px + dx] - a) < 1e-6
        assert abs(f["color"][1, py + dy, px + dx] - 0.5 * a) < 1e-6
        assert abs(f["out_depth"][0, py + dy, px + dx] - z * a) < 1e-5


def test_closed_form_two_stacked_and_background(oracle64):
    W, H = 64, 48
    px, py = W // 2 - 1, H // 2 - 1
    rs = _cam(W, H, bg=(1.0, 1.0, 1.0))
    K = util.syn.intrinsics(W, H)
    zs, os_, cols = (3.0, 1.5), (0.6, 0.7), ((1.0, 0.0, 0.0), (0.0, 1.0, 0.0))       # far one listed first
    mean = np.array([[(px - K[0, 2]) / K[0, 0] * z, (py - K[1, 2]) / K[1, 1] * z, z] for z in zs], np.float32)
    rv = dict(means3D=torch.tensor(mean), opacities=torch.tensor(os_).reshape(2, 1), colors_precomp=torch.tensor(cols),
              scales=torch.full((2, 3), 0.05), rotations=torch.tensor([[1.0, 0, 0, 0]] * 2))
    f = util.run_oracle(oracle64, rs, rv)
    sig2 = [(K[0, 0] * 0.05 / z) ** 2 + 0.3 for z in zs]
    a_far, a_near = (o * math.exp(-0.5 / (2 * s2)) for o, s2 in zip(os_, sig2))
    T = (1 - a_near) * (1 - a_far)
    exp_c = np.array([a_far * (1 - a_near) * 1.0, a_near * 1.0, 0.0]) + T * 1.0
    np.testing.assert_allclose(f["color"][:, py, px], exp_c, atol=1e-6)
    assert abs(f["opacity"][0, py, px] - (1 - T)) < 1e-6
    assert abs(f["out_depth"][0, py, px] - (zs[1] * a_near + zs[0] * a_far * (1 - a_near))) < 1e-5
    assert f["n_contrib"][py, px] == 2
    assert np.allclose(f["color"][:, 0, 0], 1.0)                                      # untouched pixel = bg


def test_callsite_invariants_depth_and_silhouette_channels(oracle32):
    """With bg=0, colours [z_cam, 1, z_cam^2] (slam_helpers.py:196-213): channel 1 == opacity output,
    channel 0 == depth output."""
    W, H, N = 80, 64, 1500
    rs, rv = util.scene(N, W, H, seed=1, w2c=util.pose(-0.1, (0.0, 0.02, 0.1)))
    cd = util.cam_dict(rs)
    p4 = np.concatenate([rv["means3D"].numpy(), np.ones((N, 1), np.float32)], 1)
    zc = (p4 @ cd["viewmatrix"])[:, 2]
    rv["colors_precomp"] = torch.tensor(np.stack([zc, np.ones_like(zc), zc * zc], 1).astype(np.float32))
    f = util.run_oracle(oracle32, rs, rv)
    np.testing.assert_allclose(f["color"][1], f["opacity"][0], atol=2e-6)
    np.testing.assert_allclose(f["color"][0], f["out_depth"][0], atol=2e-5, rtol=1e-5)
    assert ((f["radii"] > 0) == (f["tiles_touched"] > 0)).all()


def test_property_permutation_invariance_and_scale_modifier(oracle64):
    W, H, N = 64, 48, 600
    rs, rv = util.scene(N, W, H, seed=7)
    f0 = util.run_oracle(oracle64, rs, rv)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(0))
    f1 = util.run_oracle(oracle64, rs, {k: v[perm] for k, v in rv.items()})
    # ties in (tile, fp32 depth) are broken by index, so allow a handful of pixels to differ
    assert util.close_frac(f1["color"], f0["color"], 0, 1e-9) > 0.995
    assert np.array_equal(f1["radii"], f0["radii"][perm.numpy()])
    rs2 = rs._replace(scale_modifier=0.5)
    rv2 = dict(rv, scales=rv["scales"] * 2.0)
    f2 = util.run_oracle(oracle64, rs2, rv2)
    np.testing.assert_allclose(f2["color"], f0["color"], atol=1e-9)


def test_f32_and_f64_builds_agree(oracle32, oracle64):
    W, H, N = 96, 80, 3000
    rs, rv = util.scene(N, W, H, seed=2, bg=(0.3, 0.2, 0.1))
    dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1))
    a, b = util.run_oracle(oracle32, rs, rv, dL), util.run_oracle(oracle64, rs, rv, dL)
    assert np.mean(a["radii"] == b["radii"]) > 0.999
    assert util.close_frac(a["color"], b["color"], 1e-4, 2e-5) > 0.999
    assert util.psnr(a["color"], b["color"]) > 60
    for k in ("means3D", "scales", "rotations", "opacities", "colors_precomp", "means2D"):
        g = b["grads"][k]
        assert util.close_frac(a["grads"][k], g, 1e-3, 1e-5 * np.abs(g).max()) > 0.995, k


def test_no_far_plane_culling_topdown_view(oracle32):
    """Camera 1000 m away with far=100 and scale_modifier 0.01 must still render."""
    W, H = 64, 64
    f_px = 2000.0
    K = np.array([[f_px, 0, W / 2 - 1], [0, f_px, H / 2 - 1], [0, 0, 1]])
    rs = util.setup_camera(W, H, K, util.pose(0.0, (0.0, 0.0, 1000.0)), device="cpu", scale_modifier=0.01)
    g = torch.Generator().manual_seed(0)
    N = 500
    means = torch.cat([(torch.rand(N, 2, generator=g) - 0.5) * 30.0, torch.zeros(N, 1)], 1)
    rv = dict(means3D=means, opacities=torch.full((N, 1), 0.9), colors_precomp=torch.rand(N, 3, generator=g),
              scales=torch.full((N, 3), 0.05), rotations=torch.tensor([[1.0, 0, 0, 0]]).repeat(N, 1))
    f = util.run_oracle(oracle32, rs, rv)
    assert (f["radii"] > 0).sum() > 0.9 * N
    assert f["opacity"].max() > 0.5
    assert set(np.unique(f["radii"][f["radii"] > 0])) == {3}      # every splat collapses to the 0.3 px^2 low-pass footprint: ceil(3 sqrt(0.3 + sqrt(0.1)))


def test_adam_matches_torch_optim_adam(oracle64):
    """Adam semantics of splatam.py:118-124 (eps=1e-15, lr per group)."""
    torch.manual_seed(0)
    p = torch.randn(1000, dtype=torch.float64)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": [ref], "lr": 2.5e-3}], lr=0.0, eps=1e-15)
    pn, m, v = p.numpy().copy(), np.zeros(1000), np.zeros(1000)
    for step in range(1, 6):
        g = torch.randn(1000, dtype=torch.float64)
        ref.grad = g.clone()
        opt.step()
        pn, m, v = oracle64.adam(pn, g.numpy(), m, v, 2.5e-3, step)
        np.testing.assert_allclose(pn, ref.detach().numpy(), rtol=1e-12, atol=1e-14)


def test_c_oracle_depth_gradient_equals_fp64_autograd(oracle64):
    """The fused RGB-D backward: loss over colour AND the blended depth output."""
    torch.manual_seed(1)
    W, H, N = 48, 40, 300
    rs, rv = util.scene(N, W, H, seed=5, w2c=util.pose(-0.15, (0.05, 0.02, 0.1)), bg=(0.2, 0.3, 0.1))
    cd = util.cam_dict(rs)
    inp = {k: v.double().clone().requires_grad_(True) for k, v in rv.items()}
    m2d = torch.zeros(N, 3, dtype=torch.float64, requires_grad=True)
    d = render_dense(cd, inp["means3D"], inp["opacities"], colors=inp["colors_precomp"], scales=inp["scales"],
                     rotations=inp["rotations"], means2D=m2d)
    dLc = torch.randn(3, H, W, dtype=torch.float64); dLd = torch.randn(1, H, W, dtype=torch.float64)
    ((d["color"] * dLc).sum() + (d["depth"] * dLd).sum()).backward()
    f = util.run_oracle(oracle64, rs, {k: v.detach() for k, v in inp.items()})
    g = oracle64.backward(f, dLc.numpy(), dLd.numpy())
    for k, t in inp.items():
        a = t.grad.numpy()
        np.testing.assert_allclose(g[k].reshape(a.shape), a, atol=1e-10 * max(1.0, np.abs(a).max()))
    np.testing.assert_allclose(g["means2D"], m2d.grad.numpy(), atol=1e-10 * np.abs(m2d.grad.numpy()).max())
