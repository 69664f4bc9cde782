"""Seeded randomized sweep on the host-emulated kernels: image size, Gaussian count, pose, splat size, opacity range,
background, scale_modifier and colour mode drawn at random; every draw must match the oracle like the named edge cases do
(integer artefacts bit-exact, images and gradients at the stated fp32 tolerances)."""
import numpy as np
import pytest
import torch

from tests import parity_cases as pc
from tests import util


def _draw(seed, device):
    r = np.random.RandomState(seed)
    W, H = int(r.randint(17, 130)), int(r.randint(17, 110))
    N = int(r.choice([1, 2, 63, 64, 65, 255, 256, 257, int(r.randint(300, 1500))]))
    yaw = float(r.uniform(-0.6, 0.6))
    t = (float(r.uniform(-0.4, 0.4)), float(r.uniform(-0.3, 0.3)), float(r.uniform(-1.5, 0.8)))
    sh = [None, None, 0, 1, 2, 3][int(r.randint(0, 6))]
    rs, rv = util.scene(N, W, H, seed=int(r.randint(0, 10_000)), device=device, w2c=util.pose(yaw, t),
                        bg=tuple(float(v) for v in r.uniform(0, 1, 3)), scale_modifier=float(r.choice([1.0, 1.0, 0.37, 2.5])),
                        sh_degree=sh, scale_jitter=float(r.uniform(0.0, 0.8)))
    rv["scales"] = rv["scales"] * float(np.exp(r.uniform(-1.5, 3.0)))            # from sub-pixel dots to screen-filling blobs
    rv["opacities"] = (rv["opacities"] * float(r.uniform(0.02, 1.0))).clamp(0, 1)
    if r.rand() < 0.2:                                                               # precomputed covariance instead of scale/rotation
        from oracle.dense_torch import build_cov3d
        S = build_cov3d(rv["scales"].cpu(), rv["rotations"].cpu(), 1.0)
        rv["cov3D_precomp"] = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).float().to(device)
        rv.pop("scales"); rv.pop("rotations")
    return rs, rv


@pytest.mark.parametrize("seed", range(120))
def test_random_scene_matches_oracle(emu, oracle32, oracle64, seed):
    rs, rv = _draw(1000 + seed, emu)
    pc.check_forward(rs, rv, oracle32)
    pc.check_backward(rs, rv, oracle64, oracle32=oracle32)
    if seed % 3 == 0 and "colors_precomp" in rv and "cov3D_precomp" not in rv:
        pc.check_fused_rgbd(rs, rv, oracle64, seed=seed, oracle32=oracle32)   # the single-pass RGB-D render and its depth-gradient backward (as the GPU sweep does)


@pytest.mark.parametrize("seed", [4 * k + 1 for k in range(75_000, 75_008)])
def test_larger_sweep_scene_matches_oracle(emu, oracle32, oracle64, seed):
    """The sweeps' larger draw (tests/fuzz_scenes.py, seed % 4 == 1): 3 000..30 000 Gaussians on up to 200 x 160 pixels -- several binning chunks and tile
    lists of thousands on the emulated kernels."""
    from tests.fuzz_scenes import sweep_scene
    rs, rv = sweep_scene(seed, emu)
    pc.check_forward(rs, rv, oracle32)
    pc.check_backward(rs, rv, oracle64, oracle32=oracle32)


@pytest.mark.parametrize("seed", range(330_000, 330_024))
def test_anisotropic_scene_matches_oracle(emu, oracle32, oracle64, seed):
    """The sweeps' HARD = 1.2 variant: every axis of every splat scaled by an independent exp(N(0, 1.2)) (median anisotropy 8, maximum above 2 000), every odd
    seed's scene right in front of the near plane -- the scenes whose needle splats the factorised 2-D covariance was introduced for."""
    from tests.fuzz_scenes import hard_scene
    rs, rv = hard_scene(seed, emu, 1.2)
    pc.check_forward(rs, rv, oracle32, oracle64=oracle64)
    pc.check_backward(rs, rv, oracle64, oracle32=oracle32)
