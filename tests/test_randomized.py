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


@pytest.mark.parametrize("seed", range(24))
def test_random_scene_matches_oracle(emu, oracle32, oracle64, seed):
    rs, rv = _draw(1000 + seed, emu)
    pc.check_forward(rs, rv, oracle32)
    if seed % 2 == 0:
        pc.check_backward(rs, rv, oracle64, oracle32=oracle32)
