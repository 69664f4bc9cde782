"""CPU: the tiered gradient rule of tests/parity_cases.py is itself tested -- the threshold-pixel finder on the scene the
round-4 sweep flagged (seed 120013; device analysis in profiles/README.md), and the DECISION-MATCHED third tier end to end on the host-emulated
kernels with the scene the round-5 sweep flagged (seed 140658: the emulated build reproduces the device's decision there)."""
import numpy as np
import torch

from tests import parity_cases as pc
from tests import util
from tests.fuzz_scenes import sweep_scene


def test_threshold_pixels_of_the_flagged_scene(oracle64):
    rs, rv = sweep_scene(120013, "cpu")
    W, H = int(rs.image_width), int(rs.image_height)
    f = util.run_oracle(oracle64, rs, rv)
    found = pc.threshold_gaussians(f, [14305, 4158, 7626, 3586], W, H)
    assert [(i, x, y) for i, x, y, _ in found] == [(14305, 71, 18), (4158, 168, 35), (7626, 159, 44)]
    assert all(abs(d) < pc.THRESHOLD_WINDOW for _, _, _, d in found)
    # a window of 1e-5 is a rare event per Gaussian: a handful in a scene of 14 329
    n = len(pc.threshold_gaussians(f, range(0, 14329, 7), W, H))
    assert n <= 0.02 * (14329 / 7), n


def test_threshold_factor_of_the_oracle_flips_exactly_one_decision(oracle64):
    """gso_set_threshold_scale: the factor of ONE Gaussian moved past its alpha flips its decision at its threshold pixel and nowhere else; None
    restores the published algorithm."""
    rs, rv = sweep_scene(140658, "cpu")
    W, H = int(rs.image_width), int(rs.image_height)
    base = util.run_oracle(oracle64, rs, rv)
    (i, x, y, d), = pc.threshold_gaussians(base, [112], W, H)
    assert (i, x, y) == (112, 0, 39) and -1e-5 < d < 0                       # fp64: a hair below 1/255 -> skipped
    scale = np.ones(256); scale[112] = 1.0 + d - pc.DECISION_MARGIN
    try:
        oracle64.set_threshold_scale(scale)
        flipped = util.run_oracle(oracle64, rs, rv)
    finally:
        oracle64.set_threshold_scale(None)
    diff = np.abs(flipped["color"] - base["color"]).max(0)
    assert diff[39, 0] > 0 and np.count_nonzero(diff) == 1                   # one pixel of the image changed
    again = util.run_oracle(oracle64, rs, rv)
    assert np.array_equal(again["color"], base["color"])


def test_decision_matched_tier_end_to_end_on_the_emulated_kernels(emu, oracle32, oracle64):
    """Seed 140658 (round-5 sweep, 256 Gaussians on 44 x 65 pixels): Gaussian 112 -- radius 50, faint -- owns pixel (0,39) at 255 alpha - 1 = -5.9e-7.
    Both oracles skip it; the kernel's FMA / exp2 form (device AND emulated build) blends it, which moves Gaussian 112's row by a whole pixel's
    contribution and every Gaussian that blends at that pixel by alpha = 1/255 of theirs: 10 of 1024 rotation elements outside the element-wise
    tolerance at a relative L2 of 4.7e-5.  Tiers 1 and 2 fail; the oracles re-run with the other decision at that one pixel agree with the kernel."""
    rs, rv = sweep_scene(140658, emu)
    before = (pc.HATCH["fired"], pc.HATCH["decisions"])
    pc.check_backward(rs, rv, oracle64, oracle32=oracle32)
    assert pc.HATCH["decisions"] > before[1]
    where = pc.HATCH["decision_where"][before[1]:]
    assert all([(i, x, y) for i, x, y, _ in combo] == [(112, 0, 39)] for _, combo, _, _ in where), where
    assert all(rel < 1e-5 for _, _, rel, _ in where)                          # with the decision matched: 6e-7, the fp32 oracle's own level


def test_decision_matched_tier_does_not_rescue_a_defect(emu, oracle32, oracle64, monkeypatch):
    """The same scene with a DEFECT injected into the kernel's result (a Gaussian without any threshold pixel gets a wrong gradient row; and,
    separately, an error spread over all rows): no subset of flipped decisions explains either, the check fails."""
    import pytest
    rs, rv = sweep_scene(140658, emu)
    real = util.run_product

    def broken(row_scale, noise):
        def run(rs_, rv_, dL=None):
            out = real(rs_, rv_, dL)
            if dL is not None:
                g = out["grads"]["rotations"]
                g[54] = g[54] * row_scale                                     # Gaussian 54: large gradient, no pixel near the threshold
                g += noise * np.abs(g).max() * np.random.RandomState(0).randn(*g.shape).astype(g.dtype)
            return out
        return run
    for row_scale, noise in ((1.5, 0.0), (1.0, 3e-3)):
        monkeypatch.setattr(util, "run_product", broken(row_scale, noise))
        with pytest.raises(AssertionError):
            pc.check_backward(rs, rv, oracle64, oracle32=oracle32)
    assert oracle64._thresh is None and oracle32._thresh is None              # the hook is always reset


def test_depth_tie_rule_covers_only_the_pair_s_footprints():
    """The depth-tie class of the raw-parameter sweep (check_raw_entry_random_draw): differing pixels are excused only inside the footprints of BOTH splats of a pair
    whose view depths are within four fp32 ulps; a pixel outside, or a pair eight ulps apart, is a failure."""
    import torch
    from activesplat_amd import synthetic as syn
    W, H = 64, 48
    K = syn.intrinsics(W, H)
    eps = float(np.finfo(np.float32).eps)

    def at(px, py, z):                                      # camera-frame point projecting to pixel (px, py) at depth z
        return [(px - K[0][2]) / K[0][0] * z, (py - K[1][2]) / K[1][1] * z, z]
    for gap, expect in ((1.0, True), (8.0, False)):
        z0 = 2.0
        z1 = float(np.float32(z0 * (1.0 + gap * eps)))
        means = torch.tensor([at(20, 20, z0), at(22, 21, z1), at(50, 30, 3.0)], dtype=torch.float32)
        radius = torch.tensor([5, 6, 4])
        pose = [1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]          # identity: the fp64 transform of the parameters is the parameters
        d = torch.zeros(H, W)
        d[21, 21] = 0.1                                     # inside both footprints
        assert pc._depth_ties_cover(means, pose, means, radius, d, W, H, 1e-3)[0] is expect
        d[30, 50] = 0.1                                     # inside the third splat only: never excused
        ok, covered, total = pc._depth_ties_cover(means, pose, means, radius, d, W, H, 1e-3)
        assert not ok and total == 2 and covered == (1 if expect else 0)
