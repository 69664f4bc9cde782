"""CPU: the tiered gradient rule of tests/parity_cases.py is itself tested -- the threshold-pixel finder on the scene the
round-4 sweep flagged (seed 120013; device analysis in profiles/README.md) and the decision-aware tier on synthetic errors."""
import numpy as np

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


def test_decision_aware_tier_takes_out_proven_rows_only(oracle64):
    rs, rv = sweep_scene(120013, "cpu")
    W, H = int(rs.image_width), int(rs.image_height)
    f = util.run_oracle(oracle64, rs, rv)
    r = np.random.RandomState(0).randn(14329, 3)
    o32 = r * (1 + 1e-4 * np.random.RandomState(1).randn(14329, 3))          # an "fp32 oracle" at 1e-4
    g = r * (1 + 1e-4 * np.random.RandomState(2).randn(14329, 3))
    bad = g.copy(); bad[14305] += 5.0                                        # one threshold-pixel Gaussian carries the whole miss
    assert np.linalg.norm(bad - r) / np.linalg.norm(r) > 1e-3
    before = pc.HATCH["decisions"]
    assert pc.decision_aware("scales", bad, r, o32, f, W, H, min_frac=0.995)
    assert pc.HATCH["decisions"] == before + 1 and pc.HATCH["decision_where"][-1][1][0][:3] == (14305, 71, 18)
    worse = g.copy(); worse[3586] += 5.0                                     # the same miss on a Gaussian with no threshold pixel: a defect
    assert not pc.decision_aware("scales", worse, r, o32, f, W, H, min_frac=0.995)
    spread = g + 0.01 * np.random.RandomState(3).randn(14329, 3)             # an error everywhere is not rescued by taking three rows out
    assert not pc.decision_aware("scales", spread, r, o32, f, W, H, min_frac=0.995)
    pc.HATCH["decisions"] = before; pc.HATCH["decision_where"].pop()         # (keep the session tally for real comparisons)
