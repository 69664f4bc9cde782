"""Cost of one densify event (fused vs the reference's step-by-step call pattern) at N Gaussians, SH degree 3.
usage: N=3000000 python scripts/densify_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util, reference_pattern as RP  # noqa: E402

N = int(os.environ.get("N", 3_000_000))
for fused in (True, False, True):
    torch.manual_seed(0)
    r = util.configs2_optimise_loop(N, 12, "cuda", densify_fn=None if fused else RP.densify_stepwise, densify_every=5, time_it=True)
    print(f"N={N} fused={fused} densify_events_ms={[round(s * 1e3, 3) for s in r['densify_seconds']]} counts={r['counts']} "
          f"iters_per_s={12 / r['seconds']:.1f}", flush=True)
