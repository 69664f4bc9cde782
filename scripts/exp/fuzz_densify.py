"""Development helper: the fused densify / prune against the step-by-step call pattern over many map sizes and seeds on the GPU."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import _lib
_lib.get()
from tests.test_golden import check_fused_densify_and_prune
bad = 0
r = np.random.RandomState(0)
sizes = [1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4097] + [int(v) for v in r.randint(300, 200_000, 40)]
for i, N in enumerate(sizes):
    for iso in (True, False):
        try:
            check_fused_densify_and_prune("cuda", iso, N, 100 + i)
        except Exception as e:
            bad += 1
            print("FAIL N", N, "iso", iso, repr(e)[:300], flush=True)
print("%d sizes x 2: %d failures" % (len(sizes), bad))
