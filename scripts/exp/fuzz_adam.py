"""Development helper: random sweep of the Adam step inside the per-Gaussian backward (render_rgbd_raw(adam=optimizer): gs_render_backward_raw_adam) against
backward + GaussianAdam.step() -- parity_cases.check_adam_inside_the_backward on random maps, image sizes and poses instead of the suite's one scene; colours / SH rows /
isotropic maps in every draw.   GPU box: SEED0=0 SEED1=300 python scripts/exp/fuzz_adam.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import parity_cases as pc

bad = []
for seed in range(int(os.environ.get("SEED0", 0)), int(os.environ.get("SEED1", 300))):
    r = np.random.RandomState(seed)
    n = int(r.choice([int(r.randint(50, 2000)), int(r.randint(2000, 30000))]))
    W, H = (int(r.randint(24, 200)), int(r.randint(24, 160))) if seed % 3 else (int(r.randint(272, 420)), int(r.randint(256, 330)))
    a = float(r.uniform(-0.4, 0.4))
    pose = [float(np.cos(a / 2)), 0.0, float(np.sin(a / 2)), 0.0, float(r.uniform(-0.2, 0.2)), float(r.uniform(-0.1, 0.1)), float(r.uniform(-1.4, 0.2))]
    try:
        pc.check_adam_inside_the_backward("cuda", n=n, W=W, H=H, steps=2, exact=False, seed=seed, pose=pose, visible=(0.0, 1.01))
    except Exception as e:
        bad.append(seed)
        print("FAIL seed", seed, "n", n, f"{W}x{H}", repr(e)[:260], flush=True)
print("Adam-inside-the-backward sweep: seeds %s..%s, %d failures %s" % (os.environ.get("SEED0", 0), os.environ.get("SEED1", 300), len(bad), bad[:30]))
