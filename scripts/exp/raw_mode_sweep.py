"""Development helper (checker run): the raw-parameter rasteriser against the activation kernels and the torch ops over a few sizes.
GPU box: python scripts/exp/raw_mode_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import parity_cases as pc
bad = 0
for n in (300, 1000, 5000, 20000, 60000):
    try:
        pc.check_raw_parameter_mode("cuda", n=n)
        pc.check_raw_parameter_mode_sh("cuda", n=n, W=96 + n // 500, H=80 + n // 700)
        print("n =", n, "ok", flush=True)
    except Exception as e:
        bad += 1
        print("n =", n, "FAIL", repr(e)[:300], flush=True)
print("failures:", bad)
