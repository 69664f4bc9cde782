"""Development helper (uses the oracle: a checker run, not product code): the random sweeps of scripts/exp/fuzz_gpu.py on the HOST-EMULATED kernels
(tests/hipemu) -- no GPU minutes, so the seed ranges can be wide.  The emulated build compiles the same kernel sources; what it does not share with the
device build is the compiler's arithmetic (v_exp_f32, fused multiply-adds where the sources allow them), so a scene flagged here is a logic or
conditioning finding, and a clean range says nothing about the device's last bit (the GPU sweeps do)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.gs_oracle import Oracle
from tests import parity_cases as pc, util
from tests.fuzz_scenes import hard_scene, sweep_scene

torch.set_num_threads(1)
o32, o64 = Oracle("f32"), Oracle("f64")
util.use_emulated_kernels(os.path.join(ROOT, "tests", "hipemu", "libgsplat_emu.so"))
n0, n1 = int(os.environ.get("SEED0", 400000)), int(os.environ.get("SEED1", 400300))
bad = []
for seed in range(n0, n1):
    try:
        if os.environ.get("HARD"):
            rs, rv = hard_scene(seed, "cpu", float(os.environ["HARD"]))
        else:
            rs, rv = sweep_scene(seed, "cpu")
        pc.check_forward(rs, rv, o32, oracle64=o64 if os.environ.get("HARD") else None)
        pc.check_backward(rs, rv, o64, oracle32=o32)
        if seed % 3 == 1 and "colors_precomp" in rv and "cov3D_precomp" not in rv:
            pc.check_fused_rgbd(rs, rv, o64, seed=seed, oracle32=o32)
    except Exception as e:
        bad.append((seed, repr(e)[:300]))
        print("FAIL seed", seed, repr(e)[:300], flush=True)
print("emulated kernels, seeds %d..%d%s: %d failures; forward fp32-oracle tier fired %d times" % (n0, n1, " HARD=" + os.environ["HARD"] if os.environ.get("HARD") else "", len(bad), pc.HATCH.get("forward_fired", 0)))
print("fp32 escape hatch: fired %d times in %d gradient comparisons; decision-matched comparison decided %d times" % (pc.HATCH["fired"], pc.HATCH["keys_checked"], pc.HATCH["decisions"]))
print("failures:", bad)
