"""Development helper: configs[2]'s 100-iteration optimise loop, raw-parameter rasteriser on / off.  GPU box: python scripts/exp/c2_loop.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd.workloads import configs2_optimise_loop
for raw, adam in ((True, False), (True, True), (True, False), (True, True), (False, False)):
    torch.cuda.empty_cache()
    r = configs2_optimise_loop(2_000_000, 100, "cuda", time_it=True, raw=raw, fused_adam=adam)
    print("raw=%s adam_in_backward=%s: %.1f iterations/s, densify events %s ms, counts %s, loss %s" % (raw, adam, 100 / r["seconds"], [round(x * 1e3, 2) for x in r["densify_seconds"]], r["counts"],
          [round(x, 4) for x in r["losses"]]), flush=True)
