"""Development helper: the configs[4]-substitute mapper harness with every fused path on, per-frame timing breakdown."""
import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import synthetic as syn
from activesplat_amd.mapper import SplatMapper
dev = torch.device("cuda")
W, H, FR = 256, 256, 31
gt = syn.shell_scene(400_000, seed=2, W=W, H=H)
gt["logit_opacities"] = gt["logit_opacities"] + 3.0
seq = list(syn.orbit_sequence(gt, FR, W, H, dev))
for name, flags in (("fused render/loss/inputs", dict(fused_render=True, fused_loss=True, fused_inputs=True)),
                    ("+ fused growth + keyframes", dict(fused_render=True, fused_loss=True, fused_inputs=True, fused_growth=True, fused_keyframes=True)),
                    ("+ raw parameters, Adam in the backward, no autograd", dict(fused_render=True, fused_loss=True, fused_inputs=True, fused_growth=True, fused_keyframes=True,
                                                                                 fused_preprocess=True, fused_adam=True, fused_iteration=True))):
    for rep in range(2):
        mp = SplatMapper(syn.intrinsics(W, H), W, H, config=dict(step_num=FR, mapping_iters=10, **flags), device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for fr in seq:
            mp.run(fr)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "%.1f ms total, %.3f ms/iteration, N=%d" % (dt * 1e3, mp.stats["iter_time"] / mp.stats["iters"] * 1e3, mp.params["means3D"].shape[0]), flush=True)
mp = SplatMapper(syn.intrinsics(W, H), W, H, config=dict(step_num=FR, mapping_iters=10, **flags), device=dev)
pr = cProfile.Profile(); pr.enable()
for fr in seq:
    mp.run(fr)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(35)
