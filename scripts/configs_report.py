"""BASELINE configs[2] (optimise loop) and the configs[4] substitute (mapper harness on a synthetic RGB-D spin):
numbers for profiles/.  Run on the GPU box:  python scripts/configs_report.py > gpurun_out/configs.json"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activesplat_amd import mapping as M, optim as O, rasterizer as R, setup_camera  # noqa: E402
from activesplat_amd import synthetic as syn  # noqa: E402
from activesplat_amd.mapper import SplatMapper  # noqa: E402

dev = torch.device("cuda")
out = {}

# ---- configs[2]: 2M Gaussians, SH degree 3, 640x480, 100-iteration optimise loop, fused Adam + fused densify ----
from tests import util  # noqa: E402
N, ITERS = int(os.environ.get("N", 2_000_000)), 100
util.configs2_optimise_loop(4096, 12, "cuda", densify_every=5)          # one-time code-object loads of every kernel / torch op involved
torch.manual_seed(0)
r = util.configs2_optimise_loop(N, ITERS, "cuda", fused_densify=True, time_it=True)
out["configs2_optimise_loop"] = dict(gaussians_start=N, gaussians_after_densify=r["counts"], sh_degree=3, iters=ITERS, seconds=round(r["seconds"], 4),
                                     iters_per_s=round(ITERS / r["seconds"], 2), loss_first=r["losses"][0], loss_last=r["losses"][1],
                                     densify_event_ms=[round(x * 1e3, 3) for x in r["densify_seconds"]],
                                     densify="every 50 iterations (grad_thresh 2e-4, split into 2, opacity cull 0.005), fused: one classification kernel, "
                                             "one index, one gather per tensor",
                                     note="fused activations + single-pass RGB-D render (SH-3) + fused loss + fused Adam (5 tensors incl. shs)")
torch.cuda.empty_cache()

# ---- configs[4] substitute: mapper harness on a synthetic RGB-D spin (no Habitat / Gibson / ROS here) ----
W, H, FR = 256, 256, 31
gt = syn.shell_scene(400_000, seed=2, W=W, H=H)
gt["logit_opacities"] = gt["logit_opacities"] + 3.0
seq = list(syn.orbit_sequence(gt, FR, W, H, dev))
rep = {}
for name, flags in (("reference_call_pattern", {}), ("fused", dict(fused_render=True, fused_loss=True, fused_inputs=True, fused_preprocess=True, fused_growth=True, fused_keyframes=True))):
    mp = SplatMapper(syn.intrinsics(W, H), W, H, config=dict(step_num=FR, mapping_iters=10, **flags), device=dev)   # high-res setting: 2 iters/frame
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for fr in seq:
        mp.run(fr)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ps = []
    for fr in seq[::5]:
        im, depth, opacity = mp.render_rgbd(fr["w2c"])
        seen = (fr["depth"] > 0)[0]
        mse = float(((im[:, seen] - fr["color"][:, seen]) ** 2).mean())
        ps.append(10 * np.log10(1.0 / mse))
    rep[name] = dict(frames=FR, seconds=round(dt, 3), iterations=mp.stats["iters"], ms_per_iteration=round(mp.stats["iter_time"] / max(mp.stats["iters"], 1) * 1e3, 3),
                     gaussians=int(mp.params["means3D"].shape[0]), psnr_db_vs_synthetic_gt=round(float(np.mean(ps)), 2))
out["configs4_substitute_mapper_harness"] = dict(rep, note="synthetic in-place spin (10 degree turns) inside a 400k-Gaussian ground-truth scene, 256x256; "
                                                 "Habitat-sim, Gibson data, ROS and a reference checkpoint are unavailable (SURVEY 8d)")
print(json.dumps(out, indent=1))
