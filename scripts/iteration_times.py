"""Development helper: time one ActiveSplat mapping iteration (get_loss + backward + fused Adam) with the
reference's two raster passes vs the fused single pass, on a synthetic scene."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activesplat_amd import mapping as M, optim as O, setup_camera  # noqa: E402
from activesplat_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda")
N, W, H = int(os.environ.get("N", 500_000)), int(os.environ.get("W", 640)), int(os.environ.get("H", 480))
p = syn.make_params(N, W, H, seed=0)
lrs = dict(means3D=1e-4, rgb_colors=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05, log_scales=1e-3, cam_unnorm_rots=0.0, cam_trans=0.0)
MODES = ((False, False, False), (True, False, False), (True, True, False), (True, True, True))
if os.environ.get("ONLY_FUSED"):
    MODES = MODES[-1:]
for fused, floss, finp in MODES:
    params = {k: torch.nn.Parameter(v.to(dev)) for k, v in p.items()}
    params["cam_unnorm_rots"] = torch.nn.Parameter(torch.tensor([[1.0, 0, 0, 0]], device=dev).T.reshape(1, 4, 1).contiguous())
    params["cam_trans"] = torch.nn.Parameter(torch.zeros(1, 3, 1, device=dev))
    variables = {k: torch.zeros(N, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
    im, depth = syn.make_targets(W, H)
    data = dict(cam=setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev), im=im.to(dev), depth=depth.to(dev), id=0,
                w2c=torch.eye(4, device=dev))
    opt = O.initialize_optimizer(params, lrs)

    def it():
        loss, _, _ = M.get_loss(params, data, variables, 0, dict(im=0.5, depth=1.0), fused=fused, fused_loss=floss, fused_inputs=finp,
                                pose7=[1.0, 0, 0, 0, 0, 0, 0] if finp else None)
        loss.backward()
        with torch.no_grad():
            opt.step()
            opt.zero_grad(set_to_none=True)
    for _ in range(5):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        it()
    torch.cuda.synchronize()
    print(f"N={N} {W}x{H} fused_render={fused} fused_loss={floss} fused_inputs={finp}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per mapping iteration")
