"""Development helper: per-Gaussian kernel variants (GS_PRE_FWD / GS_PRE_BWD) at 2 M / SH-3 -- results must be identical to the bit,
stage times by hipEvents.  GPU box: python scripts/exp/pre_variants.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import GaussianRasterizer, _lib, setup_camera  # noqa: E402
from activesplat_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda")
N, W, H = int(os.environ.get("N", 2_000_000)), 640, 480
deg = int(os.environ.get("SH", 3))
cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev, sh_degree=deg)
rv = {k: v.to(dev).requires_grad_(True) for k, v in syn.activate(syn.make_params(N, W, H, seed=0, sh_degree=deg)).items()}
dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
lib = _lib.get()
m2d = torch.zeros(N, 3, device=dev, requires_grad=True)


def step():
    out = GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)
    g = torch.autograd.grad(out[0], list(rv.values()) + [m2d], dL)
    return out, g


def measure(tag, steps=40):
    for _ in range(5):
        step()
    lib.gs_profile_enable(1)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    lib.gs_profile_enable(0)
    print(tag, " ".join(f"{k}={ms / c * 1e3:.1f}" for k, (ms, c) in prof.items() if c), "sum=%.1f" % sum(ms / c * 1e3 for ms, c in prof.values() if c), flush=True)


variants = [dict(), dict(GS_PRE_FWD="1"), dict(GS_PRE_BWD="1"), dict(GS_PRE_FWD="1", GS_PRE_BWD="1")]
ref = None
for v in variants:
    for k in ("GS_PRE_FWD", "GS_PRE_BWD"):
        os.environ.pop(k, None)
    os.environ.update(v)
    out, g = step()
    torch.cuda.synchronize()
    cur = [t.clone() for t in out] + [t.clone() for t in g]
    if ref is None:
        ref = cur
    else:
        same = [bool(torch.equal(a, b)) for a, b in zip(ref[:4], cur[:4])]
        rel = [float((a - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(cur[4:], ref[4:])]
        print(v, "forward outputs identical:", same, "max grad rel diff: %.2e" % max(rel), flush=True)
    measure(str(v))
