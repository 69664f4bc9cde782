"""Development helper: per-stage hipEvent timings of the C2 workload under environment-variable
kernel variants (GS_*_VARIANT), all in one process.  Usage: python scripts/stage_times.py VAR=v1,v2 ..."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activesplat_amd import GaussianRasterizer, _lib, setup_camera  # noqa: E402
from activesplat_amd import synthetic as syn  # noqa: E402


def run(N=500_000, W=640, H=480, steps=int(os.environ.get("STEPS", 30)), warmup=int(os.environ.get("WARMUP", 5)), backward=True):
    dev = torch.device("cuda")
    sh = os.environ.get("SH")
    cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev, sh_degree=int(sh) if sh else 0)
    prm = syn.make_params(N, W, H, seed=0, sh_degree=int(sh) if sh else None)
    if os.environ.get("CULL"):                              # this fraction of the Gaussians, picked at random, lies behind the camera (the bench scene is 100 % visible)
        hide = torch.rand(N, generator=torch.Generator().manual_seed(3)) < float(os.environ["CULL"])
        prm["means3D"][hide, 2] = -prm["means3D"][hide, 2]
    rv = {k: v.to(dev).requires_grad_(True) for k, v in syn.activate(prm).items()}
    dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    lib = _lib.get()
    if os.environ.get("CHAIN"):                             # gs_set_backward_chain: pieces of a chained backward walk (1 = one walker per quadrant)
        lib.gs_set_backward_chain(int(os.environ["CHAIN"]), int(os.environ.get("CHAIN_MIN_TILES", -1)))
    if os.environ.get("TICKETS"):                           # gs_set_backward_chain_tickets: ordered tickets of the chained walks (A/B)
        lib.gs_set_backward_chain_tickets(int(os.environ["TICKETS"]))
    if os.environ.get("SEGS"):                              # gs_set_backward_segments: list segments per quadrant in the few-tile backward
        lib.gs_set_backward_segments(int(os.environ["SEGS"]))
    if os.environ.get("HALF"):                              # gs_set_half_quadrants: images of at most this many tiles use half-quadrant wavefronts
        lib.gs_set_half_quadrants(int(os.environ["HALF"]))

    rgbd = bool(os.environ.get("RGBD"))                     # fused single-pass RGB-D render with a depth gradient (DEPTH_GRAD kernels)
    dLd = torch.randn(1, H, W, generator=torch.Generator().manual_seed(2)).to(dev)

    def step():
        m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
        if rgbd:
            from activesplat_amd.rasterizer import render_rgbd
            color, _, depth, _, _ = render_rgbd(cam, means2D=m2d, **rv)
            if backward:
                torch.autograd.grad([color, depth], list(rv.values()) + [m2d], [dL, dLd])
            return
        color, _, _, _ = GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)
        if backward:
            torch.autograd.grad(color, list(rv.values()) + [m2d], dL)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e6
    lib.gs_profile_enable(1)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    lib.gs_profile_enable(0)
    return wall, {k: ms / c * 1e3 for k, (ms, c) in prof.items() if c}


if __name__ == "__main__":
    sweeps = [a.split("=") for a in sys.argv[1:] if "=" in a]
    N = int(os.environ.get("N", 500_000))
    if not sweeps:
        sweeps = [["GS_NONE", "0"]]
    for var, vals in sweeps:
        for v in vals.split(","):
            os.environ[var] = v
            wall, st = run(N=N, W=int(os.environ.get("W", 640)), H=int(os.environ.get("H", 480)), backward=os.environ.get("BWD", "1") != "0")
            from activesplat_amd import rasterizer as R
            print(f"{var}={v} N={N} D={R.last_stats['num_rendered']} maxtile={R.last_stats.get('max_tile_instances')} wall_us={wall:.1f} " + " ".join(f"{k}={u:.1f}" for k, u in st.items()), flush=True)
        os.environ.pop(var, None)
