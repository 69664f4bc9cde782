"""The planner's top-down map camera on the HIP path (GPU box): src/visualizer/visualizer.py:923-937 renders the map TWICE per GUI tick through a
camera 1000 m up (:1577-1601) with scale_modifier 0.01 -- the height-cut "free map" with opacity colours and the visible map -- both
forward-only.  Prints JSON: per-render and per-tick milliseconds, per-stage hipEvent averages, and the blend forward with the few-tile
(producer / consumer) kernel forced on for this 437 / 529-tile view against the plain streams kernel (which the 257..768-tile band uses)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from activesplat_amd import GaussianRasterizer, _lib, rasterizer as R  # noqa: E402
from tests import parity_cases as pc  # noqa: E402

dev = torch.device("cuda")
N = int(os.environ.get("N", 1_000_000))
lib = _lib.get()
out = {"gaussians": N}
for W, H in ((360, 300), (368, 368)):
    rs, rv = pc.topdown_scene(N, dev, W=W, H=H)
    rs = rs._replace(debug=False)
    white = rs._replace(bg=torch.ones(3, device=dev))
    m2d = torch.zeros(N, 3, device=dev)
    # the free map: Gaussians between the agent's foot and head (here: half of them), coloured by opacity (GaussianColorType.Opacity)
    keep = rv["means3D"][:, 1] > -0.7
    free = {k: v[keep].contiguous() for k, v in rv.items()}
    free["colors_precomp"] = free["opacities"].expand(-1, 3).contiguous()
    m2f = torch.zeros(free["means3D"].shape[0], 3, device=dev)

    def tick():
        with torch.no_grad():
            GaussianRasterizer(raster_settings=rs)(means2D=m2f, **free)
            GaussianRasterizer(raster_settings=white)(means2D=m2d, **rv)

    def one():
        with torch.no_grad():
            return GaussianRasterizer(raster_settings=white)(means2D=m2d, **rv)

    def bench(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    res = {"tiles": ((W + 15) // 16) * ((H + 15) // 16)}
    res["visible_map_ms"] = round(bench(one), 4)
    res["tick_two_renders_ms"] = round(bench(tick), 4)
    col, radii, depth, op = one()
    res["D"] = int(R.last_stats["num_rendered"]); res["max_tile_list"] = int(R.last_stats["max_tile_instances"])
    res["visible"] = int((radii > 0).sum()); res["radii_values"] = sorted(int(v) for v in torch.unique(radii).tolist())
    res["opacity_mean"] = round(float(op.mean()), 4)
    for name, knob in (("plain_streams_kernel", 256), ("few_tile_pc_kernel_forced", 4096)):
        _lib.check(lib.gs_set_half_quadrants(knob))
        try:
            lib.gs_profile_enable(1)
            for _ in range(20):
                one()
            torch.cuda.synchronize()
            res["stages_us_" + name] = {k: round(ms / c * 1e3, 1) for k, (ms, c) in _lib.profile_collect().items() if c}
            lib.gs_profile_enable(0)
            res["ms_" + name] = round(bench(one), 4)
            c2 = one()[0]
            res["same_image_" + name] = bool(torch.equal(c2, col))
        finally:
            _lib.check(lib.gs_set_half_quadrants(256))
    out[f"{W}x{H}"] = res
print(json.dumps(out))
