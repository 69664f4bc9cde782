"""Planner panorama timing (GPU box): reference call pattern vs fused single-stream vs fused multi-stream."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activesplat_amd import lookaround as LA, synthetic as syn  # noqa: E402
dev = torch.device("cuda")
N = int(os.environ.get("N", 1_000_000))
params = {k: v.to(dev) for k, v in syn.shell_scene(N, seed=2, W=LA.LOOK_W, H=LA.LOOK_H).items()}
c2w = np.eye(4)
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print(f"N={N}: reference call pattern {bench(lambda: LA.look_around(params, c2w, fused=False)):.3f} ms / panorama")
print(f"N={N}: fused, one pass per view {bench(lambda: LA.look_around(params, c2w, fused=True, batched=False)):.3f} ms / panorama")
print(f"N={N}: fused, multi-view atlas {bench(lambda: LA.look_around(params, c2w, fused=True, batched=True)):.3f} ms / panorama")
_cuda = torch.device("cuda").type
class _One:  # single-stream variant: pretend there is one view per call
    pass
t1 = bench(lambda: [LA.look_around(params, LA.rot_axis(c2w, 'y', np.deg2rad(120 * i)), fused=True, views=1) for i in range(3)])
print(f"N={N}: fused, three single-view calls {t1:.3f} ms / panorama")
from activesplat_amd import _lib, rasterizer as R
lib = _lib.get(); lib.gs_profile_enable(1)
for _ in range(10):
    LA.look_around(params, c2w, fused=True)
torch.cuda.synchronize()
print({k: round(ms / c * 1e3, 1) for k, (ms, c) in _lib.profile_collect().items() if c}, R.last_stats)
