"""Do small-image forwards on different streams overlap?  GPU box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import lookaround as LA, synthetic as syn, GaussianRasterizer, setup_camera  # noqa: E402
dev = torch.device("cuda")
N = int(os.environ.get("N", 1_000_000))
params = {k: v.to(dev) for k, v in syn.shell_scene(N, seed=2, W=LA.LOOK_W, H=LA.LOOK_H).items()}
rv = LA._world_rendervar(params)
cams = [setup_camera(LA.LOOK_W, LA.LOOK_H, LA.look_around_k(), np.linalg.inv(LA.rot_axis(np.eye(4), "y", np.deg2rad(120 * i))), 0.01, 100.0, device=dev, bg=(1, 1, 1)) for i in range(3)]
def one(i):
    with torch.no_grad():
        return GaussianRasterizer(raster_settings=cams[i])(**rv)
for i in range(3): one(i)
torch.cuda.synchronize()
def timed(fn, n=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("3 views, one stream   : %.3f ms" % timed(lambda: [one(i) for i in range(3)]))
pool = [torch.cuda.Stream() for _ in range(3)]
def multi():
    outs = []
    for i in range(3):
        with torch.cuda.stream(pool[i]):
            outs.append(one(i))
    return outs
print("3 views, three streams: %.3f ms" % timed(multi))
# host time of one call (enqueue only)
torch.cuda.synchronize(); t = time.perf_counter(); one(0); h = time.perf_counter() - t; torch.cuda.synchronize()
print("host time of one forward call: %.3f ms" % (h * 1e3))
