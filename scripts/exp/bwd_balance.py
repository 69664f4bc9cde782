"""Development helper: how uneven are the quadrant walks of the backward blend, and what would longest-first dispatch buy?  Per quadrant the walk
length is the deepest contributor of its 64 pixels (n_contrib); list scheduling of the 4800 walks over the resident slots (3 per SIMD x 1024) is
simulated with a duration proportional to the chunks walked (+ a constant), in the kernel's dispatch order and longest-first.
GPU box: N=2000000 python scripts/exp/bwd_balance.py"""
import heapq, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import GaussianRasterizer, setup_camera  # noqa: E402
from activesplat_amd import synthetic as syn  # noqa: E402
from tests import util  # noqa: E402

dev = torch.device("cuda")
W, H = 640, 480
N = int(os.environ.get("N", 2_000_000))
sh = int(os.environ.get("SH", 3))
cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev)._replace(debug=True)
rv = {k: v.to(dev) for k, v in syn.activate(syn.make_params(N, W, H, seed=0)).items()}
with torch.no_grad():
    GaussianRasterizer(raster_settings=cam)(means2D=torch.zeros(N, 3, device=dev), **rv)
art = util.artefacts()
nc = art["n_contrib"].reshape(H, W).astype(np.int64)
gx, gy = W // 16, H // 16
q = nc.reshape(gy, 2, 8, gx, 2, 8).max(axis=(2, 5))            # [ty, qy, tx, qx] deepest contributor of every 8x8 quadrant
wmax = q.transpose(0, 2, 1, 3).reshape(gy * gx, 4)              # [tile, quadrant]
chunks = (wmax + 63) // 64
print("quadrant walk length in 64-record chunks: mean %.1f  p50 %d  p90 %d  p99 %d  max %d" % (chunks.mean(), np.percentile(chunks, 50), np.percentile(chunks, 90),
      np.percentile(chunks, 99), chunks.max()))
tiles = gx * gy
per = (tiles + 7) // 8
# the kernel's dispatch order: block b -> XCD b & 7, entry b >> 3 of that XCD's band: tile = band * per + idx / 4, quadrant idx % 4
order = []
for b in range(8 * per * 4):
    band, idx = b & 7, b >> 3
    t = band * per + idx // 4
    if idx // 4 < per and t < tiles:
        order.append((t, idx % 4))
dur = lambda t, qd, c0: c0 + float(chunks[t, qd])              # noqa: E731


def makespan(seq, slots, c0):
    heap = [0.0] * slots
    heapq.heapify(heap)
    end = 0.0
    for t, qd in seq:
        s = heapq.heappop(heap)
        e = s + dur(t, qd, c0)
        end = max(end, e)
        heapq.heappush(heap, e)
    return end


for c0 in (0.0, 2.0):
    total = sum(dur(t, qd, c0) for t, qd in order)
    for slots in (3072, 2048):
        nat = makespan(order, slots, c0)
        lpt = makespan(sorted(order, key=lambda x: -chunks[x[0], x[1]]), slots, c0)
        print("c0=%.0f slots=%d: ideal %.1f  dispatch order %.1f (x%.3f)  longest first %.1f (x%.3f)" % (c0, slots, total / slots, nat, nat / (total / slots), lpt, lpt / (total / slots)))
