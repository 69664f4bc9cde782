"""Development helper: timeline of the streams FORWARD blend's wavefronts from a TRACE build (scripts/exp/make_trace_fwd.sh; on the GPU box:
cp activesplat_amd/libgsplat_hip_tracef.so activesplat_amd/libgsplat_hip.so first).  Prints the wave duration distribution, the load per XCC / CU /
SIMD (sum of its waves' durations) and the residency over the kernel's life.  GPU box: N=2000000 SH=3 python scripts/exp/fwd_trace.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import GaussianRasterizer, setup_camera, _lib  # noqa: E402
from activesplat_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda")
W, H = int(os.environ.get("W", 640)), int(os.environ.get("H", 480))
N = int(os.environ.get("N", 2_000_000))
sh = os.environ.get("SH")
cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev, sh_degree=int(sh) if sh else 0)
rv = {k: v.to(dev) for k, v in syn.activate(syn.make_params(N, W, H, seed=0, sh_degree=int(sh) if sh else None)).items()}
m2d = torch.zeros(N, 3, device=dev)
with torch.no_grad():
    for _ in range(4):
        GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)
torch.cuda.synchronize()
lib = _lib.get()
nw = ((W + 15) // 16) * ((H + 15) // 16) * 4
buf = (C.c_uint64 * (10 * nw))()
lib.gs_debug_fwd_trace.argtypes = [C.c_void_p, C.c_int]
assert lib.gs_debug_fwd_trace(buf, nw) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 10).astype(np.int64)
tick = 0.01                                                  # us per tick (100 MHz constant clock)
s = (a[:, 0] - a[:, 0].min()) * tick; e = (a[:, 1] - a[:, 0].min()) * tick
d = e - s
hw, xcc = a[:, 2], a[:, 3]
simd, cu, shh, se = hw >> 4 & 3, hw >> 8 & 0xf, hw >> 12 & 1, hw >> 13 & 7
print("waves %d, kernel span %.1f us; wave duration mean %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; start p99 %.1f; scans per wave %.1f, rounds %.1f" % (
    nw, e.max(), d.mean(), *np.percentile(d, [10, 50, 90, 99]), d.max(), np.percentile(s, 99), a[:, 4].mean(), a[:, 5].mean()))
cyc, ctr, ntr, cwt = a[:, 8].astype(np.float64), a[:, 6].astype(np.float64), a[:, 7].astype(np.float64), a[:, 9].astype(np.float64)
print("shader clock from the waves' own counters: %.2f GHz (cycles / wall time); cycles per wave mean %.0f; in the trip loops %.0f %% (%.0f cycles per two-entry trip, %.1f trips per wave); waiting for the prefetched records %.1f %%" % (
    (cyc / (d * 1e3)).mean(), cyc.mean(), 100.0 * ctr.sum() / cyc.sum(), ctr.sum() / ntr.sum(), ntr.mean(), 100.0 * cwt.sum() / cyc.sum()))
for name, key in (("XCC", xcc), ("CU", xcc * 1000 + se * 100 + shh * 50 + cu), ("SIMD", (xcc * 1000 + se * 100 + shh * 50 + cu) * 4 + simd)):
    ks = np.unique(key)
    load = np.array([d[key == k].sum() for k in ks]); cnt = np.array([(key == k).sum() for k in ks]); last = np.array([e[key == k].max() for k in ks])
    print("%4s: %4d units, waves per unit %d..%d, summed wave time per unit mean %.0f max %.0f (x%.2f); last wave ends mean %.1f p90 %.1f max %.1f" % (
        name, len(ks), cnt.min(), cnt.max(), load.mean(), load.max(), load.max() / load.mean(), last.mean(), np.percentile(last, 90), last.max()))
for t in np.linspace(0, e.max(), 15):
    print("t=%6.1f us resident waves %5d" % (t, int(((s <= t) & (e > t)).sum())))
