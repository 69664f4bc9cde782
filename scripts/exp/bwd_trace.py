"""Development helper: where a backward-blend wavefront's cycles go, from a TRACE build (scripts/exp/make_trace_bwd.sh; on the GPU box:
cp activesplat_amd/libgsplat_hip_traceb.so activesplat_amd/libgsplat_hip.so first).  GPU box: N=2000000 SH=3 python scripts/exp/bwd_trace.py"""
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
rv = {k: v.to(dev).requires_grad_(True) for k, v in syn.activate(syn.make_params(N, W, H, seed=0, sh_degree=int(sh) if sh else None)).items()}
dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
for _ in range(4):
    color = GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)[0]
    torch.autograd.grad(color, list(rv.values()) + [m2d], dL)
torch.cuda.synchronize()
lib = _lib.get()
nw = 32768
buf = (C.c_uint64 * (14 * nw))()
lib.gs_debug_bwd_trace.argtypes = [C.c_void_p, C.c_int]
assert lib.gs_debug_bwd_trace(buf, nw) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 14).astype(np.int64)
a = a[a[:, 11] == 1]
t0 = a[:, 0].min()
s, e = (a[:, 0] - t0) * 0.01, (a[:, 1] - t0) * 0.01
cyc = a[:, 2].astype(np.float64)
print("walkers traced %d, kernel span %.1f us, walker duration mean %.1f p90 %.1f max %.1f us; shader clock %.2f GHz" % (
    len(a), e.max(), (e - s).mean(), np.percentile(e - s, 90), (e - s).max(), (cyc / ((e - s) * 1e3)).mean()))
zero = (a[:, 9] >> 40).sum(); staged = ((a[:, 9] >> 20) & 0xfffff).sum(); a[:, 9] &= (1 << 20) - 1
print("  staged records %d, of which with ten zero sums at the flush %d (%.1f %%)" % (staged, zero, 100.0 * zero / max(staged, 1)))
names = ["scan", "round set-up (flags, ballots, lists)", "phase A", "phase B + gather", "flush"]
tot = cyc.sum()
acc = 0.0
for i, n in enumerate(names):
    c = a[:, 3 + i].astype(np.float64).sum(); acc += c
    unit = {0: a[:, 8].sum(), 1: a[:, 9].sum(), 2: a[:, 10].sum(), 3: a[:, 10].sum(), 4: a[:, 9].sum()}[i]
    print("  %-40s %5.1f %% of the walkers' cycles, %7.0f cycles per %s" % (n, 100.0 * c / tot, c / max(unit, 1), ["scan", "round", "batch", "batch", "round"][i]))
epi = (a[:, 13] >> 32).astype(np.float64).sum(); a[:, 13] &= 0xffffffff
pro, wait = a[:, 12].astype(np.float64).sum(), a[:, 13].astype(np.float64).sum()
print("  %-40s %5.1f %% (%.0f cycles per walker)" % ("prologue up to the hand-over wait", 100.0 * pro / tot, pro / len(a)))
print("  %-40s %5.1f %% (%.0f cycles per walker)" % ("hand-over wait + state read", 100.0 * wait / tot, wait / len(a)))
print("  %-40s %5.1f %% (%.0f cycles per walker)" % ("epilogue (hand-over: state, wait for the stores, flag)", 100.0 * epi / tot, epi / len(a)))
print("  %-40s %5.1f %%" % ("rest (loop control)", 100.0 * (tot - acc - pro - wait - epi) / tot))
print("  per walker: %.1f scans, %.1f rounds, %.1f batches; cycles %.0f" % (a[:, 8].mean(), a[:, 9].mean(), a[:, 10].mean(), cyc.mean()))
