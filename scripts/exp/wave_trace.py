"""Development helper: timeline of the backward blend's wavefronts from a TRACE build of the library (blend.hip with a start / end timestamp and
timestamps per workgroup in a __device__ array, exported as gs_debug_wave_trace; scripts/exp/make_trace_lib.sh builds it; on the GPU box: cp activesplat_amd/libgsplat_hip_trace.so activesplat_amd/libgsplat_hip.so first).  Prints the wave duration distribution
and the number of resident wavefronts over the kernel's life.  GPU box: python scripts/exp/wave_trace.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import GaussianRasterizer, setup_camera, _lib  # noqa: E402
from activesplat_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda")
W, H = 640, 480
N = int(os.environ.get("N", 2_000_000))
cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev)
rv = {k: v.to(dev).requires_grad_(True) for k, v in syn.activate(syn.make_params(N, W, H, seed=0)).items()}
dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
if os.environ.get("CHAIN"):
    _lib.get().gs_set_backward_chain(int(os.environ["CHAIN"]), -1)
for _ in range(6):
    color = GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)[0]
    torch.autograd.grad(color, list(rv.values()) + [m2d], dL)
torch.cuda.synchronize()
lib = _lib.get()
pieces = int(os.environ.get("CHAIN", 1))
nb = 4800 * pieces
buf = (C.c_uint64 * (5 * nb))()
lib.gs_debug_wave_trace.argtypes = [C.c_void_p, C.c_int]
assert lib.gs_debug_wave_trace(buf, nb) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 5).astype(np.int64)
base = a[:, 0].min()
tick = 0.01                                                  # us per tick (100 MHz constant clock)
s, e, tl, tw = [(a[:, i] - base) * tick for i in range(4)]
print("waves", nb, "kernel span %.1f us" % e.max())
for p_ in range(pieces):
    sl = slice(p_ * 4800, (p_ + 1) * 4800)
    print("piece %d: start p10 %.1f p50 %.1f p90 %.1f | prologue (start -> before the wait) mean %.2f p90 %.2f | wait mean %.2f p90 %.2f max %.1f | walk mean %.1f p90 %.1f | whole mean %.1f"
          % (p_, *np.percentile(s[sl], [10, 50, 90]), (tl[sl] - s[sl]).mean(), np.percentile(tl[sl] - s[sl], 90), (tw[sl] - tl[sl]).mean(),
             np.percentile(tw[sl] - tl[sl], 90), (tw[sl] - tl[sl]).max(), (e[sl] - tw[sl]).mean(), np.percentile(e[sl] - tw[sl], 90), (e[sl] - s[sl]).mean()))
for t in np.linspace(0, e.max(), 21):
    print("t=%6.1f us resident waves %5d" % (t, int(((s <= t) & (e > t)).sum())))
