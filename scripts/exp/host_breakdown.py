"""Development helper: what the host pays per drop-in forward + backward, piece by piece (tiny scene: GPU work negligible).
GPU box: python scripts/exp/host_breakdown.py"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import GaussianRasterizer, setup_camera, _lib, rasterizer as R
from activesplat_amd import synthetic as syn

dev = torch.device("cuda")
N, W, H = 2000, 120, 150
cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev)
rv = {k: v.to(dev).requires_grad_(True) for k, v in syn.activate(syn.make_params(N, W, H, seed=0)).items()}
dL = torch.randn(3, H, W, device=dev)
m2d = torch.zeros(N, 3, device=dev, requires_grad=True)


def bench(f, n=400, warm=40):
    for _ in range(warm):
        f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


class Dummy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c, d, e, f, g, h, rs):
        ctx.save_for_backward(a, c, d, e, f)
        ctx.set_materialize_grads(False)
        out = torch.empty(3, H, W, device=a.device)
        r = torch.empty(N, dtype=torch.int32, device=a.device)
        ctx.mark_non_differentiable(r)
        return out, r

    @staticmethod
    def backward(ctx, go, _r=None):
        a, c, d, e, f = ctx.saved_tensors
        z = lambda *s: torch.empty(*s, device=a.device)
        return z(N, 3), z(N, 3), None, z(N, 3), z(N, 1), z(N, 3), z(N, 4), None, None


def full():
    color = GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)[0]
    torch.autograd.grad(color, list(rv.values()) + [m2d], dL)


def fwd_nograd():
    with torch.no_grad():
        GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)


def fwd_grad():
    GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)


def dummy():
    out, _ = Dummy.apply(rv["means3D"], m2d, None, rv["colors_precomp"], rv["opacities"], rv["scales"], rv["rotations"], None, cam)
    torch.autograd.grad(out, list(rv.values()) + [m2d], dL, allow_unused=True)


def dummy_fwd():
    Dummy.apply(rv["means3D"], m2d, None, rv["colors_precomp"], rv["opacities"], rv["scales"], rv["rotations"], None, cam)


print(f"drop-in forward + backward             {bench(full):7.1f} us")
print(f"drop-in forward, grad mode             {bench(fwd_grad):7.1f} us")
print(f"drop-in forward, no_grad               {bench(fwd_nograd):7.1f} us")
print(f"empty autograd.Function fwd + bwd      {bench(dummy):7.1f} us   (torch's own floor for a Python Function of this signature)")
print(f"empty autograd.Function fwd            {bench(dummy_fwd):7.1f} us")
print(f"11 x torch.empty                       {bench(lambda: [torch.empty(1000, device=dev) for _ in range(11)]):7.1f} us")
print(f"torch.cuda.current_stream              {bench(lambda: torch.cuda.current_stream(dev)):7.1f} us")
ev_s = torch.cuda.current_stream(dev)
def evt():
    e = torch.cuda.Event(); e.record(ev_s); e.synchronize()
print(f"Event() + record + synchronize (idle)  {bench(evt):7.1f} us")
lib = _lib.get()
print(f"ctypes call gs_abi_version             {bench(lambda: lib.gs_abi_version()):7.1f} us")
gl = _lib.GsGeomLayout()
print(f"ctypes call gs_geom_layout (4 args)    {bench(lambda: lib.gs_geom_layout(N, W, H, C.byref(gl))):7.1f} us")
t = rv["means3D"]
print(f"25 x C.c_void_p(t.data_ptr())          {bench(lambda: [C.c_void_p(t.data_ptr()) for _ in range(25)]):7.1f} us")
print(f"_camera()                              {bench(lambda: R._camera(cam, dev, 0)):7.1f} us")
print(f"GaussianRasterizer(...) construction   {bench(lambda: GaussianRasterizer(raster_settings=cam)):7.1f} us")
hn = torch.zeros(2, dtype=torch.int32).pin_memory()
print(f"2 x pinned .item()                     {bench(lambda: (int(hn[0].item()), int(hn[1].item()))):7.1f} us")
# GPU time of the same frame (events around it)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
lib.gs_profile_enable(1)
for _ in range(100):
    full()
torch.cuda.synchronize()
prof = _lib.profile_collect(); lib.gs_profile_enable(0)
print("GPU stage sum per frame: %.1f us" % sum(ms / c * 1e3 for ms, c in prof.values() if c), {k: round(ms / c * 1e3, 1) for k, (ms, c) in prof.items() if c})
