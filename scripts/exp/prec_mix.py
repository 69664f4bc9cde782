"""Development helper (checker only: runs the two oracle builds, no product code): WHERE does an fp32 rasteriser lose the stated gradient tolerance on the scenes that
the fp32-oracle tier decides?  The 2-D gradients of the per-pixel replay are mixed across precisions with the per-Gaussian chain to the 3-D parameters.
usage: python scripts/exp/prec_mix.py 122050 2
(A third variant -- fp32 replay with the sums over the pixels in fp64: the oracle source with the five accumulator arrays of gso_blend_backward retyped to double, built
to a scratch library -- gave the all-fp32 error to three digits: the loss is per pixel, not in the sums.)"""
import sys, numpy as np, torch, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.gs_oracle import Oracle, _ptr
from tests import util
from tests.fuzz_scenes import sweep_scene
seed, plain = int(sys.argv[1]), (sys.argv[2] if len(sys.argv) > 2 else None)
rs, rv = sweep_scene(seed, "cpu", plain)
H, W = int(rs.image_height), int(rs.image_width)
g = torch.Generator().manual_seed(seed)
dLc = torch.randn(3, H, W, generator=g).numpy(); dLd = torch.randn(1, H, W, generator=g).numpy()
o32, o64 = Oracle("f32"), Oracle("f64")
o32.set_threads(16); o64.set_threads(16)
f64 = util.run_oracle(o64, rs, rv); f32 = util.run_oracle(o32, rs, rv)
g64 = o64.backward(f64, dLc, dLd); g32 = o32.backward(f32, dLc, dLd)
def chain(o, fwd, dxy, dconic, dfeat, dz):
    ctx = fwd["_ctx"]; c = ctx["c"]; P = c.P; r = o.real
    gg = dict(means2D=np.zeros((P, 3), r), means3D=np.zeros((P, 3), r), scales=np.zeros((P, 3), r), rotations=np.zeros((P, 4), r),
              cov3D_precomp=np.zeros((P, 6), r), shs=np.zeros((P, max(c.sh_coeffs, 1), 3), r), colors_precomp=np.zeros((P, 3), r))
    a = lambda x: np.ascontiguousarray(np.asarray(x, dtype=r))
    dxy, dconic, dfeat, dz = a(dxy), a(dconic), a(dfeat), a(dz)
    o.lib.gso_preprocess_backward(C.byref(c), _ptr(ctx["means3D"]), _ptr(ctx["shs"]), _ptr(ctx["scales"]), _ptr(ctx["rots"]), _ptr(fwd["radii"]), _ptr(fwd["cov3d"]),
                                  _ptr(fwd["clamped"]), _ptr(dxy), _ptr(dconic), _ptr(dfeat), _ptr(dz), _ptr(gg["means2D"]), _ptr(gg["means3D"]), _ptr(gg["scales"]),
                                  _ptr(gg["rotations"]), _ptr(gg["cov3D_precomp"]), _ptr(gg["shs"]), _ptr(gg["colors_precomp"]))
    return gg
# need dz: re-run blend backward to get it -- the wrapper keeps _dxy/_dconic/_dfeat only; recompute dz through a patched call
def blend(o, fwd):
    ctx = fwd["_ctx"]; c = ctx["c"]; P = c.P; r = o.real
    dpix = np.ascontiguousarray(dLc.astype(r)).reshape(3, c.H, c.W); ddep = np.ascontiguousarray(dLd.astype(r)).reshape(c.H, c.W)
    dz = np.zeros(P, r); dxy = np.zeros((P, 2), r); dconic = np.zeros((P, 3), r); dop = np.zeros(P, r); dfeat = np.zeros((P, 3), r)
    o.lib.gso_blend_backward(C.byref(c), 3, _ptr(ctx["bg"]), _ptr(fwd["ranges"]), _ptr(fwd["ids_sorted"]), _ptr(fwd["xy"]), _ptr(fwd["conic_opacity"]), _ptr(fwd["rgb"]),
                             _ptr(fwd["final_T"]), _ptr(fwd["n_contrib"]), _ptr(dpix), _ptr(dxy), _ptr(dconic), _ptr(dop), _ptr(dfeat), _ptr(fwd["depth"]), _ptr(ddep), _ptr(dz))
    return dxy, dconic, dfeat, dz
b64 = blend(o64, f64); b32 = blend(o32, f32)
mixA = chain(o64, f64, *b32)        # fp32 accumulation of the 2-D gradients, fp64 chain to 3-D
mixB = chain(o32, f32, *b64)        # fp64-accurate 2-D gradients (rounded to fp32), fp32 chain
print("seed", seed, "P", rv["means3D"].shape[0], f"{W}x{H}")
for k in ("means3D", "scales", "rotations"):
    r = g64[k]; n = np.linalg.norm(r)
    print(f"{k:10s} all-fp32 {np.linalg.norm(g32[k] - r) / n:.2e}   fp32 accumulation + fp64 chain {np.linalg.norm(mixA[k] - r) / n:.2e}   fp64 accumulation + fp32 chain {np.linalg.norm(mixB[k] - r) / n:.2e}")
for nm, i in (("dxy", 0), ("dconic", 1)):
    r = b64[i]; print(f"2-D {nm}: fp32 accumulation rel {np.linalg.norm(b32[i] - r) / np.linalg.norm(r):.2e}")
if len(sys.argv) > 3:                                       # one Gaussian in detail
    i = int(sys.argv[3])
    for k in ("means3D", "scales", "rotations"):
        r = g64[k][i]
        print(f"Gaussian {i} {k:10s} fp64 {np.round(r, 5)}  |err| all-fp32 {np.abs(g32[k][i] - r).max():.2e}  fp32 replay + fp64 chain {np.abs(mixA[k][i] - r).max():.2e}  fp64 replay + fp32 chain {np.abs(mixB[k][i] - r).max():.2e}")
    print("   2-D: dconic fp64", b64[1][i], " fp32 replay error", np.abs(b32[1][i] - b64[1][i]), " conic", f64["conic_opacity"][i][:3], "cov2d", f64["cov2d"][i])
