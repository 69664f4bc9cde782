"""Shared helpers of the parity tests: run the product path (HIP on the GPU, or the host-emulated
kernels on CPU) and the oracle on the same inputs, and decode the integer artefacts from the state
buffers through the layouts published in include/gsplat_hip.h."""
import numpy as np
import torch

from activesplat_amd import rasterizer as R
from activesplat_amd import synthetic as syn
from activesplat_amd.camera import setup_camera


def cam_dict(rs):
    return dict(W=int(rs.image_width), H=int(rs.image_height), tanfovx=float(rs.tanfovx), tanfovy=float(rs.tanfovy),
                bg=rs.bg.detach().cpu().numpy(), scale_modifier=float(rs.scale_modifier),
                viewmatrix=rs.viewmatrix.detach().cpu().contiguous().numpy().reshape(4, 4),
                projmatrix=rs.projmatrix.detach().cpu().contiguous().numpy().reshape(4, 4),
                campos=rs.campos.detach().cpu().numpy(), sh_degree=int(rs.sh_degree))


def pose(yaw=0.0, t=(0.0, 0.0, 0.0)):
    c, s = np.cos(yaw), np.sin(yaw)
    w2c = np.eye(4)
    w2c[:3, :3] = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
    w2c[:3, 3] = t
    return w2c


def scene(N, W, H, seed=0, device="cpu", w2c=None, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, sh_degree=None, K=None, **kw):
    K = syn.intrinsics(W, H) if K is None else K
    rs = setup_camera(W, H, K, np.eye(4) if w2c is None else w2c, device=device, bg=bg, scale_modifier=scale_modifier,
                      sh_degree=0 if sh_degree is None else sh_degree)
    rs = rs._replace(debug=True)
    p = syn.make_params(N, W, H, seed=seed, sh_degree=sh_degree, **kw)
    rv = {k: v.to(device) for k, v in syn.activate(p).items()}
    return rs, rv


def use_emulated_kernels(path):
    """TEST-ONLY: bind activesplat_amd to the host-emulated build of the kernel sources (tests/hipemu) and let the rasteriser take host
    tensors for it (the product path refuses them); returns the undo function."""
    from activesplat_amd import _lib
    _lib.load_for_tests(path)
    require = R._require_rocm
    R._require_rocm = lambda device: None

    def undo():
        R._require_rocm = require
        _lib.unload_for_tests()
    return undo


#: state buffers of the most recent run_product() forward (rasterizer.capture()): what artefacts() decodes
LAST = {}


def run_product(rs, rv, dL=None):
    """-> dict(color, radii, depth, opacity[, grads])"""
    inp = {k: v.detach().clone().requires_grad_(dL is not None) for k, v in rv.items()}
    P = inp["means3D"].shape[0]
    m2d = torch.zeros(P, 3, device=inp["means3D"].device, requires_grad=dL is not None)
    with R.capture() as state:
        color, radii, depth, opacity = R.GaussianRasterizer(raster_settings=rs)(means2D=m2d, **inp)
    LAST.clear(); LAST.update(state)
    out = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy(), depth=depth.cpu().numpy(),
               opacity=opacity.cpu().numpy(), D=R.last_stats["num_rendered"])
    if dL is not None:
        color.backward(dL.to(color.device))
        out["grads"] = {k: v.grad.detach().cpu().numpy() for k, v in inp.items()}
        out["grads"]["means2D"] = m2d.grad.detach().cpu().numpy()
    return out


def run_oracle(oracle, rs, rv, dL=None):
    n = lambda k: rv[k].detach().cpu().numpy() if k in rv else None  # noqa: E731
    f = oracle.forward(cam_dict(rs), n("means3D"), n("opacities"), colors=n("colors_precomp"), shs=n("shs"),
                       scales=n("scales"), rotations=n("rotations"), cov3D_precomp=n("cov3D_precomp"))
    if dL is not None:
        f["grads"] = oracle.backward(f, dL.detach().cpu().numpy())
    return f


def artefacts():
    """Integer artefacts of the most recent debug forward, decoded via the published layouts."""
    d = LAST
    P, D, W, H = d["P"], d["D"], d["W"], d["H"]
    gl, il, bl = d["gl"], d["il"], d["bl"]
    geom = d["geom"].cpu().numpy(); image = d["image"].cpu().numpy(); binning = d["binning"].cpu().numpy()

    def view(buf, off, dtype, n):
        return np.frombuffer(buf.tobytes()[off:off + n * np.dtype(dtype).itemsize], dtype=dtype).copy()
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    rect_packed = view(geom, gl.rect, np.uint32, 2 * P).reshape(P, 2)
    rect = np.stack([rect_packed[:, 0] & 0xFFFF, rect_packed[:, 1] & 0xFFFF, rect_packed[:, 0] >> 16,
                     rect_packed[:, 1] >> 16], 1).astype(np.int32)
    ranges = view(image, il.ranges, np.uint32, 2 * tiles).reshape(tiles, 2)
    out = dict(geom=view(geom, gl.geom, np.float32, 12 * P).reshape(P, 12), rect=rect, path=int(bl.path),
               tiles_touched=view(geom, gl.tiles_touched, np.uint32, P),
               point_list=d["point_list"].cpu().numpy()[:D].astype(np.uint32), ranges=ranges,
               final_T=view(image, il.final_T, np.float32, W * H).reshape(H, W),
               n_contrib=view(image, il.n_contrib, np.uint32, W * H).reshape(H, W))
    if bl.path == 1:      # GS_SORT_TILE_LDS: the sorted list is point_list (ids, tile-major); the 64-bit keys (tile << 32 | depth bits) are
        # rebuilt from it and the per-Gaussian depth bits (the bucket sort does not write sorted pairs back)
        depth_bits = view(geom, gl.depth_bits, np.uint32, P).astype(np.uint64)
        tile_of = np.repeat(np.arange(tiles, dtype=np.uint64), (ranges[:, 1] - ranges[:, 0]).astype(np.int64))
        out["keys_sorted"] = (tile_of << np.uint64(32)) | depth_bits[out["point_list"].astype(np.int64)]
        out["pairs_ids"] = out["point_list"]
        out["offsets"] = None
    else:
        out["keys_sorted"] = view(binning, bl.keys_sorted, np.uint64, D)
        out["offsets"] = view(geom, gl.offsets, np.uint32, P)
    return out


def close_frac(a, b, rtol, atol):
    """fraction of elements of a within atol + rtol*|b| of b"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.mean(np.abs(a - b) <= atol + rtol * np.abs(b)))


def psnr(a, b):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 120.0 if mse == 0 else 10 * np.log10(1.0 / mse)


from activesplat_amd.workloads import configs2_optimise_loop  # noqa: E402,F401  (moved into the package: bench.py times it too)


# ---- TEST-ONLY: a GaussianRasterizer whose forward and backward are the C oracle (oracle/gs_oracle.c) -----------------------
class _OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, oracle, rs, means3D, means2D, opacities, colors, scales, rotations):
        n = lambda t: t.detach().cpu().numpy()  # noqa: E731
        f = oracle.forward(cam_dict(rs), n(means3D), n(opacities), colors=n(colors), scales=n(scales), rotations=n(rotations))
        ctx.oracle, ctx.f = oracle, f
        H, W = int(rs.image_height), int(rs.image_width)
        dev = means3D.device
        t = lambda a, *s: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).reshape(*s).to(dev)  # noqa: E731
        color, depth, opacity = t(f["color"], 3, H, W), t(f["out_depth"], 1, H, W), t(f["opacity"], 1, H, W)
        radii = torch.from_numpy(np.asarray(f["radii"], dtype=np.int32).copy()).to(dev)
        ctx.mark_non_differentiable(radii, depth, opacity)
        return color, radii, depth, opacity

    @staticmethod
    def backward(ctx, g_color, _gr, _gd, _go):
        g = ctx.oracle.backward(ctx.f, g_color.detach().cpu().numpy())
        t = lambda k: torch.from_numpy(np.ascontiguousarray(np.asarray(g[k], dtype=np.float32))).to(g_color.device)  # noqa: E731
        return (None, None, t("means3D").reshape(-1, 3), t("means2D").reshape(-1, 3), t("opacities").reshape(-1, 1),
                t("colors_precomp").reshape(-1, 3), t("scales").reshape(-1, 3), t("rotations").reshape(-1, 4))


def oracle_rasterizer_class(oracle):
    """-> a drop-in for activesplat_amd.GaussianRasterizer (colors_precomp + scales/rotations call sites) backed by `oracle`."""
    class OracleRasterizer(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, means3D, means2D, opacities, colors_precomp=None, scales=None, rotations=None, shs=None, cov3D_precomp=None):
            assert shs is None and cov3D_precomp is None
            return _OracleRasterize.apply(oracle, self.raster_settings, means3D, means2D, opacities, colors_precomp, scales, rotations)
    return OracleRasterizer
