"""Dense, differentiable PyTorch restatement of the rasteriser contract.

TEST INFRASTRUCTURE ONLY (see oracle/gs_oracle.c header; PARITY UNPINNED for the kernel
arithmetic).  O(P x pixels) memory/time: every pixel evaluates every Gaussian in global
(depth, index) order, masked by tile-rect membership -- which is exactly what the tile-structured
algorithm computes, because a pixel's tile list is the (tile, depth, index)-sorted subsequence of
Gaussians whose rect covers that tile.  fp64 autograd of this function is the independent
gradient oracle for gs_oracle.c's hand-derived backward and for the HIP kernels.

Gradient conventions adopted from the public rasteriser family (SURVEY App. A.2 [UP]) and made
explicit here with .detach():
  * alpha = min(0.99, o*G): the clamp passes the gradient through;
  * EWA clamp of tx/tz, ty/tz to +-1.3 tanfov: the clamped coordinate is a constant;
  * skip / stop decisions (power>0, alpha<1/255, T(1-alpha)<1e-4) are piecewise constant;
  * `means2D` is an additive zero in NDC units, so its .grad is the pixel gradient x (0.5W, 0.5H)
    (the quantity the reference's densifier thresholds, slam_external.py:100-108).
"""
from __future__ import annotations

import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def sh_basis(deg: int, d: torch.Tensor) -> torch.Tensor:
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    b = [torch.full_like(x, SH_C0)]
    if deg > 0:
        b += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if deg > 2:
        b += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
              SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy),
              SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=1)


def build_cov3d(scales, rots, mod):
    r, x, y, z = rots[:, 0], rots[:, 1], rots[:, 2], rots[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    M = R * (mod * scales)[:, None, :]
    return M @ M.transpose(1, 2)


def render_dense(cam: dict, means3D, opacities, colors=None, shs=None, scales=None, rotations=None,
                 cov3D_precomp=None, means2D=None, pixel_chunk: int = 4096, tiled: bool = False):
    """cam: dict(W,H,tanfovx,tanfovy,bg[3],scale_modifier,viewmatrix[4,4],projmatrix[4,4],campos[3],sh_degree).
    Matrices exactly as stored in the settings tensors (transposed, row-vector convention).
    tiled=True walks the image tile by tile and evaluates only the Gaussians whose rect covers the tile -- the same arithmetic
    on the same ordered lists (it IS the "CPU PyTorch forward render" of BASELINE configs[0]); tiled=False is the literal dense form.
    Returns dict(color[3,H,W], depth[1,H,W], opacity[1,H,W], radii[P], final_T[H,W], n_contrib[H,W])."""
    dt = means3D.dtype
    W, H = int(cam["W"]), int(cam["H"])
    P = means3D.shape[0]
    V = torch.as_tensor(cam["viewmatrix"], dtype=dt).reshape(4, 4)     # = w2c^T
    Q = torch.as_tensor(cam["projmatrix"], dtype=dt).reshape(4, 4)
    bg = torch.as_tensor(cam["bg"], dtype=dt).reshape(3)
    tfx, tfy = float(cam["tanfovx"]), float(cam["tanfovy"])
    mod = float(cam.get("scale_modifier", 1.0))
    fx, fy = W / (2 * tfx), H / (2 * tfy)
    ones = torch.ones(P, 1, dtype=dt)
    p4 = torch.cat([means3D, ones], 1)
    t = p4 @ V                     # row-vector convention: p_view = p_row * viewmatrix
    hom = p4 @ Q
    tx, ty, tz = t[:, 0], t[:, 1], t[:, 2]
    vis = tz > 0.2
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None]
    if means2D is not None:
        ndc = ndc + means2D[:, :2]
    if cov3D_precomp is not None:
        c = cov3D_precomp
        S3 = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(-1, 3, 3)
    else:
        S3 = build_cov3d(scales, rotations, mod)
    tzs = torch.where(vis, tz, torch.ones_like(tz))
    limx, limy = 1.3 * tfx, 1.3 * tfy
    txtz, tytz = tx / tzs, ty / tzs
    okx = ~((txtz < -limx) | (txtz > limx)); oky = ~((tytz < -limy) | (tytz > limy))
    cx_ = torch.where(okx, tx, (txtz.clamp(-limx, limx) * tzs).detach())
    cy_ = torch.where(oky, ty, (tytz.clamp(-limy, limy) * tzs).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tzs, zero, -(fx * cx_) / (tzs * tzs), zero, fy / tzs, -(fy * cy_) / (tzs * tzs)], 1).reshape(-1, 2, 3)
    Wm = V[:3, :3].T               # W_rc = w2c[r][c]
    T = J @ Wm
    cov = T @ S3 @ T.transpose(1, 2)
    c00, c01, c11 = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = c00 * c11 - c01 * c01
    vis = vis & (det > 0)
    dets = torch.where(vis, det, torch.ones_like(det))
    ca, cb, cc = c11 / dets, -c01 / dets, c00 / dets
    mid = 0.5 * (c00 + c11)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    rf = torch.ceil(3 * torch.sqrt(torch.where(vis, lam, torch.ones_like(lam)))).detach()
    pix = torch.stack([((ndc[:, 0] + 1) * W - 1) * 0.5, ((ndc[:, 1] + 1) * H - 1) * 0.5], 1)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pd = pix.detach()

    def tl(v, hi):
        return v.clamp(-1048576, 1048576).trunc().clamp(0, hi).to(torch.int64)
    x0, x1 = tl((pd[:, 0] - rf) / 16, gx), tl((pd[:, 0] + rf + 15) / 16, gx)
    y0, y1 = tl((pd[:, 1] - rf) / 16, gy), tl((pd[:, 1] + rf + 15) / 16, gy)
    vis = vis & ((x1 - x0) * (y1 - y0) > 0)
    radii = torch.where(vis, rf, torch.zeros_like(rf)).to(torch.int32)
    if shs is not None:
        deg = int(cam.get("sh_degree", 0))
        campos = torch.as_tensor(cam["campos"], dtype=dt).reshape(1, 3)
        d = means3D - campos
        d = d / d.norm(dim=1, keepdim=True)
        B = sh_basis(deg, d)
        rgb = torch.einsum("pk,pkc->pc", B, shs[:, :B.shape[1], :]) + 0.5
        rgb = torch.clamp(rgb, min=0.0)
    else:
        rgb = colors
    op = opacities.reshape(-1)
    # the non-finite rule (gs_oracle.c): a Gaussian whose screen-space record holds a NaN or an infinity is culled
    fin = torch.isfinite(pix.detach()).all(1) & torch.isfinite(ca.detach()) & torch.isfinite(cb.detach()) & torch.isfinite(cc.detach()) \
        & torch.isfinite(op.detach()) & torch.isfinite(rgb.detach()).all(1) & torch.isfinite(tz.detach())
    vis = vis & fin
    radii = torch.where(vis, rf, torch.zeros_like(rf)).to(torch.int32)
    # global (depth as float32 bits, index) order
    key = tz.detach().to(torch.float32)
    key = torch.where(vis, key, torch.full_like(key, float("inf")))
    order = torch.sort(key, stable=True).indices
    order = order[vis[order]]
    xs = torch.arange(W, dtype=dt); ys = torch.arange(H, dtype=dt)
    PX = xs[None, :].expand(H, W).reshape(-1); PY = ys[:, None].expand(H, W).reshape(-1)
    col = torch.zeros(3, H * W, dtype=dt); dep = torch.zeros(H * W, dtype=dt)
    fT = torch.ones(H * W, dtype=dt); ncon = torch.zeros(H * W, dtype=torch.int64)
    o_pix, o_a, o_b, o_c, o_op, o_rgb, o_z = pix[order], ca[order], cb[order], cc[order], op[order], rgb[order], tz[order]
    ox0, ox1, oy0, oy1 = x0[order], x1[order], y0[order], y1[order]
    cols, deps, fTs, ncs = [], [], [], []
    if tiled:
        chunks = []
        for ty_ in range(gy):
            for tx_ in range(gx):
                yy, xx = torch.meshgrid(torch.arange(ty_ * 16, min(H, ty_ * 16 + 16)), torch.arange(tx_ * 16, min(W, tx_ * 16 + 16)), indexing="ij")
                sel = torch.nonzero((ox0 <= tx_) & (tx_ < ox1) & (oy0 <= ty_) & (ty_ < oy1)).reshape(-1)
                chunks.append(((yy * W + xx).reshape(-1), sel))
    else:
        chunks = [(torch.arange(s, min(s + pixel_chunk, H * W)), None) for s in range(0, H * W, pixel_chunk)]
    pix_index = []
    for pidx, sel in chunks:
        s, e = 0, int(pidx.numel())
        pix_index.append(pidx)
        px, py = PX[pidx, None], PY[pidx, None]
        tix, tiy = (px / 16).floor().to(torch.int64), (py / 16).floor().to(torch.int64)
        if sel is not None:            # the tile's ordered list
            o_pix, o_a, o_b, o_c, o_op, o_rgb, o_z = (v[order][sel] for v in (pix, ca, cb, cc, op, rgb, tz))
            ox0_, ox1_, oy0_, oy1_ = ox0[sel], ox1[sel], oy0[sel], oy1[sel]
        else:
            ox0_, ox1_, oy0_, oy1_ = ox0, ox1, oy0, oy1
        member = (tix >= ox0_[None]) & (tix < ox1_[None]) & (tiy >= oy0_[None]) & (tiy < oy1_[None])
        dx, dy = o_pix[None, :, 0] - px, o_pix[None, :, 1] - py
        power = -0.5 * (o_a[None] * dx * dx + o_c[None] * dy * dy) - o_b[None] * dx * dy
        G = torch.exp(torch.clamp(power, max=0.0))
        a_raw = o_op[None] * G
        alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()
        valid = member & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
        am = torch.where(valid, alpha, torch.zeros_like(alpha))
        Tin = torch.cumprod(1 - am, dim=1)
        keep = valid & (Tin.detach() >= 1e-4)
        am = torch.where(keep, alpha, torch.zeros_like(alpha))
        Tin = torch.cumprod(1 - am, dim=1)
        Tex = torch.cat([torch.ones_like(Tin[:, :1]), Tin[:, :-1]], 1)
        w = am * Tex
        Tf = Tin[:, -1] if Tin.shape[1] > 0 else torch.ones(e - s, dtype=dt)
        cols.append((w @ o_rgb).T + Tf[None] * bg[:, None])
        deps.append(w @ o_z)
        fTs.append(Tf)
        pos = torch.cumsum(member.to(torch.int64), 1)
        ncs.append((pos * keep).max(dim=1).values if keep.shape[1] > 0 else torch.zeros(e - s, dtype=torch.int64))
    col = torch.cat(cols, 1); dep = torch.cat(deps); fT = torch.cat(fTs); ncon = torch.cat(ncs)
    if tiled:                         # back to row-major pixel order
        inv = torch.empty(H * W, dtype=torch.int64)
        allp = torch.cat(pix_index)
        inv[allp] = torch.arange(H * W)
        col, dep, fT, ncon = col[:, inv], dep[inv], fT[inv], ncon[inv]
    return dict(color=col.reshape(3, H, W), depth=dep.reshape(1, H, W), opacity=(1 - fT).reshape(1, H, W),
                radii=radii, final_T=fT.reshape(H, W).detach(), n_contrib=ncon.reshape(H, W))
