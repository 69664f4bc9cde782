"""Development helper: random sweep of the RAW-PARAMETER rasteriser entry (render_rgbd_raw: frame transform + activations inside the per-Gaussian kernels) against the
activation kernels in front of the plain fused render (fused_rendervar + render_rgbd) -- two product paths that must agree: radii within one on a handful of boundary
Gaussians, images up to bounded threshold flips, gradients w.r.t. the PARAMETERS to 3e-4 for all rows but the two worst (1e-3 with them).  Colours / 16-coefficient SH rows,
isotropic / anisotropic maps, random poses, image sizes on both sides of the few-tile and chained-kernel thresholds.
GPU box: SEED0=0 SEED1=300 python scripts/exp/fuzz_raw.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import mapping as M, rasterizer as R
from activesplat_amd import synthetic as syn
from activesplat_amd.camera import setup_camera

dev = os.environ.get("DEVICE", "cuda")
if dev == "cpu":                                            # the host-emulated kernels (tests/hipemu): a flagged seed traced without a GPU
    import subprocess
    from tests import util
    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "hipemu"), "-j8"], stdout=subprocess.DEVNULL)
    util.use_emulated_kernels(os.path.join(ROOT, "tests", "hipemu", "libgsplat_emu.so"))
    R.use_frontend = False
bad, ties, rows = [], [], []


def depth_ties_cover(prm_means, pose, rv_means, radius, d, W, H, tol):
    """True when every pixel of `d` above `tol` lies in the footprints of a pair of Gaussians whose view depths are within four ulps in fp32 (or ordered the other
    way in fp64): the depth order of such a pair is not decided in fp32, the two entries evaluate the frame transform with different roundings, and either order is a
    valid rendering (scripts/exp/fuzz_raw_diag.py prints the pairs)."""
    K = syn.intrinsics(W, H)
    z32 = rv_means[:, 2].cpu()
    px = (rv_means[:, 0] / rv_means[:, 2] * K[0][0] + K[0][2]).cpu(); py = (rv_means[:, 1] / rv_means[:, 2] * K[1][1] + K[1][2]).cpu()
    q = torch.tensor(pose[:4], dtype=torch.float64); t = torch.tensor(pose[4:], dtype=torch.float64)
    w, x, y, z = q / q.norm()
    Rm = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], dtype=torch.float64)
    z64 = (prm_means.double().cpu() @ Rm.T + t)[:, 2]
    order = torch.argsort(z32, stable=True)
    zs, zz = z32[order], z64[order]
    near = torch.nonzero((zz[1:] < zz[:-1]) | ((zs[1:] - zs[:-1]) <= 4 * torch.finfo(torch.float32).eps * zs[1:].abs()))[:, 0]
    ra = radius.cpu().float()
    ys, xs = torch.nonzero(d.cpu() > tol, as_tuple=True)
    covered = torch.zeros(len(ys), dtype=torch.bool)
    for j in near.tolist():
        i0, i1 = int(order[j]), int(order[j + 1])
        if ra[i0] > 0 and ra[i1] > 0:
            both = torch.ones(len(ys), dtype=torch.bool)
            for i in (i0, i1):
                both &= ((xs - px[i]).abs() <= ra[i] + 1) & ((ys - py[i]).abs() <= ra[i] + 1)
            covered |= both
    return bool(covered.all()), int(covered.sum()), len(ys)

for seed in range(int(os.environ.get("SEED0", 0)), int(os.environ.get("SEED1", 300))):
    r = np.random.RandomState(seed)
    n = int(r.choice([int(r.randint(50, 2000)), int(r.randint(2000, 40000))]))
    W, H = (int(r.randint(24, 200)), int(r.randint(24, 160))) if seed % 3 else (int(r.randint(272, 420)), int(r.randint(256, 330)))
    sh, iso = seed % 4 == 0, bool(seed % 2)
    try:
        p = syn.make_params(n, W, H, seed=seed, sh_degree=3 if sh else None)
        if sh:
            p.pop("rgb_colors", None)
        if iso:
            p["log_scales"] = p["log_scales"][:, :1].contiguous()
        p["log_scales"] = p["log_scales"] + float(r.uniform(-0.5, 1.2))
        cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev, sh_degree=3 if sh else 0)
        a = float(r.uniform(-0.4, 0.4))
        pose = [float(np.cos(a / 2)), 0.0, float(np.sin(a / 2)), 0.0, float(r.uniform(-0.2, 0.2)), float(r.uniform(-0.1, 0.1)), float(r.uniform(-0.8, 0.4))]
        g = torch.Generator().manual_seed(seed)
        dLc, dLd = torch.randn(3, H, W, generator=g).to(dev), torch.randn(1, H, W, generator=g).to(dev)
        out = []
        for raw in (False, True):
            prm = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in p.items()}
            col = dict(shs=prm["shs"]) if sh else dict(colors_precomp=prm["rgb_colors"])
            if raw:
                m2d = torch.empty_like(prm["means3D"], requires_grad=True)
                im, radius, depth, sil, dsq = R.render_rgbd_raw(cam, prm["means3D"], m2d, prm["logit_opacities"], prm["log_scales"], prm["unnorm_rotations"], pose, **col)
            else:
                rv = M.fused_rendervar(dict(prm, rgb_colors=prm["shs"]) if sh else prm, 0, pose)
                rv.pop("colors_precomp")
                m2d = rv["means2D"]
                tm = rv["means3D"].detach().clone()
                im, radius, depth, sil, dsq = R.render_rgbd(cam, **col, **rv)
            ((im * dLc).sum() + (depth * dLd).sum()).backward()
            out.append((im.detach(), depth.detach(), radius.clone(), {k: v.grad.double() for k, v in prm.items() if v.grad is not None}, m2d.grad.double()))
        x, y = out
        dr = (x[2] - y[2]).abs()
        assert int(dr.max()) <= 1 and float((dr > 0).float().mean()) < 2e-3 + 2.0 / n, ("radii", int(dr.max()), int((dr > 0).sum()))
        tie = False
        for u, v, tol, nm in ((x[0], y[0], 5e-5, "colour"), (x[1], y[1], 5e-4, "depth")):
            d = (u - v).abs()
            if not (float((d > tol).float().mean()) < 2e-3 and float(d.max()) < 0.03 * max(1.0, float(v.abs().max()))):
                ok, c, m = depth_ties_cover(p["means3D"], pose, tm, x[2], d.amax(0), W, H, tol)
                assert ok, (nm, float(d.max()), int((d > tol).sum()), "pixels in depth-tie footprints: %d of %d" % (c, m))
                tie = True
        if tie:                                             # the two renders order a depth tie differently: their gradients are those of two different (valid) orders
            ties.append(seed)
            continue
        for k in x[3]:
            if iso and k == "unnorm_rotations":
                continue
            e2 = ((x[3][k] - y[3][k]).reshape(n, -1) ** 2).sum(1)
            nr = float(x[3][k].norm().clamp_min(1e-30))
            rest = float((e2.sum() - e2.sort().values[-2:].sum()).clamp_min(0).sqrt()) if n > 8 else 0.0
            assert rest < 3e-4 * nr, (k, rest / nr, float(e2.sum().sqrt()) / nr)
            if not float(e2.sum().sqrt()) < 3e-3 * nr:      # everything but two rows agrees to 3e-4: report the rows (the conditioning class of DESIGN section 6), do not hide them
                w = int(e2.argmax())
                rows.append((seed, k, w, round(float(e2.sum().sqrt()) / nr, 5), [round(float(v), 3) for v in p["log_scales"][w].exp()]))
                if os.environ.get("ROWS"):
                    K = syn.intrinsics(W, H)
                    cx, cy = float(tm[w, 0] / tm[w, 2] * K[0][0] + K[0][2]), float(tm[w, 1] / tm[w, 2] * K[1][1] + K[1][2])
                    dd = (x[0] - y[0]).abs().amax(0)
                    rr = int(x[2][w]) + 1
                    x0, x1, y0, y1 = max(0, int(cx) - rr), min(W, int(cx) + rr + 2), max(0, int(cy) - rr), min(H, int(cy) + rr + 2)
                    box = dd[y0:y1, x0:x1]
                    top = torch.topk(box.flatten(), min(5, box.numel()))
                    print("   centre (%.1f, %.1f); largest colour differences inside its footprint:" % (cx, cy), [(x0 + int(i) % (x1 - x0), y0 + int(i) // (x1 - x0), "%.2e" % float(v)) for v, i in zip(top.values, top.indices)],
                          "; median difference over the image %.2e, maximum %.2e" % (float(dd.median()), float(dd.max())))
                    print("   seed", seed, k, "row", w, "radius", int(x[2][w]), "norm of all rows %.4g" % nr, "\n      activation kernels:", x[3][k][w].tolist(), "\n      raw entry:         ", y[3][k][w].tolist(),
                          "\n      means3D rows:", x[3]["means3D"][w].tolist(), y[3]["means3D"][w].tolist(), "\n      log_scales rows:", x[3]["log_scales"][w].tolist(), y[3]["log_scales"][w].tolist())
    except Exception as e:
        bad.append(seed)
        print("FAIL seed", seed, "n", n, f"{W}x{H}", "sh" if sh else "rgb", "iso" if iso else "aniso", repr(e)[:200], flush=True)
print("raw-parameter sweep: seeds %s..%s, %d failures %s" % (os.environ.get("SEED0", 0), os.environ.get("SEED1", 300), len(bad), bad[:20]))
print("   %d scenes where the two entries order a depth tie (views depths within four fp32 ulps) differently and every differing pixel lies in the pair's footprints: %s" % (len(ties), ties))
print("   %d scenes where two rows carry a gradient difference above 3e-3 of the norm while the rest agrees to 3e-4 (seed, key, row, relative difference, scales of the row): %s" % (len(rows), rows))
