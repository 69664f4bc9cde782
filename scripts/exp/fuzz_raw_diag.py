"""Development helper: attribute the image differences scripts/exp/fuzz_raw.py flags between the raw-parameter entry and the activation-kernels path to the two
discrete decisions a one-ulp difference of the transformed centre can move: (a) a radius that lands on the other side of an integer (ceil(3 sigma): the tile
rectangle gains / loses a row or column of tiles where alpha is still up to 0.011 x opacity), (b) two Gaussians whose view depths are within an ulp and swap in the
depth order.  Prints, per seed, the bounding box of the differing pixels, the Gaussians whose radii differ and the adjacent depth pairs whose order differs
between an fp32 and an fp64 evaluation of the frame transform, each with its footprint.   SEEDS=1782,3452 python scripts/exp/fuzz_raw_diag.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import mapping as M, rasterizer as R
from activesplat_amd import synthetic as syn
from activesplat_amd.camera import setup_camera

dev = "cuda"
for seed in [int(s) for s in os.environ["SEEDS"].split(",")]:
    r = np.random.RandomState(seed)
    n = int(r.choice([int(r.randint(50, 2000)), int(r.randint(2000, 40000))]))
    W, H = (int(r.randint(24, 200)), int(r.randint(24, 160))) if seed % 3 else (int(r.randint(272, 420)), int(r.randint(256, 330)))
    sh, iso = seed % 4 == 0, bool(seed % 2)
    p = syn.make_params(n, W, H, seed=seed, sh_degree=3 if sh else None)
    if sh:
        p.pop("rgb_colors", None)
    if iso:
        p["log_scales"] = p["log_scales"][:, :1].contiguous()
    p["log_scales"] = p["log_scales"] + float(r.uniform(-0.5, 1.2))
    cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev, sh_degree=3 if sh else 0)
    a = float(r.uniform(-0.4, 0.4))
    pose = [float(np.cos(a / 2)), 0.0, float(np.sin(a / 2)), 0.0, float(r.uniform(-0.2, 0.2)), float(r.uniform(-0.1, 0.1)), float(r.uniform(-0.8, 0.4))]
    out = []
    with torch.no_grad():
        for raw in (False, True):
            prm = {k: v.clone().to(dev) for k, v in p.items()}
            col = dict(shs=prm["shs"]) if sh else dict(colors_precomp=prm["rgb_colors"])
            if raw:
                m2d = torch.empty_like(prm["means3D"])
                im, radius, depth, sil, dsq = R.render_rgbd_raw(cam, prm["means3D"], m2d, prm["logit_opacities"], prm["log_scales"], prm["unnorm_rotations"], pose, **col)
            else:
                rv = M.fused_rendervar(dict(prm, rgb_colors=prm["shs"]) if sh else prm, 0, pose)
                rv.pop("colors_precomp")
                m2d = rv["means2D"]
                z32 = rv["means3D"][:, 2].clone()
                xy = rv["means3D"][:, :2] / rv["means3D"][:, 2:3]
                im, radius, depth, sil, dsq = R.render_rgbd(cam, **col, **rv)
            out.append((im.clone(), radius.clone()))
    d = (out[0][0] - out[1][0]).abs().amax(0)
    ys, xs = torch.nonzero(d > 5e-5, as_tuple=True)
    print(f"seed {seed} n {n} {W}x{H}: {len(ys)} pixels differ by more than 5e-5 (max {float(d.max()):.2e}); box x {int(xs.min())}..{int(xs.max())} y {int(ys.min())}..{int(ys.max())}; "
          f"{int((d > 1e-3).sum())} above 1e-3")
    K = syn.intrinsics(W, H)
    px = (xy[:, 0] * K[0][0] + K[0][2]).cpu().numpy(); py = (xy[:, 1] * K[1][1] + K[1][2]).cpu().numpy()
    ra, rb = out[0][1].cpu().numpy(), out[1][1].cpu().numpy()
    for i in np.nonzero(ra != rb)[0][:6]:
        print(f"   radius differs: Gaussian {i}: {ra[i]} vs {rb[i]} at ({px[i]:.1f},{py[i]:.1f}), opacity {float(torch.sigmoid(p['logit_opacities'][i])):.3f}; "
              f"tile columns {int((px[i]-ra[i])//16)}..{int((px[i]+ra[i])//16)} vs {int((px[i]-rb[i])//16)}..{int((px[i]+rb[i])//16)}, rows {int((py[i]-ra[i])//16)}..{int((py[i]+ra[i])//16)} vs {int((py[i]-rb[i])//16)}..{int((py[i]+rb[i])//16)}")
    # the frame transform in fp64 from the same fp32 parameters, rounded once: the order an exact evaluation gives
    q = torch.tensor(pose[:4], dtype=torch.float64); t = torch.tensor(pose[4:], dtype=torch.float64)
    w, x, y, z = q / q.norm()
    Rm = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], dtype=torch.float64)
    z64 = (p["means3D"].double() @ Rm.T + t)[:, 2]
    z32c = z32.cpu()
    order = torch.argsort(z32c, stable=True)
    zz = z64[order]
    zs = z32c[order]
    inv = torch.nonzero((zz[1:] < zz[:-1]) | ((zs[1:] - zs[:-1]) <= 4 * torch.finfo(torch.float32).eps * zs[1:]))[:, 0]     # inverted in fp64, or within four ulps in fp32
    shown = 0
    for j in inv.tolist():
        i0, i1 = int(order[j]), int(order[j + 1])
        inbox = all(px[i] + ra[i] >= int(xs.min()) and px[i] - ra[i] <= int(xs.max()) and py[i] + ra[i] >= int(ys.min()) and py[i] - ra[i] <= int(ys.max()) for i in (i0, i1))
        if ra[i0] > 0 and ra[i1] > 0 and inbox and shown < 6:
            shown += 1
            print(f"   depth order not decided in fp32, both footprints reach the box: Gaussians {i0} / {i1}: z {float(z32c[i0]):.9g} / {float(z32c[i1]):.9g} (fp64 {float(z64[i0]):.12g} / {float(z64[i1]):.12g}); "
                  f"centres ({px[i0]:.1f},{py[i0]:.1f}) r {ra[i0]} / ({px[i1]:.1f},{py[i1]:.1f}) r {ra[i1]}; opacities {float(torch.sigmoid(p['logit_opacities'][i0])):.3f} / {float(torch.sigmoid(p['logit_opacities'][i1])):.3f}")
    print(f"   {len(inv)} adjacent pairs in the fp32 depth order are within four ulps in fp32 or inverted in fp64; {int((ra != rb).sum())} radii differ")
    # the pixels no undecided pair covers (tests/parity_cases._depth_ties_cover's criterion), with their differences
    rad = out[0][1].cpu().float()
    cov = torch.zeros(len(ys), dtype=torch.bool)
    xs_c, ys_c = xs.cpu(), ys.cpu()
    pxt, pyt = torch.from_numpy(px), torch.from_numpy(py)
    for j in inv.tolist():
        i0, i1 = int(order[j]), int(order[j + 1])
        if rad[i0] > 0 and rad[i1] > 0:
            both = torch.ones(len(ys), dtype=torch.bool)
            for i in (i0, i1):
                both &= ((xs_c - pxt[i]).abs() <= rad[i] + 1) & ((ys_c - pyt[i]).abs() <= rad[i] + 1)
            cov |= both
    unc = torch.nonzero(~cov)[:, 0].tolist()
    print(f"   {len(unc)} differing pixels outside every undecided pair's footprints:", [(int(xs_c[k]), int(ys_c[k]), "%.2e" % float(d[ys[k], xs[k]])) for k in unc[:12]])
