"""GPU (-m gpu): fused Adam, row compaction / surgery and the mapper harness on the real HIP library."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(n, dev, seed=0, iso=False):
    g = torch.Generator().manual_seed(seed)
    d = dict(means3D=torch.randn(n, 3, generator=g), rgb_colors=torch.rand(n, 3, generator=g), unnorm_rotations=torch.randn(n, 4, generator=g),
             logit_opacities=torch.randn(n, 1, generator=g) * 3, log_scales=torch.randn(n, 1 if iso else 3, generator=g) - 2.0,
             cam_unnorm_rots=torch.randn(1, 4, 2, generator=g), cam_trans=torch.randn(1, 3, 2, generator=g))
    return {k: torch.nn.Parameter(v.to(dev)) for k, v in d.items()}


LRS = dict(means3D=1e-4, rgb_colors=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05, log_scales=1e-3, cam_unnorm_rots=0.0, cam_trans=0.0)


def test_gaussian_adam_equals_torch_adam_on_gpu(hip):
    from activesplat_amd import optim as O
    n = 300_007
    pa, pb = _params(n, hip), _params(n, hip)
    oa = O.initialize_optimizer(pa, LRS)
    ob = torch.optim.Adam([{"params": [v], "name": k, "lr": LRS[k]} for k, v in pb.items()], lr=0.0, eps=1e-15)
    g = torch.Generator(device=hip).manual_seed(1)
    for step in range(4):
        for k in pa:
            if k.startswith("cam_"):
                pa[k].grad = pb[k].grad = None
            else:
                gr = torch.randn(pa[k].shape, generator=g, device=hip) * 10.0 ** (step - 2)
                pa[k].grad, pb[k].grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
        for k in pa:
            np.testing.assert_allclose(pa[k].detach().cpu().numpy(), pb[k].detach().cpu().numpy(), rtol=3e-6, atol=1e-7, err_msg=k)
    for k in ("means3D", "log_scales"):
        np.testing.assert_allclose(oa.state[pa[k]]["exp_avg_sq"].cpu().numpy(), ob.state[pb[k]]["exp_avg_sq"].cpu().numpy(), rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("iso", [False, True])
def test_remove_points_equals_boolean_indexing(hip, iso):
    from activesplat_amd import optim as O
    n = 123_457
    params = _params(n, hip, seed=2, iso=iso)
    opt = O.initialize_optimizer(params, LRS)
    for k, p in params.items():
        p.grad = None if k.startswith("cam_") else torch.randn_like(p)
    opt.step()
    variables = dict(means2D_gradient_accum=torch.rand(n, device=hip), denom=torch.rand(n, device=hip), max_2D_radius=torch.rand(n, device=hip),
                     timestep=torch.arange(n, device=hip).float())
    to_remove = torch.rand(n, device=hip) < 0.37
    keep = ~to_remove
    want = {k: v.detach()[keep].clone() for k, v in params.items() if not k.startswith("cam_")}
    want_m = {k: opt.state[v]["exp_avg"][keep].clone() for k, v in params.items() if not k.startswith("cam_")}
    want_v = {k: v[keep].clone() for k, v in variables.items()}
    params, variables = O.remove_points(to_remove, params, variables, opt)
    for k in want:
        assert torch.equal(params[k].detach(), want[k]) and torch.equal(opt.state[params[k]]["exp_avg"], want_m[k])
        assert isinstance(params[k], torch.nn.Parameter) and params[k].requires_grad
    for k in want_v:
        assert torch.equal(variables[k], want_v[k])
    # edge cases: remove nothing / everything
    none = torch.zeros(params["means3D"].shape[0], dtype=torch.bool, device=hip)
    n1 = params["means3D"].shape[0]
    params, variables = O.remove_points(none, params, variables, opt)
    assert params["means3D"].shape[0] == n1
    params, variables = O.remove_points(~none, params, variables, opt)
    assert params["means3D"].shape[0] == 0 and variables["denom"].shape[0] == 0


def test_densify_and_prune_on_gpu(hip):
    from activesplat_amd import optim as O
    n = 50_000
    params = _params(n, hip, seed=3)
    with torch.no_grad():
        params["log_scales"] -= 2.0
    opt = O.initialize_optimizer(params, LRS)
    for k, p in params.items():
        p.grad = None if k.startswith("cam_") else torch.randn_like(p)
    opt.step()
    m2d = torch.zeros(n, 3, device=hip, requires_grad=True)
    m2d.grad = torch.randn(n, 3, device=hip) * 3e-4
    variables = dict(means2D=m2d, seen=torch.rand(n, device=hip) > 0.3, means2D_gradient_accum=torch.rand(n, device=hip) * 4e-4,
                     denom=(torch.rand(n, device=hip) * 3).floor(), max_2D_radius=torch.zeros(n, device=hip),
                     timestep=torch.arange(n, device=hip).float(), scene_radius=torch.tensor(2.0, device=hip))
    ddict = dict(start_after=0, remove_big_after=0, stop_after=100, densify_every=10, grad_thresh=0.0002, num_to_split_into=2,
                 removal_opacity_threshold=0.005, final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=3000)
    params, variables = O.densify(params, variables, opt, 10, ddict)
    m = params["means3D"].shape[0]
    assert m != n and all(params[k].shape[0] == m for k in ("rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales"))
    assert all(variables[k].shape[0] == m for k in ("means2D_gradient_accum", "denom", "max_2D_radius", "timestep"))
    assert (torch.sigmoid(params["logit_opacities"]) >= 0.005).all() and torch.isfinite(params["means3D"]).all()
    pdict = dict(start_after=0, remove_big_after=0, stop_after=20, prune_every=20, removal_opacity_threshold=0.3,
                 final_removal_opacity_threshold=0.3, reset_opacities=True, reset_opacities_every=20)
    params, variables = O.prune_gaussians(params, variables, opt, 20, pdict)
    assert params["means3D"].shape[0] < m
    np.testing.assert_allclose(torch.sigmoid(params["logit_opacities"]).detach().cpu().numpy(), 0.01, rtol=1e-5)    # reset after the prune


def test_mapper_harness_gpu(hip):
    from tests import util
    from tests.test_mapper import run_harness
    mp, seq, log = run_harness(hip, n_gt=150_000, W=256, H=256, frames=11)
    assert [e["iters"] for e in log] == [2, 0, 0, 0, 0, 2, 0, 0, 0, 0, 2]
    assert log[4]["grew"] > 0 and log[9]["grew"] > 0 and len(mp.keyframe_list) == 3
    for fr in (seq[0], seq[9]):
        im, depth, opacity = mp.render_rgbd(fr["w2c"])
        seen = (fr["depth"] > 0)[0]
        ps = util.psnr(im[:, seen].cpu().numpy(), fr["color"][:, seen].cpu().numpy())
        print(f"mapper harness re-render PSNR frame {fr['id']}: {ps:.2f} dB")
        assert ps > 20.0, ps                        # (six mapping iterations in all: 22.1 dB at frame 0)
        err = ((depth / opacity.clamp_min(1e-6))[0][seen] - fr["depth"][0][seen]).abs().median()
        assert float(err) < 0.1


def test_mapper_harness_report_configuration_psnr(hip):
    """The configs[4] substitute as profiles/ reports it (scripts/configs_report.py): 31-frame spin inside a 400 k-Gaussian scene at 256 x 256,
    the high-resolution setting's two iterations per frame, fused paths incl. Adam inside the backward: mean re-render PSNR against the
    synthetic ground truth (reported: 26.1 dB) and SSIM."""
    from activesplat_amd import mapping as M
    from tests import util
    from tests.test_mapper import run_harness
    flags = dict(fused_render=True, fused_loss=True, fused_inputs=True, fused_preprocess=True, fused_adam=True, fused_iteration=True, fused_growth=True, fused_keyframes=True)
    mp, seq, log = run_harness(hip, n_gt=400_000, W=256, H=256, frames=31, cfg=dict(mapping_iters=10, **flags))
    assert all(e["iters"] == 2 for e in log)
    ps, ss = [], []
    for fr in seq[::5]:
        im, depth, opacity = mp.render_rgbd(fr["w2c"])
        seen = (fr["depth"] > 0)[0]
        ps.append(util.psnr(im[:, seen].cpu().numpy(), fr["color"][:, seen].cpu().numpy()))
        ss.append(float(M.calc_ssim(im.clamp(0, 1)[None], fr["color"][None].to(im.device))))
    print(f"mapper harness (report configuration): PSNR {np.mean(ps):.2f} dB, SSIM {np.mean(ss):.4f}, {mp.params['means3D'].shape[0]} Gaussians")
    assert np.mean(ps) > 24.0, ps
    assert np.mean(ss) > 0.6, ss


def test_fused_loss_and_inputs_on_gpu(hip):
    """gs_mapping_loss / gs_activate_* on the GPU == the torch mirrors of the reference functions, and the fully fused
    get_loss == the reference-style two-pass get_loss (value, gradients, seen, max_2D_radius)."""
    from activesplat_amd import mapping as M
    from activesplat_amd import setup_camera, synthetic as syn
    g = torch.Generator().manual_seed(0)
    H, W = 150, 120
    im = torch.rand(3, H, W, generator=g).to(hip).requires_grad_(True)
    depth = (torch.rand(1, H, W, generator=g) * 3).to(hip).requires_grad_(True)
    gt_im = torch.rand(3, H, W, generator=g).to(hip); gt_d = (torch.rand(1, H, W, generator=g) * 3).to(hip)
    gt_d[0, :7, :11] = 0.0
    loss, parts = M.fused_mapping_loss(im, depth, depth.detach() ** 2 + 0.1, gt_im, gt_d, dict(im=0.5, depth=1.0))
    loss.backward()
    im2, d2 = im.detach().clone().requires_grad_(True), depth.detach().clone().requires_grad_(True)
    ref = 1.0 * (gt_d - d2).abs()[gt_d > 0].mean() + 0.5 * (0.8 * M.l1_loss_v1(im2, gt_im) + 0.2 * (1.0 - M.calc_ssim(im2, gt_im)))
    ref.backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-5)
    np.testing.assert_allclose(im.grad.cpu().numpy(), im2.grad.cpu().numpy(), atol=3e-9, rtol=3e-3)
    np.testing.assert_allclose(depth.grad.cpu().numpy(), d2.grad.cpu().numpy(), atol=1e-10, rtol=1e-5)
    # full iteration loss: all fused vs none
    N, W, H = 40_000, 256, 192
    p = syn.make_params(N, W, H, seed=1)
    out = []
    for flags in (dict(), dict(fused=True, fused_loss=True, fused_inputs=True)):
        params = {k: torch.nn.Parameter(v.to(hip)) for k, v in p.items()}
        params["cam_unnorm_rots"] = torch.nn.Parameter(torch.tensor([0.98, 0.02, 0.15, -0.03], device=hip).reshape(1, 4, 1).repeat(1, 1, 2))
        params["cam_trans"] = torch.nn.Parameter(torch.tensor([0.05, -0.02, 0.1], device=hip).reshape(1, 3, 1).repeat(1, 1, 2))
        variables = {k: torch.zeros(N, device=hip) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
        tim, tdepth = syn.make_targets(W, H)
        data = dict(cam=setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=hip), im=tim.to(hip), depth=tdepth.to(hip), id=1,
                    w2c=torch.eye(4, device=hip))
        loss, variables, _ = M.get_loss(params, data, variables, 1, dict(im=0.5, depth=1.0), **flags)
        loss.backward()
        out.append((float(loss.detach()), {k: v.grad.clone() for k, v in params.items() if v.grad is not None}, variables))
    assert abs(out[0][0] - out[1][0]) < 1e-5 * abs(out[0][0])
    assert torch.equal(out[0][2]["seen"], out[1][2]["seen"]) and torch.equal(out[0][2]["max_2D_radius"], out[1][2]["max_2D_radius"])
    for k in ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales"):
        a, b = out[0][1][k], out[1][1][k]
        assert float((a - b).norm() / a.norm()) < 1e-3, k


def test_look_around_fused_equals_reference_pattern_on_gpu(hip):
    import numpy as np
    from activesplat_amd import lookaround as LA, synthetic as syn
    params = {k: v.to(hip) for k, v in syn.shell_scene(200_000, seed=2, W=LA.LOOK_W, H=LA.LOOK_H).items()}
    c2w = np.eye(4); c2w[:3, 3] = [0.1, 0.0, -0.2]
    a, b = LA.look_around(params, c2w, fused=True), LA.look_around(params, c2w, fused=False)
    assert a["opacity"].shape == (150, 360) and a["rgb"].dtype == torch.uint8
    assert torch.allclose(a["opacity"], b["opacity"], atol=1e-4)
    assert float((a["depth"] - b["depth"]).abs().max()) <= 1e-3 * float(b["depth"].abs().max())
    assert float(((a["rgb"].int() - b["rgb"].int()).abs() > 1).float().mean()) < 1e-3
    assert abs(LA.local_invisibility(params, c2w) - float((1 - b["opacity"]).sum())) <= 1e-3 * 150 * 360
    # the multi-view atlas (one raster pass for the three views) against one pass per view, on a map whose tile lists are far beyond
    # the LDS sort capacity (pairwise merge passes, segmented compositing)
    params = {k: v.to(hip) for k, v in syn.shell_scene(1_000_000, seed=3, W=LA.LOOK_W, H=LA.LOOK_H).items()}
    a, c = LA.look_around(params, c2w, fused=True, batched=True), LA.look_around(params, c2w, fused=True, batched=False)
    assert torch.allclose(a["opacity"], c["opacity"], atol=1e-4)
    assert float((a["depth"] - c["depth"]).abs().max()) <= 1e-3 * float(c["depth"].abs().max())
    assert float(((a["rgb"].int() - c["rgb"].int()).abs() > 1).float().mean()) < 1e-3


def test_height_cut_on_gpu(hip):
    from activesplat_amd import io as IO, synthetic as syn
    p = {k: v.to(hip) for k, v in syn.make_params(100_000, 640, 480, seed=2).items()}
    ref = {k: v.clone() for k, v in p.items()}
    cond = torch.logical_or(-ref["means3D"][:, 1] < -0.3, -ref["means3D"][:, 1] > 0.4)
    got = IO.cut_gaussian_by_height(p, -0.3, 0.4)
    for k in IO.GAUSSIAN_ROW_KEYS:
        assert torch.equal(got[k], ref[k][~cond]), k


def test_growth_kernel_equals_torch_growth_on_gpu(hip):
    """640x480 frame: gs_grow_gaussians vs the reference-pattern torch masks / median / gathers on the same render."""
    import numpy as np
    from activesplat_amd import mapping as M, synthetic as syn
    W, H = 640, 480
    g = torch.Generator().manual_seed(3)
    gt = (torch.rand(1, H, W, generator=g) * 3.5 + 0.5)
    gt[0, :40] = 0.0                                                    # invalid-depth band
    rd = gt[0] + torch.randn(H, W, generator=g) * 0.05
    sil = torch.rand(H, W, generator=g)
    color = torch.rand(3, H, W, generator=g)
    K = torch.tensor(syn.intrinsics(W, H), dtype=torch.float32)
    pose7 = [0.98, 0.05, -0.17, 0.02, 0.3, -0.1, 0.2]
    n = float(np.linalg.norm(pose7[:4])); pose7 = [v / n for v in pose7[:4]] + pose7[4:]
    c2w = M._c2w_from_pose7(pose7)
    gt_d, rd_d, sil_d, col_d = gt.to(hip), rd.to(hip), sil.to(hip), color.to(hip)
    rows, n_cand = M.grow_rows(rd_d, sil_d, gt_d, col_d, K, c2w, 0.5, "anisotropic")
    # torch restatement of splatam.py:340-364 on the same images
    err = (gt_d[0] - rd_d).abs() * (gt_d[0] > 0)
    non = (sil_d < 0.5) | ((rd_d > gt_d[0]) & (err > 2 * err.median()) & (sil_d > 0.5) & (gt_d[0] < 5))
    assert n_cand == int(non.sum())
    mask = (non & (gt_d[0] > 0)).reshape(-1)
    w2c = torch.tensor(np.linalg.inv(c2w), dtype=torch.float32, device=hip)
    cld, msd = M.get_pointcloud(col_d, gt_d, K.to(hip), w2c, mask=mask, compute_mean_sq_dist=True)
    ref = M._new_gaussians(cld, msd, "anisotropic")
    assert rows["means3D"].shape[0] == int(mask.sum()) > 1000
    for k in ref:
        assert torch.allclose(rows[k], ref[k], atol=5e-6, rtol=1e-5), k


def test_keyframe_overlap_kernel_equals_torch_on_gpu(hip):
    import numpy as np
    from activesplat_amd import synthetic as syn
    from activesplat_amd.keyframes import keyframe_selection_overlap
    W, H = 256, 256
    g = torch.Generator().manual_seed(5)
    depth = (torch.rand(1, H, W, generator=g) * 3 + 0.5).to(hip)
    K = torch.tensor(syn.intrinsics(W, H), dtype=torch.float32, device=hip)
    kfs = [{"id": i, "est_w2c": torch.tensor(syn.keyframe_w2c(i, 24), dtype=torch.float32, device=hip)} for i in range(24)]
    sampled = torch.randint(H * W, (1600,), generator=g)
    args = (depth, torch.eye(4, device=hip), K, kfs, 8)
    a, ra = keyframe_selection_overlap(*args, sampled=sampled, shuffle=False, return_percent=True)
    from activesplat_amd import keyframes as KF
    from tests import reference_pattern as RP
    valid = torch.stack(torch.where(depth[0] > 0), dim=1)
    pts = KF.drop_repeated_points(KF.world_points(depth, K, args[1], valid[sampled.to(hip)]))
    cnt = RP.keyframe_overlap_torch(pts, kfs, K, W, H)
    b = [i for i, c in sorted(enumerate(cnt), key=lambda t: -t[1]) if c > 0][:8]
    pa = {r["id"]: float(r["percent_inside"]) for r in ra}
    pb = {i: c / pts.shape[0] for i, c in enumerate(cnt)}
    assert max(abs(pa[i] - pb[i]) for i in pa) <= 2 / 1600            # at most a borderline point or two per keyframe
    assert len(a) == len(b) == 8


def test_adam_streaming_path_for_tensors_beyond_the_last_level_cache(hip):
    """> 256 MiB of Adam traffic per tensor switches the kernel to non-temporal loads / stores: same arithmetic."""
    from activesplat_amd import optim as O
    n = 10_000_003                                                  # 280 MB of traffic; odd tail
    g0 = torch.Generator().manual_seed(1)
    init = torch.randn(n, generator=g0)
    a = torch.nn.Parameter(init.clone().to(hip)); b = torch.nn.Parameter(init.clone().to(hip))
    oa = O.GaussianAdam([{"params": [a], "name": "x", "lr": 2e-3}], lr=0.0, eps=1e-15)
    ob = torch.optim.Adam([{"params": [b], "lr": 2e-3}], lr=0.0, eps=1e-15)
    for _ in range(3):
        g = torch.randn(n, generator=g0).to(hip)
        a.grad, b.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    assert torch.allclose(a, b, rtol=3e-6, atol=1e-7)
    assert torch.allclose(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"], rtol=3e-6, atol=1e-12)


@pytest.mark.gpu
def test_configs2_optimise_loop_2m_sh3(hip):
    """BASELINE configs[2] at full size: 2 M Gaussians, SH degree 3, 640x480, 100 iterations with fused Adam and a densify event at
    iteration 50 -- with the fused densify (one classification, one gather per tensor) and with the reference's step-by-step call
    pattern.  Both must end with the same number of Gaussians after the event and a lower loss than they started with."""
    from tests import util
    torch.manual_seed(0)
    a = util.configs2_optimise_loop(2_000_000, 100, "cuda")
    torch.manual_seed(0)
    from tests import reference_pattern as RP
    b = util.configs2_optimise_loop(2_000_000, 100, "cuda", densify_fn=RP.densify_stepwise)
    assert a["counts"] and a["counts"] == b["counts"], (a["counts"], b["counts"])
    assert a["counts"][0] != 2_000_000
    for r in (a, b):
        assert np.isfinite(r["losses"]).all() and r["losses"][1] < r["losses"][0], r["losses"]
    assert abs(a["losses"][1] - b["losses"][1]) < 2e-2 * abs(b["losses"][1])      # the split offsets are drawn from different streams


@pytest.mark.gpu
def test_mapper_on_hip_kernels_equals_mapper_on_the_oracle(hip, oracle32, monkeypatch):
    """configs[4] substitute, second half, on the device: the mapping loop on the HIP kernels against the same loop on
    a rasteriser backed by the C oracle (test-only shim) -- same schedule, same map size, re-render PSNR within 0.1 dB."""
    from activesplat_amd import mapping as M
    from tests import util
    from tests.test_mapper import run_harness, run_harness_on
    a, seq, log_a = run_harness(hip, n_gt=4000, W=64, H=48, frames=11)
    monkeypatch.setattr(M, "Renderer", util.oracle_rasterizer_class(oracle32))
    b, _, log_b = run_harness_on(seq, 64, 48, hip)
    monkeypatch.undo()
    assert [(e["iters"], e["new_opt"], e["keyframes"]) for e in log_a] == [(e["iters"], e["new_opt"], e["keyframes"]) for e in log_b]
    na, nb = a.params["means3D"].shape[0], b.params["means3D"].shape[0]
    assert abs(na - nb) <= max(3, 0.003 * nb), (na, nb)
    for fr in (seq[0], seq[9]):
        seen = (fr["depth"] > 0)[0].cpu().numpy()
        gt = fr["color"].cpu().numpy()[:, seen]
        pa = util.psnr(a.render_rgbd(fr["w2c"])[0].cpu().numpy()[:, seen], gt)
        pb = util.psnr(b.render_rgbd(fr["w2c"])[0].cpu().numpy()[:, seen], gt)
        assert abs(pa - pb) < 0.1 and pa > 18.0, (pa, pb)
    assert a.high_loss_mask is not None and int((a.high_loss_mask != b.high_loss_mask).sum()) <= 3
