"""CPU (host-emulated kernels): the mapping-loop harness follows the reference's schedule
(src/mapper/splatam/__init__.py:395-524) on a synthetic RGB-D spin and the map it builds re-renders the
sequence.  The same harness runs on the GPU in tests/test_gpu_parity.py::test_mapper_harness_gpu."""
import numpy as np
import torch

from tests import util


def run_harness(device, n_gt=6000, W=64, H=48, frames=11, cfg=None):
    from activesplat_amd import synthetic as syn
    from activesplat_amd.mapper import SplatMapper
    gt = syn.shell_scene(n_gt, seed=2, W=W, H=H)
    gt["logit_opacities"] = gt["logit_opacities"] + 3.0            # mostly opaque surfaces
    seq = list(syn.orbit_sequence(gt, frames, W, H, device))
    mp = SplatMapper(syn.intrinsics(W, H), W, H, config=dict(step_num=frames, **(cfg or {})), device=device)
    log = []
    for fr in seq:
        n_before = 0 if mp.params is None else mp.params["means3D"].shape[0]
        opt_before = mp.optimizer
        it_before = mp.stats["iters"]
        mp.run(fr)
        log.append(dict(id=fr["id"], grew=mp.params["means3D"].shape[0] - n_before, new_opt=mp.optimizer is not opt_before,
                        iters=mp.stats["iters"] - it_before, keyframes=len(mp.keyframe_list)))
    return mp, seq, log


def test_mapper_schedule_and_rerender(emu):
    mp, seq, log = run_harness(emu)
    # schedule of the shipped config: map_every = keyframe_every = 5, mapping_iters = 2
    for e in log:
        fid = e["id"]
        assert e["iters"] == (2 if fid % 5 == 0 else 0), e                        # int(2 // 5) == 0 -> 2 iters on frames 0,5,10
        assert e["new_opt"] == (fid == 0 or (fid + 1) % 5 == 0), e                # fresh Adam on densify frames
        if fid > 0 and (fid + 1) % 5 != 0:
            assert e["grew"] == 0, e
    assert [e["keyframes"] for e in log][-1] == 3                                # keyframes 0, 4, 9 (9 is also step_num - 2)
    assert any(e["grew"] > 0 for e in log if e["id"] in (4, 9))                   # the spin reveals unseen space
    # re-render frame 0 and the last densify frame from the built map
    for fr in (seq[0], seq[9]):
        im, depth, opacity = mp.render_rgbd(fr["w2c"])
        seen = fr["depth"] > 0
        assert util.psnr(im.cpu().numpy()[:, seen[0].cpu().numpy()], fr["color"].cpu().numpy()[:, seen[0].cpu().numpy()]) > 18.0
        assert float(opacity[seen].mean()) > 0.8
    inv = mp.invisibility(seq[0]["w2c"])
    assert inv.shape == (1, 150, 120) and float(inv.min()) >= 0.0 and float(inv.max()) <= 1.0
    assert all(np.isfinite(v) for v in mp.last_losses.values())
    # hand-off artefacts: packet for the visualiser queue, params.npz for the offline tools, look-around panorama
    import tempfile
    from activesplat_amd import io as IO
    pk = mp.packet(c2w=np.eye(4))
    assert pk.has_gaussians and pk.params is mp.params
    with tempfile.TemporaryDirectory() as d:
        z = np.load(mp.post_processing(d))
        assert set(z.files) == set(IO.FILE_KEYS)
        assert z["cam_trans"].shape[-1] == len(seq) == z["gt_w2c_all_frames"].shape[0]
        assert list(z["keyframe_time_indices"]) == [0, 4, 9]
        assert z["means3D"].shape[0] == mp.params["means3D"].shape[0] == z["timestep"].shape[0]
    pano = mp.look_around(np.eye(4))
    assert pano["opacity"].shape == (150, 360) and float(pano["opacity"].max()) > 0.5


def test_fused_growth_and_keyframe_scoring_build_the_same_map(emu):
    """Harness with the HIP growth / keyframe-overlap kernels vs the reference-pattern torch code: same schedule, same
    keyframes, the same number of Gaussians up to threshold flips of single pixels."""
    ref, seq, log_ref = run_harness(emu, n_gt=3000, frames=6)
    fus, _, log_fus = run_harness(emu, n_gt=3000, frames=6, cfg=dict(fused_growth=True, fused_keyframes=True))
    assert [e["keyframes"] for e in log_ref] == [e["keyframes"] for e in log_fus]
    n_ref, n_fus = ref.params["means3D"].shape[0], fus.params["means3D"].shape[0]
    assert abs(n_ref - n_fus) <= max(3, 0.005 * n_ref), (n_ref, n_fus)
    grew = [e["grew"] for e in log_fus]
    assert grew[4] > 0
    # every appended Gaussian sits (up to the two Adam steps it has since taken, lr 1e-4) on the sensor ray of a valid
    # pixel of frame 4 at its measured depth
    n0 = n_fus - grew[4] if grew[5] == 0 else None
    if n0 is not None:
        fr = seq[4]
        new = fus.params["means3D"].detach()[n0:]
        w2c = torch.as_tensor(np.asarray(fr["w2c"]), dtype=torch.float32, device=new.device)
        cam = new @ w2c[:3, :3].T + w2c[:3, 3]
        K = fus.intrinsics
        u = cam[:, 0] / cam[:, 2] * K[0, 0] + K[0, 2]
        v = cam[:, 1] / cam[:, 2] * K[1, 1] + K[1, 2]
        assert float((u - u.round()).abs().max()) < 0.05 and float((v - v.round()).abs().max()) < 0.05
        z = fr["depth"].to(new.device)[0, v.round().long(), u.round().long()]
        assert torch.allclose(z, cam[:, 2], rtol=2e-3, atol=2e-3) and bool((z > 0).all())


def test_mapper_with_the_raw_parameter_rasteriser_builds_the_same_map(emu):
    """fused_preprocess (the per-Gaussian kernels take the parameters; no activation launches) against the activation kernels in front of
    the rasteriser: same schedule and keyframes, map sizes within threshold flips, parameters close after the run."""
    base = dict(fused_render=True, fused_loss=True, fused_inputs=True)
    a, _, log_a = run_harness(emu, n_gt=2500, W=48, H=40, frames=6, cfg=base)
    b, _, log_b = run_harness(emu, n_gt=2500, W=48, H=40, frames=6, cfg=dict(base, fused_preprocess=True))
    assert [e["keyframes"] for e in log_a] == [e["keyframes"] for e in log_b]
    na, nb = a.params["means3D"].shape[0], b.params["means3D"].shape[0]
    assert abs(na - nb) <= max(3, 0.005 * na), (na, nb)
    if na == nb:
        for k in ("means3D", "rgb_colors", "logit_opacities", "log_scales"):
            d = float((a.params[k].detach() - b.params[k].detach()).abs().max())
            assert d < 5e-3 * max(1.0, float(a.params[k].detach().abs().max())), (k, d)


def test_mapper_with_adam_inside_the_backward_builds_the_same_map(emu):
    """fused_adam: every mapping iteration's Adam step inside the raw-parameter backward kernel against the separate optimizer.step()
    (on ONE emulator thread the two loops are the same arithmetic in the same order: identical maps, bit for bit)."""
    import ctypes
    omp = ctypes.CDLL("libgomp.so.1")
    before = omp.omp_get_max_threads()
    omp.omp_set_num_threads(1)
    try:
        base = dict(fused_render=True, fused_loss=True, fused_inputs=True, fused_preprocess=True)
        a, _, log_a = run_harness(emu, n_gt=2000, W=48, H=40, frames=6, cfg=base)
        b, _, log_b = run_harness(emu, n_gt=2000, W=48, H=40, frames=6, cfg=dict(base, fused_adam=True))
        c, _, log_c = run_harness(emu, n_gt=2000, W=48, H=40, frames=6, cfg=dict(base, fused_adam=True, fused_iteration=True))   # ... and without autograd
    finally:
        omp.omp_set_num_threads(before)
    assert log_a == log_b == log_c and a.stats["iters"] == b.stats["iters"] == c.stats["iters"] > 0
    for other in (b, c):
        for k in ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales"):
            assert torch.equal(a.params[k].detach(), other.params[k].detach()), k
            sa, sb = a.optimizer.state[a.params[k]], other.optimizer.state[other.params[k]]
            assert int(sa["step"]) == int(sb["step"]) and torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), k


def test_raw_frames_with_densification_resolution(emu):
    """run_raw: uint8 image + metric depth + pose in, resized mapping / densification copies out; the map is seeded and
    grown at the (half) densification resolution while the loss runs at the mapping resolution."""
    from activesplat_amd import synthetic as syn
    from activesplat_amd.mapper import SplatMapper
    W, H, frames = 64, 48, 6
    gt = syn.shell_scene(3000, seed=2, W=W, H=H)
    gt["logit_opacities"] = gt["logit_opacities"] + 3.0
    seq = list(syn.orbit_sequence(gt, frames, W, H, emu))
    mp = SplatMapper(syn.intrinsics(W, H), W, H, config=dict(step_num=frames, densify_downscale_factor=2), device=emu)
    for fr in seq:
        image = (fr["color"].permute(1, 2, 0).clamp(0, 1) * 255).round().to(torch.uint8).cpu().numpy()
        depth = fr["depth"][0].cpu().numpy()
        mp.run_raw(image, depth, np.linalg.inv(np.asarray(fr["w2c"], dtype=np.float64)), fr["id"], fr["quat"], fr["position"])
    assert (mp.densify_cam.image_width, mp.densify_cam.image_height) == (32, 24)
    assert float(mp.densify_intrinsics[0, 0]) == float(mp.intrinsics[0, 0]) / 2
    valid0 = int((torch.as_tensor(seq[0]["depth"])[0, ::2, ::2] > 0).sum())
    assert mp.params["means3D"].shape[0] >= valid0                  # seeded from the 32x24 copy of frame 0, then grown
    assert mp.params["means3D"].shape[0] < 32 * 24 * 3
    im, depth, opacity = mp.render_rgbd(seq[0]["w2c"])
    assert im.shape == (3, H, W) and float(opacity.mean()) > 0.3
    assert len(mp.gt_w2c_all_frames) == frames


def test_mapper_on_emulated_kernels_equals_mapper_on_the_oracle(emu, oracle32, monkeypatch):
    """configs[4] substitute, second half: the SAME mapping loop driven twice over the SAME synthetic RGB-D spin --
    once on this build's kernels (host-emulated here; the GPU twin is in tests/test_gpu_parity.py), once with a rasteriser whose
    forward and backward are the C oracle (test-only shim).  The loops must build the same map: identical schedule, the same
    number of Gaussians up to single-pixel threshold flips, and re-renders of the sequence within 0.1 dB of each other."""
    from activesplat_amd import mapping as M
    a, seq, log_a = run_harness(emu, n_gt=2500, W=48, H=40, frames=6)
    monkeypatch.setattr(M, "Renderer", util.oracle_rasterizer_class(oracle32))
    b, _, log_b = run_harness_on(seq, 48, 40, emu)
    monkeypatch.undo()
    assert [(e["iters"], e["new_opt"], e["keyframes"]) for e in log_a] == [(e["iters"], e["new_opt"], e["keyframes"]) for e in log_b]
    na, nb = a.params["means3D"].shape[0], b.params["means3D"].shape[0]
    assert abs(na - nb) <= max(2, 0.003 * nb), (na, nb)
    for fr in (seq[0], seq[4]):
        seen = (fr["depth"] > 0)[0].cpu().numpy()
        gt = fr["color"].cpu().numpy()[:, seen]
        pa = util.psnr(a.render_rgbd(fr["w2c"])[0].cpu().numpy()[:, seen], gt)
        pb = util.psnr(b.render_rgbd(fr["w2c"])[0].cpu().numpy()[:, seen], gt)          # b's map rendered by the product rasteriser
        assert abs(pa - pb) < 0.1 and pa > 18.0, (pa, pb)
    # the per-frame high-loss consumer ran on both
    assert a.high_loss_mask is not None and a.high_loss_mask.shape == (40, 48) and a.high_loss_mask.dtype == torch.bool
    assert int((a.high_loss_mask != b.high_loss_mask).sum()) <= 2


def run_harness_on(seq, W, H, device, cfg=None):
    from activesplat_amd import synthetic as syn
    from activesplat_amd.mapper import SplatMapper
    mp = SplatMapper(syn.intrinsics(W, H), W, H, config=dict(step_num=len(seq), **(cfg or {})), device=device)
    log = []
    for fr in seq:
        opt_before, it_before = mp.optimizer, mp.stats["iters"]
        mp.run(fr)
        log.append(dict(id=fr["id"], new_opt=mp.optimizer is not opt_before, iters=mp.stats["iters"] - it_before, keyframes=len(mp.keyframe_list)))
    return mp, seq, log
