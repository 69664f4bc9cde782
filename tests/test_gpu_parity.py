"""GPU (-m gpu): the parity tests proper.  The HIP path (through the C ABI) vs the oracle on identical
seeded inputs: every edge case of tests/parity_cases.py, a mid-size scene, BASELINE configs[1] at full
size (500k Gaussians, 640x480), and size-independent properties at full size."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import parity_cases as pc
from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", ["auto", "radix"])
@pytest.mark.parametrize("case", pc.CASES)
def test_forward_matches_oracle(hip, oracle32, case, path):
    rs, rv = pc.build_case(case, hip)
    pc.set_sort_path(path)
    try:
        pc.check_forward(rs, rv, oracle32)
        assert util.artefacts()["path"] == (2 if path == "radix" else 1)
    finally:
        pc.set_sort_path("auto")


@pytest.mark.parametrize("case", pc.CASES)
def test_backward_matches_fp64_oracle(hip, oracle64, oracle32, case):
    rs, rv = pc.build_case(case, hip)
    # (the top-down camera -- focal length 2e4 px, depths ~1000 -- leaves the fp32 ORACLE at 0.9953-0.997 of the elements inside the
    # tolerance: there, and only there, a miss is judged against the fp32 oracle's own error and tallied)
    pc.check_backward(rs, rv, oracle64, oracle32=oracle32 if case.startswith("topdown") else None)


def test_midsize_scene_forward_backward(hip, oracle32, oracle64):
    rs, rv = util.scene(60_000, 320, 240, seed=11, device=hip, w2c=util.pose(0.05, (0.02, 0.0, 0.1)), bg=(0.05, 0.1, 0.2))
    pc.check_forward(rs, rv, oracle32)
    pc.check_backward(rs, rv, oracle64)


def _full_size_gradients_vs_fp64(rs, rv, got_ref32, oracle64, H, W):
    """SURVEY 8(d)'s stated bar at full size: gradients vs the **fp64** oracle, rtol 1e-3 / atol 1e-6 |g|_inf on >= 99.5 % of the
    elements and relative L2 error < 1e-3 (the oracle's OpenMP threads are raised for this one call: fp64 sums do not care)."""
    dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1))
    g = util.run_product(rs, rv, dL)["grads"]
    oracle64.set_threads(16)
    try:
        r = util.run_oracle(oracle64, rs, rv, dL)["grads"]
    finally:
        oracle64.set_threads(1)
    for k, a in g.items():
        b = r[k].reshape(a.shape)
        rel = np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b)
        frac = util.close_frac(a, b, pc.GRAD_RTOL, 1e-6 * float(np.abs(b).max()))
        print(f"full-size gradient {k}: rel L2 {rel:.3e}, within tolerance {frac:.5f}")
        assert rel < 1e-3, (k, rel)
        assert frac >= 0.995, (k, frac)
    return g


def test_full_size_config1_against_oracle(hip, oracle32, oracle64):
    """BASELINE configs[1]: 500k Gaussians, 640x480, SH-0 -- forward vs the fp32 oracle (integers exact),
    backward vs the fp64 oracle at the stated tolerance."""
    rs, rv = util.scene(500_000, 640, 480, seed=0, device=hip)
    got, ref = pc.check_forward(rs, rv, oracle32)
    _full_size_gradients_vs_fp64(rs, rv, ref, oracle64, 480, 640)


def test_configs0_hip_forward_vs_cpu_pytorch_render(hip):
    """BASELINE configs[0]: 10k random Gaussians, one 640x480 view, forward only -- the HIP forward against the plain CPU PyTorch
    render of the same scene (oracle/dense_torch.render_dense(tiled=True), independent of the C oracle)."""
    from oracle import dense_torch as DT
    rs, rv = util.scene(10_000, 640, 480, seed=0, device="cpu")
    with torch.no_grad():
        ref = DT.render_dense(util.cam_dict(rs), rv["means3D"], rv["opacities"], colors=rv["colors_precomp"], scales=rv["scales"],
                              rotations=rv["rotations"], tiled=True)
    rs_d, rv_d = util.scene(10_000, 640, 480, seed=0, device=hip)
    got = util.run_product(rs_d, rv_d)
    for k_got, k_ref in (("color", "color"), ("depth", "depth"), ("opacity", "opacity")):
        a, b = got[k_got].reshape(-1), ref[k_ref].detach().cpu().numpy().astype(np.float32).reshape(-1)
        scale = max(1.0, float(np.abs(b).max()))
        assert util.close_frac(a, b, pc.FWD_RTOL, pc.FWD_ATOL * scale) >= 0.999, k_got
        assert np.abs(a - b).max() <= 0.02 * scale, k_got
    assert util.psnr(got["color"], ref["color"].detach().cpu().numpy()) >= 60.0
    live = ref["radii"].cpu().numpy() > 0 if "radii" in ref else None
    if live is not None:
        assert np.array_equal(got["radii"] > 0, live)


def test_full_size_properties(hip):
    """Size-independent properties at BASELINE size (no oracle): sortedness of the keys, ranges partition
    [0,D), sum(tiles_touched) == D == offsets[-1], silhouette channel == opacity output, depth channel ==
    depth output, determinism of the forward, linearity of the backward in dL/dcolor."""
    N, W, H = 500_000, 640, 480
    rs, rv = util.scene(N, W, H, seed=0, device=hip)
    z = rv["means3D"][:, 2]
    rv["colors_precomp"] = torch.stack([z, torch.ones_like(z), z * z], 1).contiguous()   # w2c = I: z_cam = z
    dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(3))
    a = util.run_product(rs, rv, dL)
    art = util.artefacts()
    D = a["D"]
    keys = art["keys_sorted"]
    assert (keys[1:] >= keys[:-1]).all()
    assert int(art["tiles_touched"].sum()) == D == int(art["ranges"][:, 1].max())
    rg = art["ranges"]; ne = rg[rg[:, 1] > rg[:, 0]]
    assert ne[0, 0] == 0 and ne[-1, 1] == D and (ne[1:, 0] == ne[:-1, 1]).all()
    tile_of = (keys >> np.uint64(32)).astype(np.int64)
    assert (np.repeat(np.arange(len(rg)), (rg[:, 1] - rg[:, 0]).astype(np.int64)) == tile_of).all()
    np.testing.assert_allclose(a["color"][1], a["opacity"][0], atol=3e-6)
    np.testing.assert_allclose(a["color"][0], a["depth"][0], rtol=1e-5, atol=3e-5)
    assert ((a["radii"] > 0) == (art["tiles_touched"] > 0)).all()
    b = util.run_product(rs, rv, 2.0 * dL)
    assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["radii"], b["radii"])      # deterministic forward
    for k in ("means3D", "opacities", "scales", "means2D"):
        ga, gb = a["grads"][k].astype(np.float64), b["grads"][k].astype(np.float64)
        assert np.linalg.norm(gb - 2 * ga) / np.linalg.norm(2 * ga) < 1e-4, k                      # atomics reorder sums only


def test_topdown_camera_full_size_properties(hip):
    """The planner's top-down view at size (1 M Gaussians, 360 x 300, 1000 m, scale_modifier 0.01, far = 100): every visible splat collapses
    to the low-pass footprint (radius 3), nothing is far-culled, the opacity map is not empty, keys are sorted and the ranges partition
    [0, D), and the image is affine in the background (white-background colour = black-background colour + 1 - opacity)."""
    N = 1_000_000
    rs, rv = pc.topdown_scene(N, hip)
    a = util.run_product(rs, rv)
    art = util.artefacts()
    assert set(np.unique(a["radii"]).tolist()) <= {0, 3}
    inside = ((rv["means3D"][:, 0] - 0.3).abs() < 8.9) & ((rv["means3D"][:, 2] + 0.2).abs() < 7.4)
    assert (a["radii"][inside.cpu().numpy()] == 3).all()                # no far cull, no near cull: everything over the footprint renders
    assert float((a["opacity"] > 0.5).mean()) > 0.9
    keys = art["keys_sorted"]
    assert (keys[1:] >= keys[:-1]).all() and int(art["tiles_touched"].sum()) == a["D"]
    rg = art["ranges"]; ne = rg[rg[:, 1] > rg[:, 0]]
    assert ne[0, 0] == 0 and ne[-1, 1] == a["D"] and (ne[1:, 0] == ne[:-1, 1]).all()
    b = util.run_product(rs._replace(bg=torch.ones(3, device=hip)), rv)
    assert np.array_equal(a["opacity"], b["opacity"]) and np.array_equal(a["depth"], b["depth"])
    np.testing.assert_allclose(b["color"], a["color"] + (1.0 - a["opacity"]), atol=2e-6)


def test_adam_matches_torch_on_gpu(hip):
    from activesplat_amd import _lib
    lib = _lib.get()
    torch.manual_seed(0)
    n = 14 * 100_003
    p = torch.randn(n, device=hip); ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": [ref], "lr": 1e-3}], lr=0.0, eps=1e-15)
    m, v = torch.zeros(n, device=hip), torch.zeros(n, device=hip)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for step in range(1, 5):
        g = torch.randn(n, device=hip)
        ref.grad = g.clone(); opt.step()
        _lib.check(lib.gs_adam_step(n, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), 1e-3, 0.9, 0.999, 1e-15, step, st))
        torch.cuda.synchronize()
        np.testing.assert_allclose(p.cpu().numpy(), ref.detach().cpu().numpy(), rtol=3e-6, atol=1e-7)


def test_reference_call_pattern_two_passes(hip, oracle32):
    """The reference's get_loss call pattern (splatam.py:208-212): RGB pass + [z,1,z^2] pass on the same
    geometry, both differentiated, means2D.grad retained from the RGB pass only."""
    from activesplat_amd import GaussianRasterizer
    rs, rv = util.scene(20_000, 256, 256, seed=5, device=hip)
    rs = rs._replace(debug=False)
    inp = {k: v.clone().requires_grad_(True) for k, v in rv.items()}
    m2d = torch.zeros(20_000, 3, device=hip, requires_grad=True)
    im, radius, _, _ = GaussianRasterizer(raster_settings=rs)(means2D=m2d, **inp)
    z = inp["means3D"][:, 2:3]
    ds = dict(inp, colors_precomp=torch.cat([z, torch.ones_like(z), z * z], 1))
    m2d_b = torch.zeros(20_000, 3, device=hip, requires_grad=True)
    depth_sil, _, _, _ = GaussianRasterizer(raster_settings=rs)(means2D=m2d_b, **ds)
    loss = im.abs().mean() + (depth_sil[0] - 2.0).abs().mean()
    loss.backward()
    seen = radius > 0
    assert torch.isfinite(m2d.grad).all() and m2d.grad[seen, :2].abs().sum() > 0 and (m2d.grad[:, 2] == 0).all()
    assert torch.max(radius[seen], torch.zeros_like(radius[seen], dtype=torch.float32)).numel() == int(seen.sum())
    for k in ("means3D", "scales", "rotations", "opacities", "colors_precomp"):
        assert torch.isfinite(inp[k].grad).all() and inp[k].grad.abs().sum() > 0


@pytest.mark.parametrize("case", ["basic", "posed_white_bg", "ragged_image", "sh2", "dense_overdraw", "mixed_sizes", "huge_gaussians", "tiny_lookaround"])
def test_fused_rgbd_matches_two_passes_and_oracle(hip, oracle64, case):
    rs, rv = pc.build_case(case, hip)
    pc.check_fused_rgbd(rs, rv, oracle64)


@pytest.mark.parametrize("case,path", [("merge_tiles", 1), ("merge_tiles_large", 1), ("merge_passes", 1), ("merge_passes_even", 1), ("bucket_lists", 1),
                                       ("bucket_lists_long", 1), ("crowded_depth", 1), ("crowded_depth_long", 1), ("equal_depth", 1), ("bucket_lists_big", 1), ("crowded_depth_big", 1)])
def test_big_tile_lists(hip, oracle32, oracle64, case, path):
    rs, rv = pc.build_case(case, hip)
    pc.check_forward(rs, rv, oracle32)
    assert util.artefacts()["path"] == path
    # (equal_depth: view depths equal up to an ulp order differently in fp32 and fp64, so the fp64 oracle blends in another order -- the fp32
    # oracle misses the fp64 gradients by exactly what the kernel does, 2.7e-3: judged against the fp32 oracle's own error there)
    pc.check_backward(rs, rv, oracle64, oracle32=oracle32 if case == "equal_depth" else None)


def test_more_tiles_than_the_lds_histogram_holds(hip, oracle32, oracle64):
    rs, rv = pc.build_case("many_tiles", hip)
    pc.check_forward(rs, rv, oracle32)
    assert util.artefacts()["path"] == 2
    # sub-pixel Gaussians at this focal length: the fp32 ORACLE itself sits at 0.988-0.995 of the fp64 one here -- the stated 0.995 bar with
    # the (tallied) fp32 escape hatch instead of a looser bar
    pc.check_backward(rs, rv, oracle64, oracle32=oracle32)


@pytest.mark.parametrize("case", ["equal_depth", "crowded_depth_big", "merge_passes_even"])
def test_radix_path_is_stable_on_long_tile_lists(hip, oracle32, case):
    """The radix binning path (csrc/sort_radix.hip: hand-written LSD sort, round 6) on the scenes built for the per-tile sorts: every key of
    `equal_depth` differs from its tile neighbours in nothing -- the order inside a tile is the sort's stability through its passes --, the others
    hold tile lists of 10 k / 75 k keys with clustered depths.  keys, ids and ranges bit-exact against the oracle's stable sort."""
    rs, rv = pc.build_case(case, hip)
    pc.set_sort_path("radix")
    try:
        pc.check_forward(rs, rv, oracle32)
        assert util.artefacts()["path"] == 2
    finally:
        pc.set_sort_path("auto")


def test_zero_gaussians(hip):
    from activesplat_amd import GaussianRasterizer
    rs, _ = util.scene(4, 70, 50, device=hip, bg=(0.2, 0.4, 0.6))
    e = lambda *s: torch.zeros(*s, device=hip, requires_grad=True)  # noqa: E731
    color, radii, depth, opacity = GaussianRasterizer(raster_settings=rs)(means3D=e(0, 3), means2D=e(0, 3), opacities=e(0, 1),
                                                                          colors_precomp=e(0, 3), scales=e(0, 3), rotations=e(0, 4))
    assert radii.numel() == 0 and float(opacity.abs().max()) == 0 and float(depth.abs().max()) == 0
    assert torch.allclose(color, torch.tensor([0.2, 0.4, 0.6], device=hip).view(3, 1, 1).expand(3, 50, 70))
    color.sum().backward()


def test_optimistic_launch_hit_and_miss_equal_exact_launch(hip):
    pc.check_optimistic_launch(hip)
    pc.check_optimistic_tile_list_growth(hip)


@pytest.mark.parametrize("frontend", [True, False], ids=["frontend", "python-twin"])
def test_render_does_not_depend_on_the_capacity_history(hip, frontend):
    """Round-5 sweep, seeds 130045 / 130237: a small image whose longest tile list sits just under 8192 (7753 / 7885).  The second render of the
    same scene is launched optimistically with the first one's counts + margin -- a tile-list bound of >= 8192, which in an image of few tiles selects
    the SEGMENTED forward (sums added with atomics) although the true lists do not call for it: colour differed from the first render by an ulp in ~1 %
    of the pixels, depending on call history.  Frames whose capacity guess would choose the algorithm now take the exact launch."""
    from activesplat_amd import GaussianRasterizer, rasterizer as R
    from tests.fuzz_scenes import sweep_scene
    rs, rv = sweep_scene(130045, hip)
    rs = rs._replace(debug=False)
    m2d = torch.zeros(rv["means3D"].shape[0], 3, device=hip)
    R.use_frontend = frontend
    try:
        R._capacity.clear()
        with torch.no_grad():
            first = [t.clone() for t in GaussianRasterizer(raster_settings=rs)(means2D=m2d, **rv)]
            assert 7000 < R.last_stats["max_tile_instances"] < 8192
            key = next(iter(R._capacity))
            assert R._capacity[key][1] >= 8192                     # the guess for the next frame crosses the threshold ...
            for _ in range(3):
                again = GaussianRasterizer(raster_settings=rs)(means2D=m2d, **rv)
                for a, b in zip(again, first):
                    assert torch.equal(a, b)                       # ... and the image does not notice
    finally:
        R.use_frontend = True
        R._capacity.clear()


def test_frontend_from_two_host_threads(hip):
    """The reference's threaded layout (a visualiser / planner thread rendering next to the mapper's, visualizer.py:157-160): two host threads, each on
    its own HIP stream, render and back-propagate different scenes through the C++ front-end at the same time (the GIL is released inside it; pinned
    counters and events are per thread and stream, the layout cache is locked).  Every result equals the thread's own serial result."""
    import threading
    from activesplat_amd import GaussianRasterizer
    jobs = []
    for seed, (n, w, h) in enumerate(((30_000, 256, 256), (12_000, 120, 150))):
        rs, rv = util.scene(n, w, h, seed=seed, device=hip)
        rs = rs._replace(debug=False)
        dL = torch.randn(3, h, w, generator=torch.Generator().manual_seed(seed)).to(hip)
        jobs.append((rs, rv, dL))

    def run(rs, rv, dL):
        inp = {k: v.detach().clone().requires_grad_(True) for k, v in rv.items()}
        m2d = torch.zeros(rv["means3D"].shape[0], 3, device=hip, requires_grad=True)
        out = GaussianRasterizer(raster_settings=rs)(means2D=m2d, **inp)
        out[0].backward(dL)
        return [t.detach().clone() for t in out], {k: v.grad.clone() for k, v in inp.items()}
    serial = [run(*j) for j in jobs]
    torch.cuda.synchronize()
    results, errors = [None, None], []

    def worker(i):
        try:
            s = torch.cuda.Stream(device=hip)
            with torch.cuda.stream(s):
                for _ in range(30):
                    results[i] = run(*jobs[i])
            s.synchronize()
        except Exception as e:                       # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for (o_s, g_s), (o_t, g_t) in zip(serial, results):
        for a, b in zip(o_s, o_t):
            assert torch.equal(a, b)
        for k in g_s:
            assert float((g_s[k] - g_t[k]).norm()) <= 2e-5 * float(g_s[k].norm().clamp_min(1e-30)), k


@pytest.fixture()
def python_twin(hip):
    """The drop-in call through rasterizer._RasterizeGaussians (the Python twin of the C++ front-end) for the duration of one test."""
    from activesplat_amd import rasterizer as R
    R.use_frontend = False
    yield hip
    R.use_frontend = True


def test_optimistic_launch_through_the_python_twin(python_twin):
    pc.check_optimistic_launch(python_twin)
    pc.check_optimistic_tile_list_growth(python_twin)


@pytest.mark.parametrize("case", ["basic", "posed_white_bg", "ragged_image", "sh3", "cov3d_precomp", "all_culled", "one_gaussian", "topdown_1000m", "huge_gaussians"])
def test_python_twin_matches_oracle_and_the_frontend(hip, oracle32, oracle64, case):
    """The drop-in call has two hosts above the C ABI: the C++ autograd front-end (csrc/torch_frontend.cpp, the default) and its Python twin
    (rasterizer._RasterizeGaussians).  Same library calls in the same order: forward outputs and integer artefacts bit-identical, gradients equal up
    to the order of the atomic sums -- and the twin passes the oracle checks the front-end passes in every other test of this file."""
    from activesplat_amd import rasterizer as R
    if case not in pc.CASES:
        pytest.skip(f"no case {case}")
    rs, rv = pc.build_case(case, hip)
    dL = torch.randn(3, int(rs.image_height), int(rs.image_width), generator=torch.Generator().manual_seed(2))
    front = util.run_product(rs, rv, dL)
    art_f = util.artefacts()
    R.use_frontend = False
    try:
        twin = util.run_product(rs, rv, dL)
        art_t = util.artefacts()
        pc.check_forward(rs, rv, oracle32)
        pc.check_backward(rs, rv, oracle64, oracle32=oracle32 if case.startswith("topdown") else None)
    finally:
        R.use_frontend = True
    for k in ("color", "depth", "opacity", "radii"):
        assert np.array_equal(front[k], twin[k]), k
    assert front["D"] == twin["D"]
    for k in ("point_list", "ranges", "n_contrib", "tiles_touched", "keys_sorted"):
        assert np.array_equal(art_f[k], art_t[k]), k
    for k, g in front["grads"].items():
        r = twin["grads"][k].astype(np.float64)
        # (two runs of the atomics: all rows but the two worst within 2e-5, everything within 1e-3 -- a screen-filling splat can carry a run-to-run
        # spread of 1e-4 of the tensor by itself, see check_fused_rgbd)
        e2 = ((g.astype(np.float64) - r).reshape(g.shape[0], -1) ** 2).sum(1) if g.shape[0] else np.zeros(1)
        nr = max(np.linalg.norm(r), 1e-30)
        assert np.sqrt(max(e2.sum() - (np.sort(e2)[-2:].sum() if g.shape[0] > 8 else 0.0), 0.0)) <= 2e-5 * nr and np.sqrt(e2.sum()) <= 1e-3 * nr, k


def test_frontend_second_backward_and_no_grad(hip):
    """retain_graph: a second backward through one front-end forward finds its gradient records dirty and has the library fill them (same
    gradients); under no_grad nothing is saved and no records are allocated; means2D = None is accepted (render_views' single-view form)."""
    from activesplat_amd import GaussianRasterizer
    rs, rv = util.scene(8000, 128, 96, seed=3, device=hip)
    rs = rs._replace(debug=False)
    inp = {k: v.clone().requires_grad_(True) for k, v in rv.items()}
    m2d = torch.zeros(8000, 3, device=hip, requires_grad=True)
    color = GaussianRasterizer(raster_settings=rs)(means2D=m2d, **inp)[0]
    dL = torch.randn_like(color)
    g1 = torch.autograd.grad(color, list(inp.values()) + [m2d], dL, retain_graph=True)
    g2 = torch.autograd.grad(color, list(inp.values()) + [m2d], dL)
    for a, b in zip(g1, g2):
        assert torch.isfinite(a).all() and float((a - b).norm()) <= 2e-5 * float(b.norm().clamp_min(1e-30))
    with torch.no_grad():
        c2, r2, d2, o2 = GaussianRasterizer(raster_settings=rs)(means2D=None, **rv)
    assert torch.equal(c2, color.detach()) and not c2.requires_grad


def test_full_size_config2_sh3_against_oracle_and_properties(hip, oracle32, oracle64):
    """BASELINE configs[2]'s render: 2 M Gaussians, SH degree 3, 640x480 -- forward vs the fp32 oracle (integer artefacts
    exact), gradients vs the oracle's fp32 build, plus permutation invariance (a shuffled scene renders the same
    image; radii and gradients follow the permutation)."""
    N, W, H = 2_000_000, 640, 480
    rs, rv = util.scene(N, W, H, seed=0, device=hip, sh_degree=3)
    got, ref = pc.check_forward(rs, rv, oracle32)
    assert util.artefacts()["path"] == 1                       # tile lists of a few thousand: chunk sort + LDS merge
    dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1))
    g = _full_size_gradients_vs_fp64(rs, rv, ref, oracle64, H, W)          # fp64 oracle, SURVEY 8(d)'s stated tolerance
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(5)).to(hip)
    rv_p = {k: v[perm].contiguous() for k, v in rv.items()}
    gp = util.run_product(rs, rv_p, dL)
    assert np.array_equal(gp["radii"], got["radii"][perm.cpu().numpy()])
    # equal depths are ordered by index, so ties may swap under a permutation: compare at tolerance, not bit for bit
    assert util.close_frac(gp["color"], got["color"], 1e-4, 1e-5) > 0.999 and util.psnr(gp["color"], got["color"]) > 60
    for k in ("means3D", "shs", "opacities"):
        a, b = gp["grads"][k].astype(np.float64), g[k][perm.cpu().numpy()].astype(np.float64)
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-3, k


@pytest.mark.parametrize("seed", range(40))
def test_random_scene_matches_oracle_on_gpu(hip, oracle32, oracle64, seed):
    from tests.test_randomized import _draw
    rs, rv = _draw(5000 + seed, hip)
    pc.check_forward(rs, rv, oracle32)
    pc.check_backward(rs, rv, oracle64, oracle32=oracle32)          # the stated 0.995 / 1e-3 bar; the fp32 escape hatch is tallied (conftest)
    if seed % 3 == 0 and "colors_precomp" in rv and "cov3D_precomp" not in rv:
        # the single-pass RGB-D render (the path every fused mapping iteration takes) and its backward with a depth gradient, same tiered rule
        pc.check_fused_rgbd(rs, rv, oracle64, seed=seed, oracle32=oracle32)


@pytest.mark.parametrize("seed,plain", __import__("tests.fuzz_scenes", fromlist=["FLAGGED_RGBD"]).FLAGGED_RGBD)
def test_flagged_sweep_scene_fused_rgbd(hip, oracle32, oracle64, seed, plain):
    """The eleven scenes the round-4 sweeps flagged (profiles/r04_fuzz5.txt), pinned by seed: forward vs the fp32 oracle, the fused RGB-D render and
    its backward vs the fp64 oracle under the tiered rule of DESIGN section 6.  Seed 120013 is the alpha = 1/255 threshold scene
    (profiles/README.md, "seed 120013"): Gaussian 14305 owns pixel (71,18) at 255 alpha - 1 = -4.1e-6, the fp32 oracle sits one ulp below 1/255 and
    skips it, the device's FMA / v_exp_f32 form keeps it; with that Gaussian's opacity nudged by +-1e-4 device and fp64 oracle agree to 8.4e-5."""
    from activesplat_amd import _lib
    from tests.fuzz_scenes import sweep_scene
    lib = _lib.get()
    rs, rv = sweep_scene(seed, hip, plain)
    try:
        if plain:
            _lib.check(lib.gs_set_half_quadrants(0)); _lib.check(lib.gs_set_backward_chain(3, 0))
        pc.check_forward(rs, rv, oracle32)
        before = pc.HATCH["decisions"]
        pc.check_fused_rgbd(rs, rv, oracle64, seed=seed, oracle32=oracle32)
        if seed == 120013:
            w = pc.HATCH["decision_where"][before:]
            # (rounds 4-5: Gaussian 14305's alpha = 1/255 decision sent this scene through the decision-matched tier.  With round 6's factorised
            # 2-D covariance the alpha may sit off the threshold: IF the tier decides, it is for that Gaussian and nothing else)
            assert all(14305 in [i for i, _, _, _ in proven] for _, proven, _, _ in w), w
    finally:
        _lib.check(lib.gs_set_half_quadrants(256)); _lib.check(lib.gs_set_backward_chain(3, -1))


@pytest.mark.parametrize("seed,plain", __import__("tests.fuzz_scenes", fromlist=["FLAGGED_R05_RGBD"]).FLAGGED_R05_RGBD +
                         __import__("tests.fuzz_scenes", fromlist=["FLAGGED_R06_RGBD"]).FLAGGED_R06_RGBD)
def test_flagged_round5_sweep_scene_fused_rgbd(hip, oracle32, oracle64, seed, plain):
    from activesplat_amd import _lib
    from tests.fuzz_scenes import sweep_scene
    lib = _lib.get()
    rs, rv = sweep_scene(seed, hip, plain)
    try:
        if plain:
            _lib.check(lib.gs_set_half_quadrants(0)); _lib.check(lib.gs_set_backward_chain(3, 0))
        pc.check_forward(rs, rv, oracle32)
        for _ in range(2):                                           # (twice: the second pass runs on the optimistic launch's capacities)
            pc.check_fused_rgbd(rs, rv, oracle64, seed=seed, oracle32=oracle32)
    finally:
        _lib.check(lib.gs_set_half_quadrants(256)); _lib.check(lib.gs_set_backward_chain(3, -1))


@pytest.mark.parametrize("seed,s", __import__("tests.fuzz_scenes", fromlist=["FLAGGED_R06_HARD"]).FLAGGED_R06_HARD)
def test_flagged_round6_strongly_anisotropic_scene(hip, oracle32, oracle64, seed, s):
    """Three scenes of the s = 1.2 sweep (needles of up to 240 : 1, radius up to 3973 px, in front of the near plane) that round 5 left flagged: forward against
    the fp32 oracle (or, where fp32 itself cannot hold the tolerance, no further from fp64 than the fp32 oracle is), gradients under the tiered rule."""
    from tests.fuzz_scenes import hard_scene
    rs, rv = hard_scene(seed, hip, s)
    pc.check_forward(rs, rv, oracle32, oracle64=oracle64)
    pc.check_backward(rs, rv, oracle64, oracle32=oracle32)


@pytest.mark.parametrize("seed,plain", __import__("tests.fuzz_scenes", fromlist=["FLAGGED_R05_BACKWARD"]).FLAGGED_R05_BACKWARD)
def test_flagged_round5_sweep_scene_backward(hip, oracle32, oracle64, seed, plain):
    """Seed 140658: 10 of 1024 rotation elements outside the element-wise tolerance at a relative L2 of 4.7e-5 -- ONE alpha = 1/255 decision (Gaussian 112,
    pixel (0,39), 255 alpha - 1 = -5.9e-7) that the kernel takes the other way; it moves that Gaussian's row and, by alpha, every Gaussian that blends
    at the pixel.  Passes through the decision-matched tier, for Gaussian 112 and nothing else."""
    from tests.fuzz_scenes import sweep_scene
    rs, rv = sweep_scene(seed, hip, plain)
    pc.check_forward(rs, rv, oracle32)
    before = pc.HATCH["decisions"]
    pc.check_backward(rs, rv, oracle64, oracle32=oracle32)
    where = pc.HATCH["decision_where"][before:]
    if seed == 140658:
        assert all([i for i, _, _, _ in combo] == [112] for _, combo, _, _ in where), where      # (if the tier decides at all under round 6's arithmetic)
    else:
        assert not where                                           # (160050: inside the stated bar since the conic-gradient block runs in fp64)


def test_fused_rgbd_at_configs1_size(hip, oracle32, oracle64):
    """BASELINE configs[1]'s frame (500 k Gaussians, 640 x 480) through the single-pass RGB-D render: colour bit-equal to the plain render, depth /
    silhouette / depth^2 equal to the second reference-style pass, gradients (colour AND depth gradient) vs the two-pass formulation and vs the
    fp64 oracle at the stated tolerance."""
    rs, rv = util.scene(500_000, 640, 480, seed=0, device=hip)
    oracle64.set_threads(16); oracle32.set_threads(16)
    try:
        pc.check_fused_rgbd(rs, rv, oracle64, seed=3, oracle32=oracle32)
    finally:
        oracle64.set_threads(1); oracle32.set_threads(1)


def test_setup_camera_matches_reference_on_gpu(hip, oracle32):
    """a1 with a device hop (recon_helpers.py:4-28): setup_camera(device='cuda') -- host pose -> ONE packed 40-float block on the device, and a
    device-resident pose -> matrices built on the device -- against the reference's golden matrices, and a render through each of them against the
    render that is handed the golden matrices themselves."""
    import os
    from activesplat_amd import setup_camera
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "camera.npz"), allow_pickle=False)
    for i in range(3):
        W, H = (int(v) for v in d[f"c{i}_WH"])
        near, far = (float(v) for v in d[f"c{i}_nearfar"])
        cams = [setup_camera(W, H, d[f"c{i}_K"], d[f"c{i}_w2c"], near, far, scale_modifier=float(d[f"c{i}_mod"]), device=hip),
                setup_camera(W, H, d[f"c{i}_K"], torch.tensor(d[f"c{i}_w2c"], dtype=torch.float32, device=hip), near, far,
                             scale_modifier=float(d[f"c{i}_mod"]), device=hip)]
        for j, cam in enumerate(cams):
            for t in (cam.viewmatrix, cam.projmatrix, cam.campos, cam.bg):
                assert t.is_cuda and t.is_contiguous() and t.data_ptr() % 16 == 0
            assert (cam.image_width, cam.image_height) == (W, H)
            # (the device-side 4x4 inverse / product of the second form round differently from the host's: 1e-6 there, 1e-7 for the packed host block)
            np.testing.assert_allclose(cam.viewmatrix.cpu().numpy(), d[f"c{i}_view"], rtol=0, atol=1e-7)
            np.testing.assert_allclose(cam.projmatrix.cpu().numpy(), d[f"c{i}_proj"], rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose([cam.tanfovx, cam.tanfovy], d[f"c{i}_tanfov"], rtol=1e-6)
            np.testing.assert_allclose(cam.campos.cpu().numpy(), d[f"c{i}_campos"], atol=1e-5)
            np.testing.assert_array_equal(cam.bg.cpu().numpy(), d[f"c{i}_bg"])
        # render through the packed block == render through the golden matrices handed over directly (bit for bit: the same fp32 values)
        golden = cams[0]._replace(viewmatrix=torch.tensor(d[f"c{i}_view"], device=hip), projmatrix=torch.tensor(d[f"c{i}_proj"], device=hip),
                                  campos=torch.tensor(d[f"c{i}_campos"], device=hip), bg=torch.tensor(d[f"c{i}_bg"], device=hip), debug=True)
        if not (np.array_equal(cams[0].viewmatrix.cpu().numpy(), d[f"c{i}_view"]) and np.array_equal(cams[0].projmatrix.cpu().numpy(), d[f"c{i}_proj"])
                and np.array_equal(cams[0].campos.cpu().numpy(), d[f"c{i}_campos"])):
            continue                                            # (campos comes out of a 4x4 inverse: equal to 1e-5, not always to the bit)
        _, rv = util.scene(3000, W, H, seed=i, device=hip)
        c2w = np.linalg.inv(d[f"c{i}_w2c"])
        rv["means3D"] = (rv["means3D"].cpu().double() @ torch.tensor(c2w[:3, :3]).T + torch.tensor(c2w[:3, 3])).float().to(hip)   # in front of THIS camera
        a = util.run_product(cams[0]._replace(debug=True), rv)
        b = util.run_product(golden, rv)
        assert a["D"] == b["D"] > 0
        for k in ("color", "depth", "opacity", "radii"):
            assert np.array_equal(a[k], b[k]), (i, k)
        pc.check_forward(cams[0]._replace(debug=True), rv, oracle32)


def test_segmented_forward(hip, oracle32):
    pc.check_segmented_forward(hip, oracle32)



def test_whole_quadrants_on_small_images(hip, oracle32, oracle64):
    pc.check_whole_quadrants_on_small_images(hip, oracle32, oracle64)


def test_few_tile_backward_segments(hip, oracle64, oracle32):
    pc.check_few_tile_backward_segments(hip, oracle64, oracle32, N=30000, W=128, H=96)
    pc.check_few_tile_backward_segments(hip, oracle64, oracle32, N=120000, W=256, H=192, seed=43)


def test_two_segment_backward_at_the_reference_resolution(hip):
    """256 x 256 (the reference's default frames: 256 tiles), 150 k Gaussians, opaque enough that pixels stop at different depths: the
    backward that walks every quadrant's list in two segments (default for such images) against the one-walker kernel; forward outputs
    identical, gradients equal up to the order of the sums."""
    from activesplat_amd import _lib
    lib = _lib.get()
    rs, rv = util.scene(150_000, 256, 256, seed=21, device=hip)
    rs = rs._replace(debug=False)
    rv["opacities"] = (rv["opacities"] * 0.5 + 0.3).clamp(0, 0.99)
    dL = torch.randn(3, 256, 256, generator=torch.Generator().manual_seed(5))
    try:
        _lib.check(lib.gs_set_half_quadrants(0))
        ref = util.run_product(rs, rv, dL)
        _lib.check(lib.gs_set_half_quadrants(256))
        got = util.run_product(rs, rv, dL)
    finally:
        _lib.check(lib.gs_set_half_quadrants(256))
    for k in ("color", "depth", "opacity", "radii"):
        assert np.array_equal(got[k], ref[k]), k
    for k, g in got["grads"].items():
        r = ref["grads"][k]
        assert np.isfinite(g).all(), k
        assert np.linalg.norm(g.astype(np.float64) - r) <= 2e-5 * max(np.linalg.norm(r), 1e-30), (k, np.linalg.norm(g - r) / np.linalg.norm(r))


@pytest.mark.parametrize("seed,N,W,H", [(33, 5000, 288, 272), (34, 9000, 336, 256), (35, 3000, 272, 272), (36, 20000, 400, 304)])
def test_chained_backward_small(hip, oracle64, oracle32, seed, N, W, H):
    pc.check_chained_backward(hip, oracle64, N=N, W=W, H=H, oracle32=oracle32, seed=seed)


def test_chained_backward_fails_safe(hip):
    pc.check_chained_backward_fails_safe(hip, N=20000, W=400, H=304, seed=36)


@pytest.mark.parametrize("tickets", [0, 1], ids=["index-order", "ordered-tickets"])
def test_chained_backward_on_two_streams_concurrently(hip, tickets):
    """the chained backward at 640 x 480 (1200 tiles: the default there) on two HIP streams at once, 100 iterations -- each
    stream's gradients finite and equal to its serial run to 2e-5, the status word silent.  Two launches share the chip's walker slots, so a
    piece's predecessor may be queued behind the other launch's workgroups: the case ordered tickets exist for."""
    from activesplat_amd import _lib, GaussianRasterizer
    _lib.poll_async_status()
    _lib.check(_lib.get().gs_set_backward_chain_tickets(tickets))
    scenes = []
    for seed in (0, 1):
        rs, rv = util.scene(150_000, 640, 480, seed=seed, device=hip)
        rs = rs._replace(debug=False)
        dL = torch.randn(3, 480, 640, generator=torch.Generator().manual_seed(5 + seed)).to(hip)
        scenes.append((rs, rv, dL))

    def fwd_bwd(rs, rv, dL):
        inp = {k: v.detach().clone().requires_grad_(True) for k, v in rv.items()}
        m2d = torch.zeros(rv["means3D"].shape[0], 3, device=hip, requires_grad=True)
        color = GaussianRasterizer(raster_settings=rs)(means2D=m2d, **inp)[0]
        color.backward(dL)
        return {k: v.grad for k, v in inp.items()}
    serial = [fwd_bwd(*s) for s in scenes]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=hip) for _ in scenes]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    worst = 0.0
    for it in range(100):
        got = []
        for s, sc in zip(streams, scenes):
            with torch.cuda.stream(s):
                got.append(fwd_bwd(*sc))
        if it % 10 == 9 or it == 0:
            torch.cuda.synchronize()
            for g, r in zip(got, serial):
                for k in g:
                    assert torch.isfinite(g[k]).all(), (it, k)
                    rel = float((g[k].double() - r[k].double()).norm() / r[k].double().norm().clamp_min(1e-30))
                    worst = max(worst, rel)
                    assert rel <= 2e-5, (it, k, rel)
    torch.cuda.synchronize()
    _lib.check(_lib.get().gs_set_backward_chain_tickets(0))
    _lib.poll_async_status()                                      # silent: no wait ran out
    print(f"two concurrent chained backwards, 100 iterations: worst relative difference to the serial run {worst:.2e}")


def test_chained_backward_at_full_size_equals_one_walker_per_quadrant(hip):
    """BASELINE configs[1]'s frame (640 x 480: 1200 tiles, the chained backward is the default there): three pieces per quadrant against
    one walker per quadrant -- forward identical, gradients equal up to the order of the atomic sums."""
    from activesplat_amd import _lib
    lib = _lib.get()
    rs, rv = util.scene(500_000, 640, 480, seed=0, device=hip)
    rs = rs._replace(debug=False)
    dL = torch.randn(3, 480, 640, generator=torch.Generator().manual_seed(5))
    try:
        _lib.check(lib.gs_set_backward_chain(1, -1))
        ref = util.run_product(rs, rv, dL)
        _lib.check(lib.gs_set_backward_chain(3, -1))
        got = util.run_product(rs, rv, dL)
    finally:
        _lib.check(lib.gs_set_backward_chain(3, -1))
    for k in ("color", "depth", "opacity", "radii"):
        assert np.array_equal(got[k], ref[k]), k
    for k, g in got["grads"].items():
        r = ref["grads"][k]
        assert np.isfinite(g).all(), k
        assert np.linalg.norm(g.astype(np.float64) - r) <= 2e-5 * max(np.linalg.norm(r), 1e-30), (k, np.linalg.norm(g - r) / np.linalg.norm(r))


def test_adam_inside_the_backward_equals_backward_plus_step(hip):
    pc.check_adam_inside_the_backward(hip, n=30000, W=160, H=128, exact=False)


def test_adam_inside_the_backward_is_applied_once_and_never_silently(hip):
    pc.check_adam_backward_guards(hip, n=5000, W=96, H=64)


@pytest.mark.gpu
def test_unrendered_rows_of_dense_gradients_are_zero(hip):
    pc.check_unrendered_rows_of_dense_gradients(hip)


@pytest.mark.gpu
def test_unrendered_rows_of_a_keyframe_batch_are_written(hip):
    """the keyframe-batch gate of the per-Gaussian backward must still write dL/dmeans2D for wavefronts it skips."""
    pc.check_unrendered_rows_are_written(hip, n=8192, W=96, H=64)


def test_mapping_iteration_without_autograd_equals_the_autograd_path(hip):
    pc.check_mapping_iteration_without_autograd(hip, n=20000, exact=False)


def test_fused_loss_masks_nonfinite_depth_pixels(hip):
    pc.check_loss_masks_nonfinite_depth(hip, n=20000)


def test_raw_parameter_rasteriser_equals_the_activation_kernels(hip):
    pc.check_raw_parameter_mode(hip, n=20000)
    pc.check_raw_parameter_mode_sh(hip, n=20000, W=160, H=128)
    pc.check_raw_parameter_mode_nonfinite(hip, n=20000)


@pytest.mark.parametrize("seed", list(range(12)) + [135, 444, 604, 630, 812])
def test_loss_call_fully_fused_equals_the_reference_call_pattern_on_random_draws(hip, seed):
    """A dozen draws of the 1200-draw sweep of profiles/r05_fuzz_get_loss.txt (scripts/exp/fuzz_get_loss.py) plus five it flagged: get_loss as the reference runs it
    (torch activations, two raster passes, torch loss) against get_loss(fused=True, fused_loss=True, fused_preprocess=True).  135 / 812: sign(im - gt) of the reference's
    L1 colour term differs at one pixel-channel between the two renders; 444: the same for the depth term; 604: a depth tie; 630: one alpha = 1/255 decision.  The draw
    asserts that a difference above rounding is one of these classes (which one hangs on the last bit of the device's arithmetic and is not asserted)."""
    v = pc.check_get_loss_random_draw(seed, hip)
    assert v == "ok" or v[0] in ("L1 kink", "depth L1 kink", "depth tie", "rows")


@pytest.mark.parametrize("seed", list(range(24)) + [1782, 3452, 3570, 3988, 4385, 4668, 8395, 8572])
def test_raw_parameter_rasteriser_on_random_draws(hip, seed):
    """Two dozen draws of the 4000-scene sweep of profiles/r05_fuzz_raw.txt (scripts/exp/fuzz_raw.py), plus four it flagged: 1782 / 3452 (two overlapping splats
    one / two fp32 ulps apart in view depth, blended in either order by the two entries: 0.15 / 0.065 on their footprints) and 3570 / 3988 (one alpha = 1/255
    decision at one pixel moves one Gaussian's gradient row by 2-5 %) and two the round-6 soak flagged: 4385 / 4668 (a depth tie on 60 / 63 pixels AND one / two ordinary
    threshold flips of 4e-4 .. 2.4e-3 elsewhere: the residue outside the pair's footprints is held to the bound of a scene without a tie) and 8395 / 8572 (a tie on
    15 / 11 pixels -- inside the image bound -- that moves more than two gradient rows: 3.4e-4 / 4.5e-4 without the two worst).  The draw itself asserts; the classes are what the sweep recorded, not asserted here (they
    hang on the last bit of the device's arithmetic)."""
    assert pc.check_raw_entry_random_draw(seed, hip) in ("ok", "depth tie") or seed in (3570, 3988)
