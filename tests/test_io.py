"""params.npz writer / loader, GaussianPacket and the height-cut filter (activesplat_amd/io.py; reference
common_utils.py:25-44,61-68, splatam/__init__.py:555-572, gui_utils.py:76-88, visualizer.py:2277-2286)."""
import numpy as np
import torch

from activesplat_amd import io as IO
from activesplat_amd import synthetic as syn


def _run_state(frames_alloc=12, frames_done=5, n=300):
    p = syn.make_params(n, 64, 48, seed=1)
    params = {k: torch.nn.Parameter(v) for k, v in p.items()}
    params["cam_unnorm_rots"] = torch.nn.Parameter(torch.randn(1, 4, frames_alloc))
    params["cam_trans"] = torch.nn.Parameter(torch.randn(1, 3, frames_alloc))
    variables = {"timestep": torch.arange(n).float() % frames_done}
    gt = [torch.eye(4) * (i + 1) for i in range(frames_done)]
    return params, variables, gt


def test_params_npz_has_the_reference_keys_shapes_and_trimming(tmp_path):
    params, variables, gt = _run_state()
    out = IO.finalize_params(params, variables, torch.eye(3) * 2, torch.eye(4), 640, 480, gt, [0, 2, 4])
    path = IO.save_params(out, str(tmp_path / "run"))
    assert path.endswith("params.npz")
    z = np.load(path)
    assert set(z.files) == set(IO.FILE_KEYS)
    assert z["cam_trans"].shape == (1, 3, 5) and z["cam_unnorm_rots"].shape == (1, 4, 5)      # trimmed to mapped frames
    assert np.array_equal(z["cam_trans"], params["cam_trans"].detach().numpy()[..., :5])
    assert z["gt_w2c_all_frames"].shape == (5, 4, 4) and z["gt_w2c_all_frames"][3, 0, 0] == 4
    assert list(z["keyframe_time_indices"]) == [0, 2, 4] and int(z["org_width"]) == 640 and int(z["org_height"]) == 480
    assert z["means3D"].dtype == np.float32 and np.array_equal(z["means3D"], params["means3D"].detach().numpy())
    assert np.array_equal(z["timestep"], variables["timestep"].numpy())
    ck = IO.save_params_ckpt(out, str(tmp_path / "ck"), 7)
    assert ck.endswith("params7.npz") and set(np.load(ck).files) == set(IO.FILE_KEYS)
    loaded, extras = IO.load_params(path, "cpu")
    assert set(loaded) == set(IO.PARAM_KEYS) and all(isinstance(v, torch.nn.Parameter) for v in loaded.values())
    assert torch.equal(loaded["log_scales"], params["log_scales"]) and extras["w2c"].shape == (4, 4)


def test_gaussian_packet_flags():
    assert IO.GaussianPacket().has_gaussians is False and not hasattr(IO.GaussianPacket(), "params")
    pk = IO.GaussianPacket({"means3D": 1}, current_frame="c2w")
    assert pk.has_gaussians and pk.params == {"means3D": 1} and pk.current_frame == "c2w"


def test_height_cut_equals_boolean_mask_filter(emu):
    p = {k: v.to(emu) for k, v in syn.make_params(1000, 64, 48, seed=2).items()}
    ref = {k: v.clone() for k, v in p.items()}
    upper, lower = -0.3, 0.4
    cond = torch.logical_or(-ref["means3D"][:, 1] < upper, -ref["means3D"][:, 1] > lower)
    got = IO.cut_gaussian_by_height(p, upper, lower)
    assert 0 < got["means3D"].shape[0] < 1000
    for k in IO.GAUSSIAN_ROW_KEYS:
        assert torch.equal(got[k], ref[k][~cond]), k
