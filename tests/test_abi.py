"""CPU: the C-ABI library loads and exports every symbol include/gsplat_hip.h declares; argument
errors surface as Python exceptions with the reference API's messages; no compute without a GPU."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "gsplat_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    from activesplat_amd import _lib
    assert _declared_symbols() == sorted(_lib.SYMBOLS)


def test_hip_library_loads_and_exports_every_symbol():
    import __graft_entry__ as ge
    from activesplat_amd import _lib
    ge.build()
    _lib.unload_for_tests()
    lib = _lib.get()
    for s in _declared_symbols():
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.gs_version()
    import ctypes as C
    gl = _lib.GsGeomLayout()
    assert lib.gs_geom_layout(1000, 640, 480, C.byref(gl)) == 0 and gl.total_bytes >= 1000 * (48 + 8 + 4 + 4)
    assert lib.gs_geom_layout(-1, 640, 480, C.byref(gl)) != 0 and b"bad argument" in lib.gs_last_error()


def test_no_cpu_fallback_and_argument_errors():
    from activesplat_amd import GaussianRasterizer, _lib
    from tests import util
    _lib.unload_for_tests()
    rs, rv = util.scene(10, 32, 32)
    m2d = torch.zeros(10, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GaussianRasterizer(raster_settings=rs)(means2D=m2d, **rv)
    r = GaussianRasterizer(raster_settings=rs)
    with pytest.raises(Exception, match="either SHs or precomputed colors"):
        r(means3D=rv["means3D"], means2D=m2d, opacities=rv["opacities"], scales=rv["scales"], rotations=rv["rotations"])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=rv["means3D"], means2D=m2d, opacities=rv["opacities"], colors_precomp=rv["colors_precomp"])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=rv["means3D"], means2D=m2d, opacities=rv["opacities"], colors_precomp=rv["colors_precomp"],
          scales=rv["scales"], rotations=rv["rotations"], cov3D_precomp=torch.zeros(10, 6))


def test_raw_parameter_entry_points_refuse_what_they_do_not_take():
    """render_rgbd_raw: exactly one of colours / SH rows, SH rows of 16 coefficients only, visibility tensors of the stated types; the
    knob gs_set_backward_chain refuses piece counts outside 1..3 (all checked before anything touches the device)."""
    import __graft_entry__ as ge
    from activesplat_amd import _lib, rasterizer as R
    from tests import util
    ge.build()
    _lib.unload_for_tests()
    lib = _lib.get()
    rs, rv = util.scene(10, 32, 32)
    m2d = torch.zeros(10, 3)
    raw = dict(means3D=rv["means3D"], means2D=m2d, logit_opacities=torch.zeros(10, 1), log_scales=torch.zeros(10, 3),
               unnorm_rotations=rv["rotations"], pose7=[1.0, 0, 0, 0, 0, 0, 0])
    with pytest.raises(Exception, match="either SHs or precomputed colors"):
        R.render_rgbd_raw(rs, **raw)
    with pytest.raises(Exception, match="16 coefficients"):
        R.render_rgbd_raw(rs, shs=torch.zeros(10, 9, 3), **raw)
    with pytest.raises(Exception, match="visibility"):
        R.render_rgbd_raw(rs, colors_precomp=rv["colors_precomp"], visibility=(torch.zeros(10, dtype=torch.float64), torch.zeros(10, dtype=torch.bool)), **raw)
    assert lib.gs_set_backward_chain(0, -1) != 0 and b"pieces out of range" in lib.gs_last_error()
    assert lib.gs_set_backward_chain(4, -1) != 0
    assert lib.gs_set_backward_chain(3, -1) == 0
    assert lib.gs_set_backward_segments(0) != 0 and lib.gs_set_backward_segments(4) != 0 and lib.gs_set_backward_segments(3) == 0


def test_dropin_module_name_and_settings_tuple():
    import diff_gaussian_rasterization as d
    from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    assert Camera._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                              "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    assert issubclass(Renderer, torch.nn.Module) and hasattr(d, "rasterize_gaussians")


def test_recorded_cut_positions_are_nearest_in_ratio():
    """cut_nearest chose the lower power of two for every target up to 2.83 x it (a shift by 17 where 16 was meant): the nearest recorded
    position in RATIO switches at sqrt(2) x.  gs_recorded_cut exposes the device function (the few-tile backward cuts its walks there)."""
    import ctypes as C
    from activesplat_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    lib.gs_recorded_cut.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]

    def cut(t):
        pos, lvl = C.c_uint32(), C.c_int32()
        assert lib.gs_recorded_cut(t, C.byref(pos), C.byref(lvl)) == 0
        return pos.value, lvl.value
    assert cut(100) == (0, -1) and cut(128) == (256, 0) and cut(1000) == (1024, 3) and cut(4096) == (4096, 15)
    assert cut(5000) == (4096, 15) and cut(5792) == (4096, 15) and cut(5794) == (8192, 16)          # 4096 sqrt 2 = 5792.6
    assert cut(6000) == (8192, 16) and cut(11585) == (8192, 16) and cut(11586) == (16384, 17)
    assert cut(12000) == (16384, 17) and cut(24000) == (32768, 18) and cut(10 ** 9) == (131072, 20)
    # three walkers of a 30 000-deep list: cuts at 8192 and 16384 (round 4: 4096 and 8192 -- the last walker took 73 % of the walk)
    assert cut(30000 // 3)[0] == 8192 and cut(2 * 30000 // 3)[0] == 16384


def test_frontend_builds_loads_and_refuses_host_tensors():
    """The C++ autograd front-end of the drop-in call (csrc/torch_frontend.cpp): builds with g++ against this interpreter's torch and the C ABI library,
    loads without a GPU, reports the ABI version it was linked against, and -- like the Python twin -- has no CPU fallback."""
    import pytest
    import torch
    from activesplat_amd import _frontend, _lib
    _frontend.build()
    ext = _frontend.get()
    assert int(ext.abi_version()) == _lib.ABI_VERSION
    z = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ext.rasterize(z, None, None, z, torch.zeros(4, 1), z, torch.zeros(4, 4), None, torch.zeros(3), torch.eye(4), torch.eye(4), torch.zeros(3),
                      32, 32, 1.0, 1.0, 1.0, 0, False, 0, 0, False)


def test_frontend_first_use_from_several_processes_and_the_fallback():
    """concurrent first use (the ranks of a torchrun launch) must never load a half-written module -- builders queue on a file lock, the
    compiler writes a private temporary file and the library appears atomically; a process that arrives while another builds waits and finds it
    up to date.  Where the build is impossible (no g++ on the box) the drop-in call falls back to the Python twin after ONE warning."""
    import subprocess
    import sys
    import warnings
    from activesplat_amd import _frontend
    _frontend.build()
    assert not _frontend._stale()
    # three processes, all told the module is stale at the same moment: one compiles, the others find it fresh once they hold the lock
    code = ("import os, sys, time; sys.path.insert(0, %r); from activesplat_amd import _frontend as F; "
            "os.utime(F.SO, (1, 1)) if sys.argv[1] == '0' else time.sleep(1.0); t = time.time(); F.build(); "
            "print('built' if os.path.getmtime(F.SO) > 1 else 'stale', round(time.time() - t, 1)); F.get()") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(i)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(3)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    assert all(o[0].startswith("built") for o in outs), outs
    assert not _frontend._stale() and not [f for f in os.listdir(os.path.dirname(_frontend.SO)) if f.endswith(".tmp")]
    # the fallback: a box where the build fails
    real = (_frontend.build, _frontend._stale, _frontend._mod, _frontend.unavailable)
    try:
        _frontend._mod, _frontend.unavailable = None, None
        _frontend._stale = lambda: True

        def no_compiler(force=False):
            raise FileNotFoundError("g++")
        _frontend.build = no_compiler
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert _frontend.get_or_none() is None and _frontend.get_or_none() is None
        assert len([x for x in w if "Python twin" in str(x.message)]) == 1
    finally:
        _frontend.build, _frontend._stale, _frontend._mod, _frontend.unavailable = real
