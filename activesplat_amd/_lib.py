"""ctypes binding of the C ABI in include/gsplat_hip.h (libgsplat_hip.so, hipcc-built for gfx950).

There is no CPU fallback: `get()` raises if the HIP library is missing.  `load_for_tests(path)` exists
only so that tests/ can point the same binding at the host-emulated build of the kernel sources
(tests/hipemu) while debugging kernel logic without a GPU; the product never calls it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsplat_hip.so")

_lib = None
_emulated = False


class GsCamera(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("sh_degree", C.c_int32),
                ("sh_coeffs", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("scale_modifier", C.c_float), ("num_views", C.c_int32), ("bg", C.c_void_p),
                ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p)]


class GsGeomLayout(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("total_bytes", "geom", "rect", "tiles_touched", "offsets", "block_sums", "clamped",
                                          "tile_total", "tile_base", "sh_jac", "depth_bits")]


class GsImageLayout(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("total_bytes", "ranges", "final_T", "n_contrib", "split_state")]


class GsAdamTensor(C.Structure):
    _fields_ = [("n", C.c_int64), ("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("step", C.c_int32),
                ("reserved", C.c_int32)]


class GsRowTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("grad", C.c_void_p),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("width", C.c_int32),
                ("step", C.c_int32)]


class GsBinLayout(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("total_bytes", "path", "pairs", "keys_unsorted", "vals_unsorted", "keys_sorted",
                                          "sort_temp", "segments", "seg_T", "pairs_alt")]


SORT_AUTO, SORT_TILE_LDS, SORT_RADIX = 0, 1, 2


# every symbol include/gsplat_hip.h declares (tests check the library exports all of them)
#: the GS_ABI_VERSION of include/gsplat_hip.h this binding was written against (checked when a library is bound)
ABI_VERSION = 10

SYMBOLS = ("gs_abi_version", "gs_geom_layout", "gs_image_layout", "gs_bin_layout", "gs_backward_scratch_bytes", "gs_last_error",
           "gs_version", "gs_set_sort_path", "gs_set_forward_segments", "gs_set_half_quadrants", "gs_set_backward_chain", "gs_set_backward_chain_tickets", "gs_set_backward_chain_polls", "gs_async_status_word", "gs_async_status_clear", "gs_recorded_cut", "gs_set_backward_segments", "gs_preprocess_forward", "gs_preprocess_forward_raw", "gs_render_forward", "gs_render_backward", "gs_render_backward_raw", "gs_render_backward_raw_adam", "gs_adam_step", "gs_adam_step_multi",
           "gs_profile_enable", "gs_profile_stage_count", "gs_profile_stage_name", "gs_profile_collect",
           "gs_compact_scratch_bytes", "gs_compact_index", "gs_gather_rows", "gs_mapping_loss_scratch_bytes", "gs_mapping_loss", "gs_activate_forward", "gs_activate_backward", "gs_activate_backward_accumulate",
           "gs_grow_scratch_bytes", "gs_grow_gaussians", "gs_keyframe_overlap", "gs_visibility_stats", "gs_accumulate_grad2d",
           "gs_gather_rows_zero_tail", "gs_densify_classify", "gs_densify_children", "gs_atlas_layout",
           "gs_pack_columns", "gs_adam_rows", "gs_unpack_columns", "gs_compact3_scratch_bytes", "gs_compact_index3")


def _bind(lib):
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    # a stale prebuilt library (another round's .so, an old emulated build) must fail HERE, not misread pointers later
    try:
        lib.gs_abi_version.restype = i32
        have = int(lib.gs_abi_version())
    except AttributeError:
        have = None
    if have != ABI_VERSION:
        raise RuntimeError(f"{getattr(lib, '_name', 'library')}: C ABI version {have}, this binding needs {ABI_VERSION} (include/gsplat_hip.h "
                           "GS_ABI_VERSION) -- rebuild it: python -c 'import __graft_entry__ as g; g.build()'")
    lib.gs_geom_layout.argtypes = [i32, i32, i32, C.POINTER(GsGeomLayout)]
    lib.gs_atlas_layout.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.gs_atlas_layout.restype = C.c_int
    lib.gs_set_sort_path.argtypes = [i32]
    lib.gs_set_sort_path.restype = C.c_int
    lib.gs_set_forward_segments.argtypes = [i32]
    lib.gs_set_half_quadrants.argtypes = [i32]
    lib.gs_set_half_quadrants.restype = C.c_int
    lib.gs_set_backward_chain.argtypes = [i32, i32]
    lib.gs_set_backward_chain.restype = C.c_int
    lib.gs_set_backward_segments.argtypes = [i32]
    lib.gs_set_backward_segments.restype = C.c_int
    lib.gs_set_backward_chain_tickets.argtypes = [i32]
    lib.gs_set_backward_chain_tickets.restype = C.c_int
    lib.gs_set_backward_chain_polls.argtypes = [i32]
    lib.gs_set_backward_chain_polls.restype = C.c_int
    lib.gs_async_status_clear.argtypes = []
    lib.gs_async_status_clear.restype = C.c_int
    lib.gs_async_status_word.argtypes = [C.POINTER(C.POINTER(C.c_uint32))]
    lib.gs_async_status_word.restype = C.c_int
    lib.gs_recorded_cut.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(i32)]
    lib.gs_recorded_cut.restype = C.c_int
    lib.gs_set_forward_segments.restype = C.c_int
    lib.gs_image_layout.argtypes = [i32, i32, C.POINTER(GsImageLayout)]
    lib.gs_bin_layout.argtypes = [i64, C.c_uint32, i32, i32, C.POINTER(GsBinLayout)]
    lib.gs_backward_scratch_bytes.argtypes = [i32]
    lib.gs_backward_scratch_bytes.restype = C.c_uint64
    lib.gs_last_error.restype = C.c_char_p
    lib.gs_version.restype = C.c_char_p
    lib.gs_preprocess_forward.argtypes = [C.POINTER(GsCamera), i32] + [vp] * 7 + [vp, vp, vp, vp, vp, i32, vp]
    lib.gs_render_forward.argtypes = [C.POINTER(GsCamera), i32, i64, C.c_uint32] + [vp] * 8 + [vp, vp]
    lib.gs_render_backward.argtypes = [C.POINTER(GsCamera), i32, i64] + [vp] * 6 + [vp] * 6 + [vp] * 8 + [vp, i32, i32, vp]
    # (cam, P, means3D, shs, colors, logit, log_scales, unnorm_rot, h_pose7, isotropic, max_2D_radius, seen, radii, geom, image, d_counts, h_counts, want_backward, stream)
    lib.gs_preprocess_forward_raw.argtypes = [C.POINTER(GsCamera), i32] + [vp] * 6 + [vp, i32] + [vp, vp] + [vp] * 5 + [i32, vp]
    # (cam, P, D, means3D, shs, colors, logit, log_scales, unnorm_rot, h_pose7, isotropic, accumulate, radii, geom, point_list, image,
    #  dL_dcolor, dL_ddepth, 7 gradient outputs, scratch, scratch_zeroed, have_sh_jacobian, stream)
    lib.gs_render_backward_raw.argtypes = [C.POINTER(GsCamera), i32, i64] + [vp] * 6 + [vp, i32, i32] + [vp] * 4 + [vp] * 2 + [vp] * 7 + [vp, i32, i32, vp]
    # (cam, P, D, means3D, shs, colors, logit, log_scales, unnorm_rot, h_pose7, isotropic, radii, geom, point_list, image, dL_dcolor, dL_ddepth,
    #  dL_dmeans2D, scratch, scratch_zeroed, have_sh_jacobian, adam5, stream)
    lib.gs_render_backward_raw_adam.argtypes = [C.POINTER(GsCamera), i32, i64] + [vp] * 6 + [vp, i32] + [vp] * 4 + [vp] * 2 + [vp] + [vp, i32, i32] + \
        [C.POINTER(GsAdamTensor), vp]
    lib.gs_render_backward_raw_adam.restype = C.c_int
    lib.gs_preprocess_forward_raw.restype = C.c_int
    lib.gs_render_backward_raw.restype = C.c_int
    lib.gs_adam_step.argtypes = [i64, vp, vp, vp, vp, C.c_double, C.c_double, C.c_double, C.c_double, i32, vp]
    lib.gs_adam_step_multi.argtypes = [i32, C.POINTER(GsAdamTensor), vp]
    lib.gs_profile_enable.argtypes = [i32]
    lib.gs_profile_stage_count.restype = i32
    lib.gs_profile_stage_name.argtypes = [i32]
    lib.gs_profile_stage_name.restype = C.c_char_p
    lib.gs_profile_collect.argtypes = [vp, vp, i32]
    lib.gs_profile_collect.restype = C.c_int
    lib.gs_activate_forward.argtypes = [i32, i32] + [vp] * 9 + [vp]
    lib.gs_activate_forward.restype = C.c_int
    lib.gs_activate_backward.argtypes = [i32, i32] + [vp] * 12 + [vp]
    lib.gs_activate_backward.restype = C.c_int
    lib.gs_activate_backward_accumulate.argtypes = [i32, i32] + [vp] * 12 + [vp]
    lib.gs_activate_backward_accumulate.restype = C.c_int
    lib.gs_mapping_loss_scratch_bytes.argtypes = [i32, i32]
    lib.gs_mapping_loss_scratch_bytes.restype = C.c_uint64
    lib.gs_mapping_loss.argtypes = [i32, i32, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, i64, vp]
    lib.gs_mapping_loss.restype = C.c_int
    lib.gs_compact_scratch_bytes.argtypes = [i64]
    lib.gs_compact_scratch_bytes.restype = C.c_uint64
    lib.gs_compact_index.argtypes = [i64, vp, vp, vp, vp, vp]
    lib.gs_compact_index.restype = C.c_int
    lib.gs_gather_rows.argtypes = [i64, i32, vp, vp, vp, vp]
    lib.gs_gather_rows.restype = C.c_int
    lib.gs_gather_rows_zero_tail.argtypes = [i64, i64, i32, vp, vp, vp, vp]
    lib.gs_gather_rows_zero_tail.restype = C.c_int
    lib.gs_densify_classify.argtypes = [i32, i32, vp, vp, vp, vp, vp, f32, f32, i32, i32, vp, vp, vp, vp, vp]
    lib.gs_densify_classify.restype = C.c_int
    lib.gs_densify_children.argtypes = [i32, i32, i32, vp, vp, C.c_uint64, vp, vp, vp]
    lib.gs_compact3_scratch_bytes.argtypes = [i64]
    lib.gs_compact3_scratch_bytes.restype = C.c_uint64
    lib.gs_compact_index3.argtypes = [i64, vp, vp, vp, i32, vp, vp, vp, vp]
    lib.gs_compact_index3.restype = C.c_int
    lib.gs_densify_children.restype = C.c_int
    lib.gs_grow_scratch_bytes.argtypes = [i32, i32]
    lib.gs_grow_scratch_bytes.restype = C.c_uint64
    lib.gs_grow_gaussians.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gs_grow_gaussians.restype = C.c_int
    lib.gs_keyframe_overlap.argtypes = [i32, vp, i32, vp, vp, i32, i32, i32, vp, vp]
    lib.gs_keyframe_overlap.restype = C.c_int
    lib.gs_visibility_stats.argtypes = [i32, vp, vp, vp, vp]
    lib.gs_visibility_stats.restype = C.c_int
    lib.gs_accumulate_grad2d.argtypes = [i32, vp, vp, vp, vp, vp]
    lib.gs_accumulate_grad2d.restype = C.c_int
    lib.gs_adam_step_multi.restype = C.c_int
    lib.gs_pack_columns.argtypes = [i32, C.POINTER(GsRowTensor), i64, i64, vp, vp]
    lib.gs_pack_columns.restype = C.c_int
    lib.gs_adam_rows.argtypes = [i32, C.POINTER(GsRowTensor), i64, i64, i64, vp, vp, vp]
    lib.gs_adam_rows.restype = C.c_int
    lib.gs_unpack_columns.argtypes = [i32, C.POINTER(GsRowTensor), i64, vp, vp]
    lib.gs_unpack_columns.restype = C.c_int
    for n in ("gs_geom_layout", "gs_image_layout", "gs_bin_layout", "gs_preprocess_forward", "gs_render_forward",
              "gs_render_backward", "gs_adam_step"):
        getattr(lib, n).restype = C.c_int
    return lib


def get():
    """The HIP library; raises loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). activesplat_amd has no CPU fallback.")
        _lib = _bind(C.CDLL(LIB_PATH))
    return _lib


def load_for_tests(path: str):
    """TEST HOOK ONLY: bind an alternative build of the same C ABI (the host-emulated kernels)."""
    global _lib, _emulated, _status
    _lib = _bind(C.CDLL(path))
    _emulated = True
    _status = None
    return _lib


def unload_for_tests():
    global _lib, _emulated, _status
    _lib = None
    _emulated = False
    _status = None


#: the library's host-mapped status word (gs_async_status_word), as a ctypes view: a plain host load per poll
_status = None


def poll_async_status():
    """Called in front of every render: has a chained backward walk of an EARLIER launch run out of its bounded wait (include/gsplat_hip.h,
    gs_set_backward_chain_polls)?  Then that backward's gradients hold NaNs: the word is cleared, chaining is switched off for the rest of the
    process (every quadrant walked by one wavefront again: slower at 640 x 480, no hand-over to wait for) and the caller is told."""
    global _status
    if _status is None:
        w = C.POINTER(C.c_uint32)()
        check(get().gs_async_status_word(C.byref(w)))
        _status = w
    if _status[0]:
        check(get().gs_set_backward_chain(1, -1))
        check(get().gs_async_status_clear())      # (synchronises the device, then clears the sticky device word: optimiser steps run again)
        _status[0] = 0
        raise RuntimeError("activesplat_amd: a chained backward walk timed out waiting for the piece in front of it -- the gradients of the previous "
                           "backward on this process are invalid (NaN); optimiser steps enqueued behind it were SKIPPED by their kernels (parameters and "
                           "moments untouched, step counters one ahead).  Chained walks are now off (gs_set_backward_chain(1, -1)); render that frame again.")


def emulated() -> bool:
    return _emulated


def check(rc: int):
    if rc != 0:
        raise Exception(get().gs_last_error().decode())


_raw_stream = None


def stream_handle(device) -> int:
    """The current HIP stream of `device` as an integer handle (0 for the host-emulated test build).  torch.cuda.current_stream()
    builds a Stream object per call (~6 us; a mapping iteration asks seven times): the raw accessor returns the same handle in a
    fraction of a microsecond.  Falls back to the public call if this torch build does not have it."""
    global _raw_stream
    if device.type != "cuda":
        return 0
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if _raw_stream:
        return int(_raw_stream(idx))
    return int(torch.cuda.current_stream(device).cuda_stream)


def stream_ptr(device):
    return C.c_void_p(stream_handle(device))


def profile_collect():
    """-> {stage_name: (total_ms, calls)} since gs_profile_enable(1); synchronises the recorded events."""
    lib = get()
    n = lib.gs_profile_stage_count()
    ms = (C.c_float * n)()
    calls = (C.c_int32 * n)()
    check(lib.gs_profile_collect(ms, calls, n))
    return {lib.gs_profile_stage_name(i).decode(): (float(ms[i]), int(calls[i])) for i in range(n)}
