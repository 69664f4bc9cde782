"""ctypes front-end of the CPU oracle (oracle/gs_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of gs_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by activesplat_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False) -> None:
    """Compile libgso_f32.so / libgso_f64.so with gcc (seconds)."""
    libs = [os.path.join(_HERE, n) for n in ("libgso_f32.so", "libgso_f64.so")]
    src = os.path.join(_HERE, "gs_oracle.c")
    if not force and all(os.path.exists(p) and os.path.getmtime(p) >= os.path.getmtime(src) for p in libs):
        return
    subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)


class _Cam32(C.Structure):
    _fields_ = [("P", C.c_int32), ("W", C.c_int32), ("H", C.c_int32), ("sh_degree", C.c_int32),
                ("sh_coeffs", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("scale_modifier", C.c_float), ("bg", C.c_float * 3), ("viewmatrix", C.c_float * 16),
                ("projmatrix", C.c_float * 16), ("campos", C.c_float * 3)]


class _Cam64(C.Structure):
    _fields_ = [("P", C.c_int32), ("W", C.c_int32), ("H", C.c_int32), ("sh_degree", C.c_int32),
                ("sh_coeffs", C.c_int32), ("tanfovx", C.c_double), ("tanfovy", C.c_double),
                ("scale_modifier", C.c_double), ("bg", C.c_double * 3), ("viewmatrix", C.c_double * 16),
                ("projmatrix", C.c_double * 16), ("campos", C.c_double * 3)]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """precision: 'f32' (bit-level spec incl. integer artefacts) or 'f64' (gradient oracle)."""

    def __init__(self, precision: str = "f32"):
        build()
        assert precision in ("f32", "f64")
        self.real = np.float32 if precision == "f32" else np.float64
        self.lib = C.CDLL(os.path.join(_HERE, f"libgso_{precision}.so"))
        assert self.lib.gso_real_size() == np.dtype(self.real).itemsize
        self.lib.gso_preprocess.restype = C.c_int64
        self._Cam = _Cam32 if precision == "f32" else _Cam64

    # -- helpers -------------------------------------------------------------------------
    def set_threads(self, n: int) -> None:
        """Host threads of the pixel loops (default 1 = the deterministic sequential order the parity tests use)."""
        self.lib.gso_set_threads(int(n))

    def set_threshold_scale(self, per_gaussian=None) -> None:
        """TEST HOOK: per-Gaussian factor on the alpha >= 1/255 threshold (None: the published algorithm) -- see gso_set_threshold_scale."""
        self._thresh = None if per_gaussian is None else np.ascontiguousarray(np.asarray(per_gaussian, dtype=self.real))
        self.lib.gso_set_threshold_scale(_ptr(self._thresh))

    def _r(self, a):
        return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=self.real))

    def _cam(self, cam: dict, P: int, sh_coeffs: int):
        c = self._Cam()
        c.P, c.W, c.H = P, int(cam["W"]), int(cam["H"])
        c.sh_degree, c.sh_coeffs = int(cam.get("sh_degree", 0)), sh_coeffs
        c.tanfovx, c.tanfovy = float(cam["tanfovx"]), float(cam["tanfovy"])
        c.scale_modifier = float(cam.get("scale_modifier", 1.0))
        for i, v in enumerate(np.asarray(cam["bg"], dtype=np.float64).reshape(3)):
            c.bg[i] = v
        for i, v in enumerate(np.asarray(cam["viewmatrix"], dtype=np.float64).reshape(16)):
            c.viewmatrix[i] = v
        for i, v in enumerate(np.asarray(cam["projmatrix"], dtype=np.float64).reshape(16)):
            c.projmatrix[i] = v
        for i, v in enumerate(np.asarray(cam["campos"], dtype=np.float64).reshape(3)):
            c.campos[i] = v
        return c

    # -- forward --------------------------------------------------------------------------
    def forward(self, cam: dict, means3D, opacities, colors=None, shs=None, scales=None, rotations=None,
                cov3D_precomp=None) -> dict:
        r = self._r
        means3D = r(means3D); P = means3D.shape[0]
        opac = r(opacities).reshape(-1)
        colors, shs, scales, rots, cov3 = r(colors), r(shs), r(scales), r(rotations), r(cov3D_precomp)
        M = 0 if shs is None else shs.shape[1]
        c = self._cam(cam, P, M)
        W, H = c.W, c.H
        gx, gy = (W + 15) // 16, (H + 15) // 16
        o = dict(radii=np.zeros(P, np.int32), xy=np.zeros((P, 2), self.real), depth=np.zeros(P, self.real),
                 cov2d=np.zeros((P, 3), self.real), conic_opacity=np.zeros((P, 4), self.real),
                 rgb=np.zeros((P, 3), self.real), clamped=np.zeros((P, 3), np.uint8),
                 rect=np.zeros((P, 4), np.int32), tiles_touched=np.zeros(P, np.uint32),
                 offsets=np.zeros(P, np.uint32), cov3d=np.zeros((P, 6), self.real))
        D = self.lib.gso_preprocess(C.byref(c), _ptr(means3D), _ptr(shs), _ptr(colors), _ptr(opac), _ptr(scales),
                                    _ptr(rots), _ptr(cov3), _ptr(o["radii"]), _ptr(o["xy"]), _ptr(o["depth"]),
                                    _ptr(o["cov2d"]), _ptr(o["conic_opacity"]), _ptr(o["rgb"]), _ptr(o["clamped"]),
                                    _ptr(o["rect"]), _ptr(o["tiles_touched"]), _ptr(o["offsets"]), _ptr(o["cov3d"]))
        o["D"] = int(D)
        n = max(int(D), 1)
        o.update(keys_unsorted=np.zeros(n, np.uint64), ids_unsorted=np.zeros(n, np.uint32),
                 keys_sorted=np.zeros(n, np.uint64), ids_sorted=np.zeros(n, np.uint32),
                 ranges=np.zeros((gx * gy, 2), np.uint32))
        self.lib.gso_bin(C.byref(c), _ptr(o["depth"]), _ptr(o["rect"]), _ptr(o["offsets"]), C.c_int64(D),
                         _ptr(o["keys_unsorted"]), _ptr(o["ids_unsorted"]), _ptr(o["keys_sorted"]),
                         _ptr(o["ids_sorted"]), _ptr(o["ranges"]))
        for k in ("keys_unsorted", "ids_unsorted", "keys_sorted", "ids_sorted"):
            o[k] = o[k][:D]
        bg = r(cam["bg"]).reshape(3)
        o.update(color=np.zeros((3, H, W), self.real), out_depth=np.zeros((1, H, W), self.real),
                 opacity=np.zeros((1, H, W), self.real), final_T=np.zeros((H, W), self.real),
                 n_contrib=np.zeros((H, W), np.uint32), depth_sq=np.zeros((1, H, W), self.real))
        self.lib.gso_blend_forward(C.byref(c), 3, _ptr(bg), _ptr(o["ranges"]), _ptr(o["ids_sorted"]), _ptr(o["xy"]),
                                   _ptr(o["depth"]), _ptr(o["conic_opacity"]), _ptr(o["rgb"]), _ptr(o["color"]),
                                   _ptr(o["out_depth"]), _ptr(o["opacity"]), _ptr(o["final_T"]), _ptr(o["n_contrib"]),
                                   _ptr(o["depth_sq"]))
        o["_ctx"] = dict(c=c, means3D=means3D, shs=shs, scales=scales, rots=rots, bg=bg, colors=colors)
        return o

    # -- backward -------------------------------------------------------------------------
    def backward(self, fwd: dict, dL_dcolor, dL_ddepth=None) -> dict:
        """dL_ddepth [1,H,W] (optional): gradient w.r.t. the `out_depth` output (fused RGB-D path)."""
        ctx = fwd["_ctx"]; c = ctx["c"]; P = c.P
        dpix = self._r(dL_dcolor).reshape(3, c.H, c.W)
        ddep = None if dL_ddepth is None else self._r(dL_ddepth).reshape(c.H, c.W)
        dz = None if dL_ddepth is None else np.zeros(P, self.real)
        dxy = np.zeros((P, 2), self.real); dconic = np.zeros((P, 3), self.real)
        dop = np.zeros(P, self.real); dfeat = np.zeros((P, 3), self.real)
        self.lib.gso_blend_backward(C.byref(c), 3, _ptr(ctx["bg"]), _ptr(fwd["ranges"]), _ptr(fwd["ids_sorted"]),
                                    _ptr(fwd["xy"]), _ptr(fwd["conic_opacity"]), _ptr(fwd["rgb"]),
                                    _ptr(fwd["final_T"]), _ptr(fwd["n_contrib"]), _ptr(dpix), _ptr(dxy),
                                    _ptr(dconic), _ptr(dop), _ptr(dfeat), _ptr(fwd["depth"]), _ptr(ddep), _ptr(dz))
        M = c.sh_coeffs
        g = dict(means2D=np.zeros((P, 3), self.real), means3D=np.zeros((P, 3), self.real),
                 scales=np.zeros((P, 3), self.real), rotations=np.zeros((P, 4), self.real),
                 cov3D_precomp=np.zeros((P, 6), self.real), shs=np.zeros((P, max(M, 1), 3), self.real),
                 colors_precomp=np.zeros((P, 3), self.real))
        self.lib.gso_preprocess_backward(C.byref(c), _ptr(ctx["means3D"]), _ptr(ctx["shs"]), _ptr(ctx["scales"]),
                                         _ptr(ctx["rots"]), _ptr(fwd["radii"]), _ptr(fwd["cov3d"]),
                                         _ptr(fwd["clamped"]), _ptr(dxy), _ptr(dconic), _ptr(dfeat), _ptr(dz),
                                         _ptr(g["means2D"]), _ptr(g["means3D"]), _ptr(g["scales"]),
                                         _ptr(g["rotations"]), _ptr(g["cov3D_precomp"]), _ptr(g["shs"]),
                                         _ptr(g["colors_precomp"]))
        g["opacities"] = dop.reshape(P, 1)
        g.update(_dxy=dxy, _dconic=dconic, _dfeat=dfeat)
        return g

    def adam(self, p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-15):
        """In-place on copies; returns (p, m, v)."""
        p, m, v = (np.array(a, dtype=self.real, copy=True).reshape(-1) for a in (p, m, v))
        g = self._r(g).reshape(-1)
        rt = C.c_float if self.real == np.float32 else C.c_double
        self.lib.gso_adam(C.c_int64(p.size), _ptr(p), _ptr(g), _ptr(m), _ptr(v), rt(lr), rt(b1), rt(b2), rt(eps),
                          C.c_int32(step))
        return p, m, v
