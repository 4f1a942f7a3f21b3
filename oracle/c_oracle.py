"""
TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/raster_oracle.c header).

ctypes binding + build recipe for the plain-C CPU oracle.  Used by tests/, smoke() and bench.py's
`cpu_baseline` leg only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "raster_oracle.c")
LIB = os.path.join(HERE, "libraster_oracle.so")

_lib = None


def build(force: bool = False) -> str:
    deps = [SRC, os.path.join(HERE, "..", "include", "t4d_config.h")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        cmd = ["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", SRC,
               "-o", LIB, "-lm"]
        try:
            subprocess.check_call(cmd)
        except subprocess.CalledProcessError:
            # -march=native of the build host may not exist on the run host: retry portable
            cmd.remove("-march=native")
            subprocess.check_call(cmd)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        try:
            _lib = C.CDLL(LIB)
        except OSError:
            build(force=True)
            _lib = C.CDLL(LIB)
        _lib.orc_forward.restype = C.c_void_p
        _lib.orc_num_rendered.restype = C.c_uint64
        for n in ("orc_xy", "orc_depth", "orc_conic_opacity", "orc_rgb", "orc_cov3D", "orc_final_T"):
            getattr(_lib, n).restype = C.POINTER(C.c_float)
            getattr(_lib, n).argtypes = [C.c_void_p]
        for n in ("orc_point_list", "orc_ranges", "orc_n_contrib", "orc_tiles_touched"):
            getattr(_lib, n).restype = C.POINTER(C.c_uint32)
            getattr(_lib, n).argtypes = [C.c_void_p]
        _lib.orc_free.argtypes = [C.c_void_p]
        _lib.orc_num_rendered.argtypes = [C.c_void_p]
    return _lib


def _f(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


class OracleRender:
    """One forward render of one view; keeps the C state alive for backward()/introspection."""

    def __init__(self, view, means3D, opacities, scales=None, rotations=None, colors_precomp=None, shs=None,
                 cov3D_precomp=None):
        L = lib()
        self.L = L
        H, W = int(view.image_height), int(view.image_width)
        self.H, self.W = H, W
        to_np = lambda t: None if t is None else np.ascontiguousarray(
            t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t), dtype=np.float32)
        self.means3D, m_p = _f(to_np(means3D))
        self.P = P = self.means3D.shape[0]
        self.op, op_p = _f(to_np(opacities).reshape(-1))
        self.sc, sc_p = _f(to_np(scales))
        self.ro, ro_p = _f(to_np(rotations))
        self.cp, cp_p = _f(to_np(colors_precomp))
        self.sh, sh_p = _f(to_np(shs))
        self.cv, cv_p = _f(to_np(cov3D_precomp))
        self.M = 0 if self.sh is None else self.sh.shape[1]
        self.deg = int(view.sh_degree)
        vm, vm_p = _f(to_np(view.viewmatrix).reshape(16))
        pm, pm_p = _f(to_np(view.projmatrix).reshape(16))
        cam, cam_p = _f(to_np(view.campos).reshape(3))
        bg, bg_p = _f(to_np(view.bg).reshape(3))
        self.color = np.zeros((3, H, W), np.float32)
        self.depth = np.zeros((1, H, W), np.float32)
        self.alpha = np.zeros((1, H, W), np.float32)
        self.radii = np.zeros(P, np.int32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        self.h = L.orc_forward(
            C.c_int(P), C.c_int(H), C.c_int(W), C.c_int(self.deg), C.c_int(self.M),
            C.c_float(view.tanfovx), C.c_float(view.tanfovy), C.c_float(view.scale_modifier),
            bg_p, vm_p, pm_p, cam_p, m_p, sh_p, cp_p, op_p, sc_p, ro_p, cv_p, C.c_int(0),
            fp(self.color), fp(self.depth), fp(self.alpha), self.radii.ctypes.data_as(C.POINTER(C.c_int)))
        self._ptrs = (m_p, sh_p, sc_p, ro_p, cv_p)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.orc_free(C.c_void_p(self.h))
                self.h = None
        except Exception:
            pass

    # ---- introspection -------------------------------------------------------------------------------------
    @property
    def num_rendered(self):
        return int(self.L.orc_num_rendered(C.c_void_p(self.h)))

    def _arr(self, name, n, dtype):
        p = getattr(self.L, name)(C.c_void_p(self.h))
        return np.ctypeslib.as_array(p, shape=(n,)).astype(dtype).copy()

    def state(self):
        P, H, W = self.P, self.H, self.W
        gx, gy = (W + 15) // 16, (H + 15) // 16
        R = self.num_rendered
        return dict(
            xy=self._arr("orc_xy", 2 * P, np.float32).reshape(P, 2),
            depth=self._arr("orc_depth", P, np.float32),
            conic_opacity=self._arr("orc_conic_opacity", 4 * P, np.float32).reshape(P, 4),
            rgb=self._arr("orc_rgb", 3 * P, np.float32).reshape(P, 3),
            cov3D=self._arr("orc_cov3D", 6 * P, np.float32).reshape(P, 6),
            tiles_touched=self._arr("orc_tiles_touched", P, np.uint32),
            point_list=self._arr("orc_point_list", max(R, 1), np.uint32)[:R],
            ranges=self._arr("orc_ranges", 2 * gx * gy, np.uint32).reshape(gx * gy, 2),
            final_T=self._arr("orc_final_T", H * W, np.float32).reshape(H, W),
            n_contrib=self._arr("orc_n_contrib", H * W, np.uint32).reshape(H, W),
        )

    # ---- backward ------------------------------------------------------------------------------------------
    def backward(self, dL_dcolor, dL_ddepth=None, dL_dalpha=None):
        P, M = self.P, self.M
        to_np = lambda t: None if t is None else np.ascontiguousarray(
            t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t), dtype=np.float32)
        dc, dc_p = _f(to_np(dL_dcolor))
        dd, dd_p = _f(to_np(dL_ddepth))
        da, da_p = _f(to_np(dL_dalpha))
        g = dict(
            means3D=np.zeros((P, 3), np.float32), means2D=np.zeros((P, 3), np.float32),
            colors_precomp=np.zeros((P, 3), np.float32), shs=np.zeros((P, max(M, 1), 3), np.float32),
            opacities=np.zeros((P, 1), np.float32), scales=np.zeros((P, 3), np.float32),
            rotations=np.zeros((P, 4), np.float32), cov3D_precomp=np.zeros((P, 6), np.float32))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        m_p, sh_p, sc_p, ro_p, cv_p = self._ptrs
        self.L.orc_backward(C.c_void_p(self.h), m_p, sh_p, sc_p, ro_p, cv_p, dc_p, dd_p, da_p,
                            fp(g["means3D"]), fp(g["means2D"]), fp(g["colors_precomp"]), fp(g["shs"]),
                            fp(g["opacities"]), fp(g["scales"]), fp(g["rotations"]), fp(g["cov3D_precomp"]))
        if M == 0:
            g["shs"] = g["shs"][:, :0]
        return g


def mark_visible(means3D, viewmatrix):
    L = lib()
    m, m_p = _f(means3D)
    v, v_p = _f(np.asarray(viewmatrix, np.float32).reshape(16))
    out = np.zeros(m.shape[0], np.uint8)
    L.orc_mark_visible(C.c_int(m.shape[0]), m_p, v_p, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.astype(bool)
