"""
TEST INFRASTRUCTURE.  ctypes access to (a) the REAL reference `_render_colors_core` built by oracle/build_ref.py
into oracle/_ref/ and (b) the plain-C restatement oracle/texture_oracle.c.  Used by tests/ and tools/bench_bake.py
(as the CPU baseline, kind "reference") only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "libmesh_core_ref.so")
REF_SYMBOL = "_Z19_render_colors_corePfS_PiS_S_iiiii"        # void _render_colors_core(float*,float*,int*,float*,float*,int,int,int,int,int)
PORT_SRC = os.path.join(HERE, "texture_oracle.c")
PORT_SO = os.path.join(HERE, "libtexture_oracle.so")


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def build_port(force=False):
    if force or not os.path.exists(PORT_SO) or os.path.getmtime(PORT_SRC) > os.path.getmtime(PORT_SO):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", PORT_SRC, "-o", PORT_SO, "-lm"])
    return PORT_SO


def _prep(vertices, triangles, colors, h, w, c, BG):
    v = np.ascontiguousarray(vertices, np.float32).copy()
    t = np.ascontiguousarray(triangles, np.int32).copy()
    col = np.ascontiguousarray(colors, np.float32).copy()
    img = np.zeros((h, w, c), np.float32) if BG is None else np.ascontiguousarray(BG, np.float32).copy()
    depth = np.zeros((h, w), np.float32) - 999999.0                    # face3d/mesh/render.py:72
    return v, t, col, img, depth


def _call(fn, v, t, col, img, depth, h, w, c):
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    fn(fp(img), fp(v), t.ctypes.data_as(C.POINTER(C.c_int)), fp(col), fp(depth), C.c_int(v.shape[0]), C.c_int(t.shape[0]),
       C.c_int(h), C.c_int(w), C.c_int(c))
    return img


def render_colors_ref(vertices, triangles, colors, h, w, c=3, BG=None, return_depth=False):
    """The reference's own compiled code (argument order of mesh_core.h:63-69)."""
    lib = C.CDLL(REF_SO)
    fn = getattr(lib, REF_SYMBOL)
    fn.restype = None
    v, t, col, img, depth = _prep(vertices, triangles, colors, h, w, c, BG)
    _call(fn, v, t, col, img, depth, h, w, c)
    return (img, depth) if return_depth else img


def render_colors_port(vertices, triangles, colors, h, w, c=3, BG=None, return_depth=False):
    lib = C.CDLL(build_port())
    lib.tex_render_colors.restype = None
    v, t, col, img, depth = _prep(vertices, triangles, colors, h, w, c, BG)
    _call(lib.tex_render_colors, v, t, col, img, depth, h, w, c)
    return (img, depth) if return_depth else img


def render_colors_cpu(*a, **k):
    """Best available CPU answer: the real reference when its .so is present, else the port."""
    return render_colors_ref(*a, **k) if have_ref() else render_colors_port(*a, **k)
