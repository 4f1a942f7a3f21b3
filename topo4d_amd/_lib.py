"""
ctypes binding of the C-ABI shared library (include/topo4d_raster.h).

The library is built in-tree by `__graft_entry__.build()` / `python -m topo4d_amd.build` with
`hipcc --offload-arch=gfx950`.  There is NO fallback: if the shared object is missing or does not load, every
entry point of the product raises — a rasterizer that silently ran on the CPU would void every parity claim.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# T4D_LIB: load another build of the SAME library (an experiment build next to the shipped one, tools/ab_build.sh); there
# is still no fallback - a path that does not load raises.
LIB_PATH = os.environ.get("T4D_LIB") or os.path.join(HERE, "csrc", "libtopo4d_raster.so")

T4D_ABI_VERSION = 4
T4D_VIEW_FLOATS = 40
T4D_GRAD_PAIR_FLOATS = 10

T4D_OK, T4D_ERR_ARG, T4D_ERR_HIP, T4D_ERR_PAIR_OVERFLOW, T4D_ERR_STATE_SIZE = 0, 1, 2, 3, 4
T4D_FLAG_CHECKED, T4D_FLAG_DEBUG_SYNC, T4D_FLAG_PREFILTERED, T4D_FLAG_ASYNC_STATUS, T4D_FLAG_NO_LONG_BINS = 1, 2, 4, 8, 16
T4D_FLAG_SHORT_BINS = 32
T4D_FLAG_LONG_LISTS = 64
T4D_FLAG_RAW_PARAMS = 128

# every symbol include/topo4d_raster.h declares (tests/test_abi.py checks header <-> this list <-> the .so)
EXPORTS = (
    "t4d_abi_version", "t4d_last_error", "t4d_state_bytes", "t4d_backward_scratch_bytes",
    "t4d_rasterize_forward", "t4d_rasterize_backward", "t4d_fetch_status", "t4d_mark_visible",
    "t4d_debug_state_layout", "t4d_profile_begin", "t4d_profile_end", "t4d_view_dot", "t4d_view_dot_scratch_bytes",
    "t4d_texture_bake", "t4d_texture_render_colors", "t4d_texture_bake_scratch_bytes", "t4d_photometric_loss", "t4d_photometric_scratch_bytes",
    "t4d_masked_l1_loss", "t4d_masked_l1_scratch_bytes",
    "t4d_adam_pin_step", "t4d_adam_pin_step_graph", "t4d_adam_step_counters", "t4d_dense_interpolate", "t4d_activate_forward", "t4d_activate_backward",
    "t4d_sum_views", "t4d_label_mask_target", "t4d_soft_color_loss", "t4d_soft_color_scratch_bytes",
)


class T4DProblem(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("n_views", C.c_int32), ("P", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32),
                ("scale_modifier", C.c_float), ("pair_capacity", C.c_int64), ("flags", C.c_uint32),
                ("views_per_param_set", C.c_uint32)]


class T4DStatus(C.Structure):
    _fields_ = [("max_pairs_per_view", C.c_int64), ("total_pairs", C.c_int64), ("overflow", C.c_int32),
                ("max_tile_pairs", C.c_int32)]


class T4DForwardIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "views", "means3D", "opacities", "scales", "rotations", "cov3D_precomp", "colors_precomp", "shs",
        "out_color", "out_depth", "out_alpha", "out_radii", "state")] + [("state_bytes", C.c_size_t)]


class T4DBackwardIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "views", "means3D", "opacities", "scales", "rotations", "cov3D_precomp", "colors_precomp", "shs",
        "radii", "state")] + [("state_bytes", C.c_size_t)] + [(n, C.c_void_p) for n in (
            "dL_dcolor", "dL_ddepth", "dL_dalpha", "dL_dmeans3D", "dL_dmeans2D", "dL_dcolors", "dL_dshs",
            "dL_dopacities", "dL_dscales", "dL_drotations", "dL_dcov3D", "scratch")] + [
                ("scratch_bytes", C.c_size_t), ("cotangent_dot", C.c_void_p)]


class T4DKernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("total_ms", C.c_double), ("launches", C.c_int64)]


T4D_ADAM_MAX_TENSORS = 12
T4D_MAX_MASK_LABELS = 16


class T4DAdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("pin_mask", C.c_void_p), ("pin_values", C.c_void_p), ("rows", C.c_int64), ("width", C.c_int32),
                ("lr", C.c_float), ("step", C.c_int32), ("flags", C.c_int32)]


T4D_ADAM_CLEAR_GRAD = 1


class ExtensionMissing(RuntimeError):
    pass


_lib = None


def load():
    """Load libtopo4d_raster.so or raise ExtensionMissing.  Never falls back to anything."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExtensionMissing(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m topo4d_amd.build` "
            "(hipcc --offload-arch=gfx950). topo4d_amd has no CPU fallback by design.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise ExtensionMissing(f"could not load {LIB_PATH}: {e}") from e
    lib.t4d_abi_version.restype = C.c_uint32
    lib.t4d_last_error.restype = C.c_char_p
    lib.t4d_state_bytes.restype = C.c_size_t
    lib.t4d_state_bytes.argtypes = [C.POINTER(T4DProblem)]
    lib.t4d_backward_scratch_bytes.restype = C.c_size_t
    lib.t4d_backward_scratch_bytes.argtypes = [C.POINTER(T4DProblem)]
    lib.t4d_rasterize_forward.restype = C.c_int
    lib.t4d_rasterize_forward.argtypes = [C.POINTER(T4DProblem), C.POINTER(T4DForwardIO), C.POINTER(T4DStatus),
                                          C.c_void_p]
    lib.t4d_rasterize_backward.restype = C.c_int
    lib.t4d_rasterize_backward.argtypes = [C.POINTER(T4DProblem), C.POINTER(T4DBackwardIO), C.c_void_p]
    lib.t4d_fetch_status.restype = C.c_int
    lib.t4d_fetch_status.argtypes = [C.POINTER(T4DProblem), C.c_void_p, C.POINTER(T4DStatus), C.c_void_p]
    lib.t4d_mark_visible.restype = C.c_int
    lib.t4d_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.t4d_debug_state_layout.restype = C.c_int
    lib.t4d_debug_state_layout.argtypes = [C.POINTER(T4DProblem), C.c_int, C.POINTER(C.c_uint64), C.c_int]
    lib.t4d_view_dot_scratch_bytes.restype = C.c_size_t
    lib.t4d_view_dot_scratch_bytes.argtypes = [C.c_int32]
    lib.t4d_sum_views.restype = C.c_int
    lib.t4d_sum_views.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p]
    lib.t4d_view_dot.restype = C.c_int
    lib.t4d_view_dot.argtypes = [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.t4d_texture_bake_scratch_bytes.restype = C.c_size_t
    lib.t4d_texture_bake_scratch_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int64]
    lib.t4d_texture_bake.restype = C.c_int
    lib.t4d_texture_bake.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64,
                                     C.POINTER(C.c_int64), C.c_void_p]
    lib.t4d_texture_render_colors.restype = C.c_int
    lib.t4d_texture_render_colors.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 7 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                              C.c_int64, C.POINTER(C.c_int64), C.c_void_p]
    lib.t4d_photometric_scratch_bytes.restype = C.c_size_t
    lib.t4d_photometric_scratch_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.t4d_photometric_loss.restype = C.c_int
    lib.t4d_photometric_loss.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 10 + [C.c_size_t, C.c_void_p]
    lib.t4d_masked_l1_scratch_bytes.restype = C.c_size_t
    lib.t4d_masked_l1_scratch_bytes.argtypes = [C.c_int32]
    lib.t4d_masked_l1_loss.restype = C.c_int
    lib.t4d_masked_l1_loss.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 7 + [C.c_size_t, C.c_void_p]
    lib.t4d_label_mask_target.restype = C.c_int
    lib.t4d_label_mask_target.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.c_void_p, C.c_float,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
    lib.t4d_soft_color_scratch_bytes.restype = C.c_size_t
    lib.t4d_soft_color_scratch_bytes.argtypes = []
    lib.t4d_soft_color_loss.restype = C.c_int
    lib.t4d_soft_color_loss.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                        C.c_size_t, C.c_void_p]
    lib.t4d_adam_pin_step.restype = C.c_int
    lib.t4d_adam_pin_step.argtypes = [C.POINTER(T4DAdamTensor), C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p]
    lib.t4d_adam_step_counters.restype = C.c_int64
    lib.t4d_adam_step_counters.argtypes = [C.POINTER(T4DAdamTensor), C.c_int32]
    lib.t4d_adam_pin_step_graph.restype = C.c_int
    lib.t4d_adam_pin_step_graph.argtypes = [C.POINTER(T4DAdamTensor), C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int64,
                                            C.c_void_p, C.c_void_p]
    lib.t4d_dense_interpolate.restype = C.c_int
    lib.t4d_dense_interpolate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                          C.c_void_p, C.c_void_p]
    lib.t4d_activate_forward.restype = C.c_int
    lib.t4d_activate_forward.argtypes = [C.c_int64] + [C.c_void_p] * 7
    lib.t4d_activate_backward.restype = C.c_int
    lib.t4d_activate_backward.argtypes = [C.c_int64] + [C.c_void_p] * 10
    lib.t4d_profile_begin.restype = C.c_int
    lib.t4d_profile_end.restype = C.c_int
    lib.t4d_profile_end.argtypes = [C.POINTER(T4DKernelTime), C.c_int, C.POINTER(C.c_int)]
    if lib.t4d_abi_version() != T4D_ABI_VERSION:
        raise ExtensionMissing(f"ABI mismatch: library {lib.t4d_abi_version()} vs python {T4D_ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def profile_begin() -> None:
    load().t4d_profile_begin()


def profile_end() -> dict:
    """{kernel name: (total_ms, launches)} since profile_begin()."""
    arr = (T4DKernelTime * 16)()
    n = C.c_int(0)
    rc = load().t4d_profile_end(arr, 16, C.byref(n))
    if rc != T4D_OK:
        raise RuntimeError(f"t4d_profile_end failed: {last_error()}")
    return {arr[i].name.decode(): (arr[i].total_ms, arr[i].launches) for i in range(n.value)}


def last_error() -> str:
    return load().t4d_last_error().decode(errors="replace")
