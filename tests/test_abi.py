"""CPU: the C-ABI shared library loads, exports every symbol include/topo4d_raster.h declares, and its struct
layouts match the ctypes mirrors.  No compute call is made (there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "topo4d_raster.h")


def header_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(t4d_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from topo4d_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    declared = header_functions()
    assert declared, "no functions parsed from the header"
    assert sorted(_lib.EXPORTS) == declared
    for name in declared:
        assert getattr(lib, name) is not None
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert exported == declared, "the .so exports exactly the header's functions (everything else is hidden)"
    assert lib.t4d_abi_version() == _lib.T4D_ABI_VERSION


def test_struct_layouts_match_ctypes():
    from topo4d_amd import _lib
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "topo4d_raster.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(T4DProblem), sizeof(T4DStatus), sizeof(T4DForwardIO), sizeof(T4DBackwardIO), sizeof(T4DKernelTime));
  printf("%zu %zu %zu %zu\n", offsetof(T4DProblem, pair_capacity), offsetof(T4DProblem, flags), offsetof(T4DProblem, scale_modifier), offsetof(T4DProblem, views_per_param_set));
  printf("%zu %zu\n", offsetof(T4DForwardIO, state_bytes), offsetof(T4DBackwardIO, scratch_bytes));
  printf("%d %d %d\n", T4D_ABI_VERSION, T4D_VIEW_FLOATS, T4D_GRAD_PAIR_FLOATS);
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe])   # header is plain C
        lines = subprocess.check_output([exe], text=True).split("\n")
    sizes = [int(x) for x in lines[0].split()]
    assert sizes == [C.sizeof(_lib.T4DProblem), C.sizeof(_lib.T4DStatus), C.sizeof(_lib.T4DForwardIO),
                     C.sizeof(_lib.T4DBackwardIO), C.sizeof(_lib.T4DKernelTime)]
    offs = [int(x) for x in lines[1].split()]
    assert offs == [_lib.T4DProblem.pair_capacity.offset, _lib.T4DProblem.flags.offset, _lib.T4DProblem.scale_modifier.offset,
                    _lib.T4DProblem.views_per_param_set.offset]
    offs2 = [int(x) for x in lines[2].split()]
    assert offs2 == [_lib.T4DForwardIO.state_bytes.offset, _lib.T4DBackwardIO.scratch_bytes.offset]
    consts = [int(x) for x in lines[3].split()]
    assert consts == [_lib.T4D_ABI_VERSION, _lib.T4D_VIEW_FLOATS, _lib.T4D_GRAD_PAIR_FLOATS]


def test_size_queries_and_argument_errors_without_a_gpu():
    from topo4d_amd import _lib
    lib = _lib.load()
    prob = _lib.T4DProblem(_lib.T4D_ABI_VERSION, 24, 30000, 512, 512, 0, 0, 1.0, 120000, 0, 0)
    sb = lib.t4d_state_bytes(C.byref(prob))
    assert sb > 24 * 120000 * 8 and sb % 256 == 0
    assert lib.t4d_backward_scratch_bytes(C.byref(prob)) >= 24 * 120000 * 4 * _lib.T4D_GRAD_PAIR_FLOATS
    bad = _lib.T4DProblem(99, 24, 30000, 512, 512, 0, 0, 1.0, 120000, 0, 0)
    assert lib.t4d_state_bytes(C.byref(bad)) == 0
    assert b"abi_version" in lib.t4d_last_error()
    # NULL pointers are rejected before anything touches a device
    io = _lib.T4DForwardIO()
    assert lib.t4d_rasterize_forward(C.byref(prob), C.byref(io), None, None) == _lib.T4D_ERR_ARG
    bio = _lib.T4DBackwardIO()
    assert lib.t4d_rasterize_backward(C.byref(prob), C.byref(bio), None) == _lib.T4D_ERR_ARG
    zero = _lib.T4DProblem(_lib.T4D_ABI_VERSION, 0, 30000, 512, 512, 0, 0, 1.0, 120000, 0, 0)
    assert lib.t4d_state_bytes(C.byref(zero)) == 0
    assert lib.t4d_view_dot_scratch_bytes(24) == 24 * 64 * 4


def test_every_compute_export_rejects_bad_arguments_before_touching_a_device():
    """SURVEY 8(b) 'Errors': a non-zero code and a message, never a crash - checked for every compute entry point with the arguments a
    careless binding would pass (NULL buffers, zero or negative sizes, scratch too small, more labels than the table holds)."""
    from topo4d_amd import _lib
    lib = _lib.load()
    ARG, SIZE = _lib.T4D_ERR_ARG, _lib.T4D_ERR_STATE_SIZE
    one = C.c_void_p(64)                                  # "some address": never dereferenced by a call that is rejected
    none = None

    def rejected(rc, code=ARG):
        assert rc == code, (rc, lib.t4d_last_error())
        assert lib.t4d_last_error()                       # every refusal leaves a message

    # photometric loss: NULL buffers, empty images, unpaired affine, scratch too small
    nb = lib.t4d_photometric_scratch_bytes(2, 64, 64)
    assert nb > 0 and lib.t4d_photometric_scratch_bytes(0, 64, 64) == 0
    rejected(lib.t4d_photometric_loss(2, 64, 64, none, one, none, none, none, one, one, none, none, one, nb, none))
    rejected(lib.t4d_photometric_loss(2, 0, 64, one, one, none, none, none, one, one, none, none, one, nb, none))
    rejected(lib.t4d_photometric_loss(2, 64, 64, one, one, one, none, none, one, one, none, none, one, nb, none))
    rejected(lib.t4d_photometric_loss(2, 64, 64, one, one, none, none, none, one, one, one, none, one, nb, none))
    rejected(lib.t4d_photometric_loss(2, 64, 64, one, one, none, none, none, one, one, none, none, one, nb - 1, none), SIZE)
    rejected(lib.t4d_masked_l1_loss(1, 8, 8, one, one, none, none, one, one, one, lib.t4d_masked_l1_scratch_bytes(1), none))
    rejected(lib.t4d_masked_l1_loss(1, 8, 8, one, one, one, none, one, one, one, 0, none), SIZE)
    # masked target: no output at all, a target without its image, too many labels
    colors = (C.c_float * (3 * (_lib.T4D_MAX_MASK_LABELS + 1)))()
    rejected(lib.t4d_label_mask_target(1, 8, 8, one, colors, 1, one, 0.1, none, none, none))
    rejected(lib.t4d_label_mask_target(1, 8, 8, one, colors, 1, none, 0.1, none, one, none))
    rejected(lib.t4d_label_mask_target(1, 8, 8, one, colors, _lib.T4D_MAX_MASK_LABELS + 1, one, 0.1, one, one, none))
    rejected(lib.t4d_label_mask_target(0, 8, 8, one, colors, 1, one, 0.1, one, one, none))
    # soft colour
    sb = lib.t4d_soft_color_scratch_bytes()
    rejected(lib.t4d_soft_color_loss(0, 3, one, one, 1.0, one, none, 0, one, sb, none))
    rejected(lib.t4d_soft_color_loss(4, 3, one, none, 1.0, one, none, 0, one, sb, none))
    rejected(lib.t4d_soft_color_loss(4, 3, one, one, 1.0, one, none, 0, one, sb - 1, none), SIZE)
    # Adam + pins: no tensors, too many, a gradient without moments, a mask without values, step 0, graph form without its device words
    T = _lib.T4DAdamTensor
    ok = T(64, 64, 64, 64, None, None, 4, 3, 1e-3, 1, 0)
    arr = lambda *ts: (T * len(ts))(*ts)
    rejected(lib.t4d_adam_pin_step(None, 1, 0.9, 0.999, 1e-15, none))
    rejected(lib.t4d_adam_pin_step(arr(*[ok] * (_lib.T4D_ADAM_MAX_TENSORS + 1)), _lib.T4D_ADAM_MAX_TENSORS + 1, 0.9, 0.999, 1e-15, none))
    rejected(lib.t4d_adam_pin_step(arr(T(64, 64, None, 64, None, None, 4, 3, 1e-3, 1, 0)), 1, 0.9, 0.999, 1e-15, none))
    rejected(lib.t4d_adam_pin_step(arr(T(64, 64, 64, 64, 64, None, 4, 3, 1e-3, 1, 0)), 1, 0.9, 0.999, 1e-15, none))
    rejected(lib.t4d_adam_pin_step(arr(T(64, 64, 64, 64, None, None, 4, 3, 1e-3, 0, 0)), 1, 0.9, 0.999, 1e-15, none))
    rejected(lib.t4d_adam_pin_step_graph(arr(ok), 1, 0.9, 0.999, 1e-15, none, 1, one, none))
    rejected(lib.t4d_adam_pin_step_graph(arr(ok), 1, 0.9, 0.999, 1e-15, one, 2, one, none))       # a counter array of another layout
    assert lib.t4d_adam_step_counters(arr(ok, T(64, 64, 64, 64, None, None, 1000, 3, 1e-3, 1, 0)), 2) == 1 + 12
    assert lib.t4d_adam_step_counters(None, 2) == 0
    # activations, dense interpolation, view sums and dots, visibility
    rejected(lib.t4d_activate_forward(4, one, one, one, one, one, none, none))
    rejected(lib.t4d_activate_backward(4, one, none, one, one, one, one, one, one, one, none))
    rejected(lib.t4d_activate_forward(-1, one, one, one, one, one, one, none))
    assert lib.t4d_activate_forward(0, none, none, none, none, none, none, none) == _lib.T4D_OK      # nothing to do is not an error
    rejected(lib.t4d_dense_interpolate(one, none, one, one, 4, 4, 3, one, none))
    rejected(lib.t4d_dense_interpolate(one, one, one, one, 4, 4, 0, one, none))
    rejected(lib.t4d_sum_views(0, 1, None, None, None, none))
    rejected(lib.t4d_view_dot(0, 16, one, one, one, one, none))
    rejected(lib.t4d_mark_visible(4, one, none, one, none))
    assert lib.t4d_mark_visible(0, none, none, none, none) == _lib.T4D_OK
    # texture bake: NULL buffers, an empty or inverted row band, no pair arena, scratch too small
    need = C.c_int64(0)
    tb = lib.t4d_texture_bake_scratch_bytes(64, 64, 1000)
    assert tb > 0
    bake = lambda *a: lib.t4d_texture_bake(*a)
    rejected(bake(none, one, one, 10, 4, 64, 64, 3, 0, 64, one, one, one, tb, 1000, C.byref(need), none))
    rejected(bake(one, one, one, 10, 4, 64, 64, 3, 32, 32, one, one, one, tb, 1000, C.byref(need), none))
    rejected(bake(one, one, one, 10, 4, 64, 64, 3, 0, 65, one, one, one, tb, 1000, C.byref(need), none))
    rejected(bake(one, one, one, 10, 4, 64, 64, 3, 0, 64, one, one, one, tb, 0, C.byref(need), none))
    rejected(bake(one, one, one, 10, 4, 64, 64, 3, 0, 64, one, one, one, tb - 1, 1000, C.byref(need), none), SIZE)
    rejected(lib.t4d_texture_render_colors(one, one, one, none, 0, 4, 64, 64, 3, 0, 64, one, one, one, tb, 1000, C.byref(need), none))
