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
  printf("%zu %zu %zu\n", offsetof(T4DProblem, pair_capacity), offsetof(T4DProblem, flags), offsetof(T4DProblem, scale_modifier));
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
    assert offs == [_lib.T4DProblem.pair_capacity.offset, _lib.T4DProblem.flags.offset, _lib.T4DProblem.scale_modifier.offset]
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
