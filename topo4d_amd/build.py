"""
Build recipe for the HIP extension: `python -m topo4d_amd.build [--force]`.

One translation unit per .hip file under csrc/, compiled for gfx950 only and linked into
csrc/libtopo4d_raster.so (in-tree so that it travels to the GPU box; *.so is git-ignored).
hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libtopo4d_raster.so")
# -fno-slp-vectorize: the SLP pass pairs up the alpha evaluations of two different splats into v_pk_* instructions and
# pays for it with a v_mov per operand (measured: render_fwd 149 -> 126 us without it); packed math is written explicitly
# (ext_vector_type) where the operands are adjacent by construction.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-fno-slp-vectorize"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def raster_source_sha256() -> str:
    """sha256 over the rasterizer's translation unit - csrc/t4d_raster.hip and the t4d_raster_*.h parts it includes, in name
    order: the stamp the committed rocprofv3 counters carry (tools/prof.sh, tools/count_lanes.py) and bench.py compares."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "t4d_raster*.hip")) + glob.glob(os.path.join(CSRC, "t4d_raster*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(HERE, "..", "include", "*.h")) + glob.glob(os.path.join(CSRC, "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("T4D_CFLAGS", "").split()          # e.g. -DT4D_FWD_BATCH=128 (tuning constants only: the sources carry no experiment switches)
    cmd = [hipcc] + FLAGS + extra + ["-shared"] + sources() + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
