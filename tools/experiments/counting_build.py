#!/usr/bin/env python
"""The COUNTING build of the render kernels, made from the shipped sources without touching them.

The shipped kernel sources carry no experiment switch of any kind.  What `tools/count_lanes.py` needs - sixteen device counters
(`g_count`: wave-steps, row visits, contributing lanes of k_render_fwd / k_render_bwd) and the export `t4d_debug_read_counters` -
is kept HERE as a list of insertions (`counting_hooks.json`: file, the line to insert before, optionally the line that must
precede it, the text).  This script copies topo4d_amd/csrc/ to topo4d_amd/csrc_count/ (git-ignored), applies the insertions -
every anchor must match exactly once, so a source change that moves a hook fails loudly instead of counting the wrong thing -
and builds topo4d_amd/csrc/variants/lib_count.so:

    python tools/experiments/counting_build.py && T4D_LIB=topo4d_amd/csrc/variants/lib_count.so python tools/count_lanes.py C2 A
"""
import glob
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "topo4d_amd", "csrc")
DST = os.path.join(ROOT, "topo4d_amd", "csrc_count")          # same depth as csrc/: the sources include "../../include/..."
OUT = os.path.join(SRC, "variants", "lib_count.so")


def instrument():
    if os.path.isdir(DST):
        shutil.rmtree(DST)
    os.makedirs(DST)
    for f in glob.glob(os.path.join(SRC, "*.h")) + glob.glob(os.path.join(SRC, "*.hip")):
        shutil.copy(f, DST)
    hooks = json.load(open(os.path.join(HERE, "counting_hooks.json")))
    for name in sorted({h["file"] for h in hooks}):
        path = os.path.join(DST, name)
        lines = open(path).read().split("\n")
        for h in [h for h in hooks if h["file"] == name]:
            hits = [i for i, L in enumerate(lines) if L == h["before"] and ("after" not in h or (i > 0 and lines[i - 1] == h["after"]))]
            if len(hits) != 1:
                raise SystemExit(f"counting_build: anchor of a hook in {name} matches {len(hits)} lines (expected 1): {h['before'].strip()[:80]!r}")
            lines[hits[0]:hits[0]] = h["insert"]
        open(path, "w").write("\n".join(lines))


def build():
    sys.path.insert(0, ROOT)
    from topo4d_amd import build as b
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + b.FLAGS + ["-shared"] + sorted(glob.glob(os.path.join(DST, "*.hip"))) + ["-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    instrument()
    print(build())
