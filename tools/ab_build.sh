#!/bin/bash
# Experiment builds next to the shipped library: tools/ab_build.sh <tag> [extra hipcc flags, e.g. -DT4D_FWD_BATCH=64]
# -> topo4d_amd/csrc/variants/lib_<tag>.so ; run anything with T4D_LIB=<that path> to use it (topo4d_amd/_lib.py).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
mkdir -p $ROOT/topo4d_amd/csrc/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -fno-slp-vectorize \
    "$@" -shared $ROOT/topo4d_amd/csrc/*.hip -o $ROOT/topo4d_amd/csrc/variants/lib_$TAG.so
echo $ROOT/topo4d_amd/csrc/variants/lib_$TAG.so
