#!/bin/bash
# Runs ON THE GPU BOX: per-kernel HIP-event times of bench.py (one frame in flight) for the shipped library and for experiment
# builds made HERE by tools/ab_build.sh (they travel with the snapshot).   usage: tools/ab_bench.sh <tag> [<tag> ...]
#   tag "shipped" = topo4d_amd/csrc/libtopo4d_raster.so;  BENCH_ARGS="--config C4" for other workloads;  REPS=2 repeats the round
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for rep in $(seq 1 ${REPS:-1}); do
for tag in "$@"; do
  lib=$ROOT/topo4d_amd/csrc/variants/lib_$tag.so
  [ "$tag" = shipped ] && lib=$ROOT/topo4d_amd/csrc/libtopo4d_raster.so
  T4D_LIB=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --frames-in-flight 1 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('%-10s step %.3f ms  ' % ('$tag', d['ms_per_step']) + '  '.join('%s %.1f' % (n[2:], v['avg_us']) for n, v in k.items()))"
done
done
