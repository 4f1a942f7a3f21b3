#!/bin/bash
# Runs ON THE GPU BOX: rebuilds the library with each set of -D flags and prints tools/bench_loss.py's kernel times.
#   usage: tools/sweep_loss.sh "-DT4D_PH_WAVES=3" "-DT4D_PH_WAVES=4" ...   (env T4D_PH_ROWS=<rows per segment> is passed through)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for flags in "$@"; do
  T4D_CFLAGS="$flags" python -m topo4d_amd.build --force > /dev/null 2>&1
  for rows in ${ROWS:-0}; do
    T4D_PH_ROWS=$rows python tools/bench_loss.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-36s rows %-4s  24x512^2 %.1f us   1x512x375 %.1f us   24x2048^2 %.1f us   with autograd %.3f ms' % ('$flags', '$rows', d['kernels_us_24x512x512'], d['kernels_us_1x512x375'], d['kernels_us_24x2048x2048'], d['fused_hip_ms_per_24_views']))"
  done
done
python -m topo4d_amd.build --force > /dev/null 2>&1
