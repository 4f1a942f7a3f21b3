#!/bin/bash
# Runs ON THE GPU BOX: rebuilds the library with each set of -D flags (one quoted argument per build) and prints every kernel's
# HIP-event time from bench.py (one frame in flight); the shipped build is restored at the end.
#   usage: tools/sweep_kernels.sh "-DA=1" "-DA=2 -DB=3" ...     (BENCH_ARGS="--opacity B" for other workloads)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for flags in "$@"; do
  T4D_CFLAGS="$flags" python -m topo4d_amd.build --force > /dev/null 2>&1
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --frames-in-flight 1 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('%-44s step %.3f ms  ' % ('$flags', d['ms_per_step']) + '  '.join('%s %.1f' % (n[2:], v['avg_us']) for n, v in k.items()))"
done
python -m topo4d_amd.build --force > /dev/null 2>&1
