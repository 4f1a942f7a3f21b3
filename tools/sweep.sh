#!/bin/bash
# Runs ON THE GPU BOX: rebuilds the library with each set of -D flags (one quoted argument per build) and prints the
# per-kernel times of bench.py; the shipped build is restored at the end.  usage: tools/sweep.sh "-DA=1" "-DA=2 -DB=3" ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for flags in "$@"; do
  T4D_CFLAGS="$flags" python -m topo4d_amd.build --force > /dev/null 2>&1
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('%-40s step %.3f ms  fwd %.1f  bwd %.1f  sort %.1f  pre %.1f  preb %.1f' % ('$flags', d['ms_per_step'], k['k_render_fwd']['avg_us'], k['k_render_bwd']['avg_us'], k['k_sort_tiles']['avg_us'], k['k_preprocess']['avg_us'], k['k_preprocess_bwd']['avg_us']))"
done
python -m topo4d_amd.build --force > /dev/null 2>&1
