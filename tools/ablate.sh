#!/bin/bash
# Runs ON THE GPU BOX: rebuilds the library with each ablation macro and prints the per-kernel times of bench.py.
# Ablation builds produce wrong results by design; the shipped build is restored at the end.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for abl in 0 3 4 5; do
  T4D_CFLAGS="-DT4D_ABL=$abl" python -m topo4d_amd.build --force > /dev/null 2>&1
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --frames-in-flight 1 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('ABL=$abl step %.3f ms  fwd %.1f  bwd %.1f  sort %.1f' % (d['ms_per_step'], k['k_render_fwd']['avg_us'], k['k_render_bwd']['avg_us'], k['k_sort_tiles']['avg_us']))"
done
python -m topo4d_amd.build --force > /dev/null 2>&1
