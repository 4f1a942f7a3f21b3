#!/bin/bash
# Runs ON THE GPU BOX: tools/bench_bake.py --no-cpu under the timing builds of k_tex_render (tools/ab_build.sh tex_abl<N> -DT4D_TEX_ABL=<N>;
# results are wrong on purpose) and under the shipped library, with rocprofv3 kernel stats of the latter.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
echo "== shipped"; python tools/bench_bake.py --no-cpu 2>/dev/null | tail -1
for lib in topo4d_amd/csrc/variants/lib_tex_abl*.so; do echo "== $lib"; T4D_LIB=$ROOT/$lib python tools/bench_bake.py --no-cpu 2>/dev/null | tail -1; done
export TMPDIR=/tmp; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/texp -o t -- python $ROOT/tools/bench_bake.py --no-cpu > /dev/null 2>&1; head -8 /tmp/texp/t_kernel_stats.csv | cut -c1-120
