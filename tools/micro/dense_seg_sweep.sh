# GPU box: dense one-view scene (views 12 and 4): the long-tile threshold 2048 (shipped) / 1024 / 4096 (tools/ab_build.sh seglong<N> -DT4D_SEG_LONG_MIN=<N>),
# and whole tiles only (T4D_NO_SEGMENTS=1 T4D_NO_LONG_FWD=1)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
f() { python tools/micro/dense_poles.py 2>/dev/null | grep "full" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d['kernels_us']; print('   view', d['view'], 'fwd', k['k_render_fwd'], 'bwd', k['k_render_bwd'], 'sort', k['k_sort_tiles'], 'sum', d['sum_us'])"; }
echo "== whole tiles, one-pass forward"; T4D_NO_SEGMENTS=1 T4D_NO_LONG_FWD=1 f
echo "== 2048 (shipped)"; f
for t in 1024 4096; do echo "== $t"; T4D_LIB=$ROOT/topo4d_amd/csrc/variants/lib_seglong$t.so f; done
