# GPU box: dense one-view scene (views 12 and 4): whole tiles only (T4D_NO_SEGMENTS), and the long-tile threshold 1024 (shipped) / 2048 / 4096 / 8192
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
f() { python tools/micro/dense_poles.py 2>/dev/null | grep "full" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d['kernels_us']; print('   view', d['view'], 'fwd', k['k_render_fwd'], 'bwd', k['k_render_bwd'], 'sum', d['sum_us'])"; }
for r in 1 2; do
echo "== whole tiles"; T4D_NO_SEGMENTS=1 f
echo "== 1024 (shipped)"; f
for t in 2048 4096; do echo "== $t"; T4D_LIB=$ROOT/topo4d_amd/csrc/variants/lib_seglong$t.so f; done
done
