# GPU box: dense one-view scene, shipped library (long-tile segments of 128 positions) against lib_seg256.so (256)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
f() { python tools/micro/dense_poles.py 2>/dev/null | grep "full" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d['kernels_us']; print('   view', d['view'], 'fwd', k['k_render_fwd'], 'bwd', k['k_render_bwd'], 'sort', k['k_sort_tiles'], 'sum', d['sum_us'])"; }
for r in 1; do
echo "== 128 (shipped)"; f
echo "== 256"; T4D_LIB=$ROOT/topo4d_amd/csrc/variants/lib_seg256.so f
echo "== 64"; T4D_LIB=$ROOT/topo4d_amd/csrc/variants/lib_seg64.so f
done
