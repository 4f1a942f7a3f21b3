# GPU box: one and three views per call (Topo4D's shape and the config-2 scene) under the shipped library and under lib_prehybrid.so
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
f() { python tools/small_launch.py 2>/dev/null | awk '/default/{p=1} /throughput build/{p=0} p' | grep '"V": [13],' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['shape'], 'V', d['V'], 'gpu_us', d['gpu_us'], 'wall', d['wall_us'], 'fwd', d['fwd'], 'bwd', d['bwd'], 'pre', d['pre'], 'pre_bwd', d['pre_bwd'])"; }
for r in 1 2; do
echo "== shipped"; f
echo "== pre-hybrid"; T4D_LIB=$ROOT/topo4d_amd/csrc/variants/lib_prehybrid.so f
done
