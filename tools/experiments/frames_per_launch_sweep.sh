cd ${GRAFT_REPO_ROOT:-/root/repo}
for g in 1 2 3 4 8; do for f in 1 2 3; do
python bench.py --shard views --frames-per-launch $g --frames-in-flight $f --no-extras --no-cpu-baseline --steps 48 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('frames_per_launch $g in_flight $f: value', d['value'], 'ms/frame', d['ms_per_step'], 'set', d['config']['ms_per_launch_set'])"
done; done
