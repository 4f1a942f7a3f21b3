# Runs ON THE GPU BOX: per-tile kernels' grid-size policy (T4D_TILE_DIV: tiles per workgroup of the grid-stride loops; 0 = the built-in policy) at config 2 and 4.
for cfg in C2 C4; do
for d in 0 1 2 3 4 8 16; do
  T4D_TILE_DIV=$d python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-extras --frames-in-flight 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$cfg div $d  step %.3f ms  ' % d['ms_per_step'] + '  '.join('%s %.1f' % (n[2:], v['avg_us']) for n, v in k.items() if 'render' in n or 'sort' in n))"
done
done
