#!/usr/bin/env python
"""Static census of a kernel's ISA (from `hipcc --save-temps`): VALU / SALU / LDS / VMEM instructions per basic block, so that
"where do the vector instructions sit" has an answer that does not need the GPU.
    python tools/isa_census.py <file.s> <kernel-name-substring> [<substring> ...]"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
for want in sys.argv[2:]:
    start = next(i for i, l in enumerate(src) if re.match(r'^_Z\w*' + re.escape(want) + r'\w*:', l))
    end = next(i for i in range(start, len(src)) if src[i].startswith('.Lfunc_end'))
    print('=====', src[start].split(':')[0], end - start, 'lines')
    cnt = dict(v=0, s=0, ds=0, g=0); tot = dict(cnt)
    def flush(tag):
        if any(cnt.values()):
            print('%-58s VALU %4d SALU %4d LDS %3d VMEM %3d' % (tag, cnt['v'], cnt['s'], cnt['ds'], cnt['g']))
        for k in cnt: tot[k] += cnt[k]; cnt[k] = 0
    for l in src[start + 1:end]:
        t = l.strip()
        if re.match(r'^\.LBB\d+_\d+:', t): flush('  .. to ' + t.split(':')[0]); print(t.split(':')[0] + ':'); continue
        if not t or t[0] in ';.': continue
        op = t.split()[0]
        if op.startswith('v_'): cnt['v'] += 1
        elif op.startswith('ds_'): cnt['ds'] += 1
        elif op.split('_')[0] in ('global', 'buffer', 'flat', 'scratch'): cnt['g'] += 1
        elif op.startswith('s_'):
            cnt['s'] += 1
            if op == 's_barrier' or op.startswith('s_cbranch') or op == 's_branch' or op == 's_endpgm': flush('    ' + t[:52])
    flush('end'); print('TOTAL', tot)
