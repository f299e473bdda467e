#!/usr/bin/env python3
"""static instruction mix of the kernels in a device assembly file (tools/isa.sh):  tools/isa_count.py out.s [kernel-substring]"""
import collections, re, sys
txt = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else "k_"
cur, counts = None, {}
for l in txt:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        cur = m.group(1)
        counts[cur] = collections.Counter()
        continue
    if cur is None:
        continue
    if l.startswith('.Lfunc_end'):
        cur = None
        continue
    t = l.strip().split()
    if not t or t[0].startswith(('.', ';', '//')) or t[0].endswith(':'):
        continue
    op = t[0]
    c = counts[cur]
    c['total'] += 1
    if op.startswith('v_'): c['valu'] += 1
    elif op.startswith('s_load') or op.startswith('s_buffer'): c['smem'] += 1
    elif op.startswith('s_waitcnt'): c['wait'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    elif op.startswith(('global_', 'scratch_', 'buffer_', 'flat_')): c['vmem'] += 1
    else: c['other'] += 1
for k, c in counts.items():
    if want in k:
        print(f"{k[:48]:48s} " + " ".join(f"{n}={c[n]}" for n in ('total', 'valu', 'salu', 'smem', 'lds', 'vmem', 'wait')))
