#!/usr/bin/env python3
"""static instruction mix of ONE kernel attributed to the source functions inlined into it:
   tools/isa.sh /tmp/a.s -gline-tables-only && tools/isa_by_function.py /tmp/a.s k_sweep
(the innermost .loc line of every instruction decides; functions = the definitions in gph_locus.h / gph_kernels.h /
gph_math.h / gph_rt.h by line range)"""
import collections, os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm, kern = sys.argv[1], sys.argv[2]
lines = open(asm).read().splitlines()
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
# function ranges per source file
ranges = {}
for fn in ("gph_locus.h", "gph_kernels.h", "gph_math.h", "gph_rt.h", "gph_engine.hip", "gph_global.h"):
    path = os.path.join(os.environ.get("GPH_ISA_REPO", REPO), "g-phocs_amd", "csrc", fn)
    if not os.path.exists(path):
        continue
    src = open(path).read().splitlines()
    starts = []
    for i, l in enumerate(src, 1):
        m = re.match(r'^(?:template <[^>]*>\s*)?(?:GPH_DEV\w*|GPH_MATH_FN|GPH_HD\w*|GPH_KERNEL\(|static|__global__)[^;=]*?\b(\w+)\s*\(', l)
        if m and not l.strip().endswith(';'):
            starts.append((i, m.group(1)))
    ranges[fn] = starts
def func_of(fn, line):
    best = "?"
    for s, name in ranges.get(fn, []):
        if s <= line: best = name
        else: break
    return best
inside, cur = False, ("?", 0)
cnt = collections.defaultdict(collections.Counter)
for l in lines:
    if re.match(r'^_Z\w*%s\w*:' % kern, l): inside = True; continue
    if inside and l.startswith('.Lfunc_end'): break
    if not inside: continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2))); continue
    t = l.strip().split()
    if not t or t[0].startswith(('.', ';', '//')) or t[0].endswith(':'): continue
    op = t[0]
    f = func_of(*cur)
    c = cnt[f]
    c['total'] += 1
    if op.startswith('v_'): c['valu'] += 1
    elif op.startswith('s_load'): c['smem'] += 1
    elif op.startswith('s_waitcnt'): c['wait'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    else: c['vmem'] += 1
tot = sum(c['total'] for c in cnt.values())
print(f"{kern}: {tot} instructions")
for f, c in sorted(cnt.items(), key=lambda kv: -kv[1]['total']):
    print(f"  {f:28s} " + " ".join(f"{n}={c[n]:5d}" for n in ('total', 'valu', 'salu', 'smem', 'lds', 'vmem', 'wait')))
