#!/usr/bin/env python3
"""register-copy overhead per loop of ONE kernel:  tools/isa.sh /tmp/a.s -gline-tables-only && tools/isa_loops.py /tmp/a.s k_sweep
Blocks carry the assembler's "in Loop: Header=BBn_m Depth=d" annotation; every instruction is charged to the innermost
loop of its block.  Per loop: instructions, VALU, register copies (v_mov_b32/b64 between registers, s_mov between
registers) and the source lines (file:line of the .loc tables) the loop covers -- the copies are what the register
allocator paid for merges (phi copies) and are the first thing to look at in a hot loop."""
import collections, re, sys
asm, kern = sys.argv[1], sys.argv[2]
lines = open(asm).read().splitlines()
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
inside = False
loop = None
cur = None
st = collections.defaultdict(lambda: dict(n=0, valu=0, vmov=0, smov=0, lines=collections.Counter(), depth=0))
for l in lines:
    if re.match(r'^_Z\w*%s\w*:' % kern, l): inside = True; continue
    if inside and l.startswith('.Lfunc_end'): break
    if not inside: continue
    m = re.match(r'^\.LBB\d+_\d+:\s*(;.*)?$', l)
    if m:
        c = m.group(1) or ''
        mm = re.search(r'Header=(BB\d+_\d+) Depth=(\d+)', c)
        if mm: loop = mm.group(1); st[loop]['depth'] = int(mm.group(2))
        elif 'Loop Header' in c:
            mm = re.search(r'Depth=(\d+)', c); loop = l.split(':')[0][2:]; st[loop]['depth'] = int(mm.group(1)) if mm else 0
        else: loop = None
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (files.get(int(m.group(1)), '?'), int(m.group(2))); continue
    t = l.strip().split(';')[0].split()
    if not t or t[0].startswith('.') or t[0].endswith(':') or loop is None: continue
    s = st[loop]; s['n'] += 1
    op = t[0]
    if op.startswith('v_'): s['valu'] += 1
    args = ' '.join(t[1:])
    if op in ('v_mov_b32_e32', 'v_mov_b64_e32') and re.match(r'v\S*, v', args): s['vmov'] += 2 if 'b64' in op else 1
    if op in ('s_mov_b32', 's_mov_b64') and re.match(r's\S*, s\[?\d', args): s['smov'] += 1
    if cur and cur[1]: s['lines'][cur] += 1
print(f"{'loop':12s} depth instrs  valu  v-copies(dwords) s-copies  main source lines")
for k, s in sorted(st.items(), key=lambda kv: -(kv[1]['vmov'] + kv[1]['smov'])):
    if s['n'] < 8: continue
    top = ' '.join(f"{f.replace('gph_','').replace('.h','')}:{ln}" for (f, ln), _ in s['lines'].most_common(4))
    print(f"{k:12s} {s['depth']:5d} {s['n']:6d} {s['valu']:5d} {s['vmov']:8d} {s['smov']:14d}  {top}")
