"""v_cndmask_b32_e32 vD, vA, vB, vcc -> the VOP3 encoding when it directly follows another v_cndmask (tools/probe/class_probe.cpp:
consecutive VOP2 selects on vcc hold the SIMD ~20 cycles each, the VOP3 form 4.2).  MODE=all rewrites every one."""
import os, re, sys
mode = os.environ.get("MODE", "second")
out, prev_cnd, n = [], False, 0
for l in open(sys.argv[1]):
    t = l.strip()
    is_ins = bool(t) and not t.startswith(('.', ';', '//')) and not t.split()[0].endswith(':')
    if t.startswith('v_cndmask_b32_e32') and (mode == "all" or prev_cnd):
        l = l.replace('v_cndmask_b32_e32', 'v_cndmask_b32_e64', 1)
        n += 1
    if is_ins:
        prev_cnd = t.startswith('v_cndmask_b32')
    out.append(l)
open(sys.argv[2], 'w').write(''.join(out))
print(f"patch_cnd_e64: {n} instructions rewritten", file=sys.stderr)
