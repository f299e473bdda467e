#!/usr/bin/env python3
"""Assembly-level instrumentation of k_sweep: a counter per basic block (tools/bbcount.sh).
   patch_bbcount.py in.s out.s
Every basic block of the kernel (labels .LBB1_N and the label-less fall-through blocks `; %bb.N:`) gets
   save exec; exec = lane 0; global_atomic_add counters[N], 1; s_waitcnt vmcnt(0); restore exec
on registers the kernel does not use (v64, v65, four SGPRs behind the kernel's own; the kernel descriptor is widened -- occupancy
does not matter for counting).  The counters live in the decision-transcript buffer of the engine (GphDev.slog, kernel
argument offset of k_sweep's GphDev + 112, read from the metadata; allocated by gph_engine_steplog_enable in a -DGPH_BBCOUNT build).  The wait after the
atomic keeps the kernel's own vmcnt bookkeeping exact (nothing of the instrumentation is outstanding when its code goes on)."""
import re
import sys

import os
KERNEL = os.environ.get("BB_KERNEL", "_Z7k_sweep8GphKargs6GphDeviidddd")     # any kernel whose second argument is GphDev
SLOG_IN_GPHDEV = 112             # offsetof(GphDev, slog) (gph_kernels.h: ten pointers, L, Ltot, locus_begin, err, slog_map)


def slog_kernarg_offset(text):
    """kernarg offset of k_sweep's second argument (GphDev, by value), from the code-object metadata in the assembly"""
    j = text.index(".name:           " + KERNEL)
    k = text.rfind("- .agpr_count", 0, j)
    offs = re.findall(r"\.offset:\s+(\d+)\n\s+\.size:\s+(\d+)", text[k:j])
    assert len(offs) >= 2 and int(offs[1][1]) == 136, offs       # sizeof(GphDev)
    return int(offs[1][0]) + SLOG_IN_GPHDEV


def snippet(n, b):
    return [f"\ts_mov_b64 s[{b + 2}:{b + 3}], exec", "\ts_mov_b64 exec, 1", f"\tv_mov_b32_e32 v64, {4 * n}",
            f"\tglobal_atomic_add v64, v65, s[{b}:{b + 1}]", "\ts_waitcnt vmcnt(0)", f"\ts_mov_b64 exec, s[{b + 2}:{b + 3}]"]


def main():
    text = open(sys.argv[1]).read()
    slog_off = slog_kernarg_offset(text)
    d0 = text.index(".amdhsa_kernel " + KERNEL)
    nsg = int(re.search(r"\.amdhsa_next_free_sgpr (\d+)", text[d0:]).group(1))
    nvg = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", text[d0:]).group(1))
    base = (nsg + 1) & ~1                      # four scalar registers behind the kernel's own
    assert base + 4 <= 96 and nvg <= 64, (nsg, nvg)
    src = text.split("\n")
    out, inside, nblocks, desc = [], False, 0, None
    for ln in src:
        if ln.startswith(KERNEL + ":"):
            inside = True
        if inside and ln.startswith(".Lfunc_end"):
            inside = False
        out.append(ln)
        t = ln.strip()
        if t.startswith(".amdhsa_kernel "):
            desc = t.split()[1]
        if desc == KERNEL:
            if t.startswith(".amdhsa_next_free_vgpr"):
                out[-1] = "\t\t.amdhsa_next_free_vgpr 72"
            elif t.startswith(".amdhsa_accum_offset"):
                out[-1] = "\t\t.amdhsa_accum_offset 72"
            elif t.startswith(".amdhsa_next_free_sgpr"):
                out[-1] = f"\t\t.amdhsa_next_free_sgpr {base + 4}"
        if t == ".end_amdhsa_kernel":
            desc = None
        if not inside:
            continue
        m = re.match(r"^\.LBB\d+_(\d+):", ln) or re.match(r"^; %bb\.(\d+):", ln)
        if m:
            n = int(m.group(1))
            if n == 0:
                out += [f"\ts_load_dwordx2 s[{base}:{base + 1}], s[0:1], {hex(slog_off)}", "\tv_mov_b32_e32 v65, 1", "\ts_waitcnt lgkmcnt(0)"]
            out += snippet(n, base)
            nblocks += 1
    open(sys.argv[2], "w").write("\n".join(out))
    print(f"patch_bbcount: {nblocks} basic blocks of {KERNEL} instrumented", file=sys.stderr)


if __name__ == "__main__":
    main()
