#!/usr/bin/env python3
"""Step 3 of tools/bbcount.sh: executions per basic block (gpurun_out/bbcount.json) x the instructions of the block in the
un-instrumented assembly (bench_cache/bbcount_dev.s.gz, compiled with line tables: every instruction carries its source line
and inline chain) = dynamic instructions of k_sweep per locus and sweep, by class, by source function (exclusive and
inclusive), and the hottest blocks.   python3 tools/bbcount_report.py [counts.json] > profiles/rNN_bbcount.txt"""
import collections
import gzip
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = os.environ.get("BB_KERNEL", "_Z7k_sweep8GphKargs6GphDeviidddd")
CSRC = os.path.join(REPO, "g-phocs_amd", "csrc")


def cls(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "v lane r/w"
    if op.startswith("v_mov"): return "v mov"
    if op.startswith(("v_cmp", "v_cndmask")): return "v cmp/select"
    if op.startswith("v_") and "f64" in op: return "v f64"
    if op.startswith("v_"): return "v int/other"
    if op.startswith(("s_waitcnt", "s_nop")): return "wait/nop"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_barrier")): return "branch"
    if op.startswith("s_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    return "vmem"


CLASSES = ["v f64", "v int/other", "v mov", "v lane r/w", "v cmp/select", "salu", "branch", "wait/nop", "lds", "smem", "vmem"]


def functions_of(path):
    """line -> name of the enclosing function (a definition starts at column 0 and its body opens on the same or next line)"""
    out, cur = {}, None
    lines = open(path, errors="replace").read().split("\n")
    hdr = re.compile(r"^(?:template\s*<[^>]*>\s*)?(?:GPH_\w+|static|inline|__global__|__device__|extern)\b[^;=]*?\b([A-Za-z_]\w*)\s*\([^;]*$")
    for i, ln in enumerate(lines, 1):
        m = hdr.match(ln)
        if m and not ln.startswith(("#", "//")) and m.group(1) not in ("if", "for", "while", "switch", "defined", "__attribute__", "__launch_bounds__"):
            cur = m.group(1)
        out[i] = cur
    return out


def main():
    cj = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "bbcount.json")))
    loci = cj["loci"] * int(os.environ.get("BB_LAUNCHES", "1"))      # per locus and launch
    counts = {int(k): v for k, v in cj["counts"].items()}
    src = gzip.open(os.path.join(REPO, "bench_cache", "bbcount_dev.s.gz"), "rt").read().split("\n")
    fmap = {}

    def fn(fileline):
        f, l = fileline.rsplit(":", 1)
        f = os.path.basename(f)
        if f not in fmap:
            p = os.path.join(CSRC, f)
            fmap[f] = functions_of(p) if os.path.exists(p) else {}
        return fmap[f].get(int(l)) or f
    inside, blk, chain = False, None, ()
    blocks = collections.OrderedDict()          # id -> list of (op, chain)   chain = tuple of function names, innermost first
    for ln in src:
        if ln.startswith(KERNEL + ":"):
            inside = True
            continue
        if inside and ln.startswith(".Lfunc_end"):
            break
        if not inside:
            continue
        m = re.match(r"^\.LBB\d+_(\d+):", ln) or re.match(r"^; %bb\.(\d+):", ln)
        if m:
            blk = int(m.group(1))
            blocks.setdefault(blk, [])
            continue
        t = ln.strip()
        if t.startswith(".loc"):
            c = t.split(";", 1)[1] if ";" in t else ""
            locs = re.findall(r"([\w./\-]+\.(?:h|hip|cpp)):(\d+):\d+", c)
            names = []
            for f, l in locs:
                n = fn(f + ":" + l)
                if not names or names[-1] != n:
                    names.append(n)
            chain = tuple(names) or ("?",)
            continue
        if not ln.startswith("\t") or not t or t.startswith((".", ";")) or blk is None:
            continue
        blocks[blk].append((t.split()[0], chain, t))
    tot = collections.Counter()
    excl = collections.defaultdict(collections.Counter)
    incl = collections.defaultdict(collections.Counter)
    hot = []
    for b, ins in blocks.items():
        n = counts.get(b, 0)
        if not n or not ins:
            continue
        for op, ch, _ in ins:
            c = cls(op)
            tot[c] += n
            excl[ch[0]][c] += n
            for f in set(ch):
                incl[f][c] += n
        hot.append((n * len(ins), b, n, len(ins)))
    per = lambda v: v / loci
    allv = sum(tot.values())
    print(f"# {KERNEL}, dynamic instructions per locus and launch by basic-block counters (tools/bbcount.sh): {loci} loci of the bench data set,")
    import subprocess
    head = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    print(f"# one sweep after {cj['preroll']} iterations; variant-s sources at {head} + -gline-tables-only; {len(blocks)} blocks, {sum(1 for b in blocks if counts.get(b))} executed")
    print(f"\n## by class (per locus and sweep)\ntotal {per(allv):10.0f}")
    for c in CLASSES:
        print(f"  {c:14s} {per(tot[c]):10.0f}  {100.0 * tot[c] / allv:5.1f} %")
    valu = sum(tot[c] for c in CLASSES[:5])
    print(f"  (vector {per(valu):.0f}, scalar incl. branches {per(tot['salu'] + tot['branch']):.0f}, LDS {per(tot['lds']):.0f}, scalar memory {per(tot['smem']):.0f}: compare SQ_INSTS_* per wavefront in traffic_k_sweep.json)")

    def table(title, d, top):
        print(f"\n## {title}\n{'function':34s} {'total':>9s} {'share':>6s} | " + " ".join(f"{c:>12s}" for c in CLASSES))
        for f, c in sorted(d.items(), key=lambda kv: -sum(kv[1].values()))[:top]:
            s = sum(c.values())
            print(f"{f[:34]:34s} {per(s):9.0f} {100.0 * s / allv:5.1f}% | " + " ".join(f"{per(c[k]):12.0f}" for k in CLASSES))
    table("exclusive: instructions whose innermost source function is ...", excl, 45)
    table("inclusive: instructions with ... anywhere in their inline chain", incl, 45)
    print("\n## hottest basic blocks (dynamic instructions per locus and sweep; executions per locus; static size; source of the first instructions)")
    for w, b, n, k in sorted(hot, reverse=True)[:40]:
        ch = blocks[b][0][1]
        print(f"  BB1_{b:<5d} {per(w):8.0f}  x{per(n):8.1f}  {k:4d} instr  {' < '.join(ch[:4])}")


if __name__ == "__main__":
    main()
