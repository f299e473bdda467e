#!/usr/bin/env python3
"""Decision-level fixtures: the real reference compiled with -DLOG_STEPS (oracle/_ref/gphocs_ref_log,
oracle/Makefile) writes a per-proposal transcript G-PhoCS-debug.txt -- old --> new value, considerEventMove's event
ids, lnacceptance, accepting / rejecting (GPhoCS.c:2363-2401, 2540-2577, 2654-2718; patch.c:1451-1454).  Kept: the
first 200 proposals of the three genealogy sweeps for two loci per case, verbatim.
    python3 make_logsteps.py        (from tests/golden, where /root/reference exists)  ->  <case>.logsteps"""
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref", "gphocs_ref_log")
CASES = {"m3": (3, 11), "a7": (2, 9), "j1": (2, 7)}      # case: the two loci (global indices, "gen" upstream)
KEEP = 200

for name, loci in CASES.items():
    with tempfile.TemporaryDirectory() as td:
        for ext in (".ctl", ".seq"):
            shutil.copy(os.path.join(HERE, name + ext), td)
        subprocess.run([REF, "main", "-n", "1", name + ".ctl"], cwd=td, check=True, capture_output=True, timeout=900)
        lines = open(os.path.join(td, "G-PhoCS-debug.txt")).read().splitlines()
    kept = {g: [] for g in loci}
    cur, cur_gen = None, None
    for ln in lines:
        if ln.startswith("  gen "):
            cur_gen = int(ln.split(",")[0].split()[1])
            cur = [ln]
        elif cur is not None:
            cur.append(ln)
        if cur is not None and (ln.endswith("accepting.") or ln.endswith("rejecting.")):
            if cur_gen in kept and len(kept[cur_gen]) < KEEP:
                kept[cur_gen].append("\n".join(cur))
            cur = None
    with open(os.path.join(HERE, name + ".logsteps"), "w") as f:
        for g in loci:
            assert len(kept[g]) == KEEP, (name, g, len(kept[g]))
            f.write(f"# locus {g}: its first {KEEP} proposals (UpdateGB_InternalNode / _MigrationNode / _MigSPR), as upstream's LOG_STEPS build printed them\n")
            for p in kept[g]:
                f.write(p + "\n")
    print(name, {g: len(v) for g, v in kept.items()})
