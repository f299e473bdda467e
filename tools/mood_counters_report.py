#!/usr/bin/env python3
"""per process of tools/mood_counters.sh: mean duration and counters per dispatch of the evaluate kernels, then -- per counter --
the correlation of the per-process means with the per-process mean duration"""
import csv
import glob
import os
import sys

import numpy as np

root = sys.argv[1]
rows = []
for d in sorted(glob.glob(os.path.join(root, "[ABCD]*"))):
    if not os.path.isdir(d):
        continue
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"].split("(")[0]
                if k not in ("k_tau_eval", "k_mix_eval", "k_sweep"):
                    continue
                e = per.setdefault(k, {}).setdefault(r["Dispatch_Id"], {"dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "c": {}})
                e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for k, disp in per.items():
            ds = list(disp.values())[len(disp) // 4:]          # drop the first quarter (start-up of the chain)
            names = sorted(ds[0]["c"])
            rows.append(dict(run=os.path.basename(d), kernel=k, n=len(ds), dur_us=np.mean([x["dur"] for x in ds]) / 1e3,
                             **{n: np.mean([x["c"].get(n, 0.0) for x in ds]) for n in names}))
for k in ("k_tau_eval", "k_mix_eval", "k_sweep"):
    rk = [r for r in rows if r["kernel"] == k]
    if not rk:
        continue
    print(f"## {k}: per process (under --pmc: dispatches are serialised and a little slower than in a plain run)")
    for s in "ABCD":
        rs = [r for r in rk if r["run"].startswith(s)]
        if not rs:
            continue
        names = [n for n in rs[0] if n not in ("run", "kernel", "n", "dur_us")]
        print("run   dispatches   mean us   " + "   ".join(f"{n:>24s}" for n in names))
        for r in sorted(rs, key=lambda r: r["dur_us"]):
            print(f"{r['run']:4s}  {r['n']:10d}  {r['dur_us']:8.1f}   " + "   ".join(f"{r[n]:24.1f}" for n in names))
        d = np.array([r["dur_us"] for r in rs])
        if len(rs) >= 3 and d.std() > 0:
            for n in names:
                v = np.array([r[n] for r in rs])
                cc = np.corrcoef(d, v)[0, 1] if v.std() > 0 else float("nan")
                if n == "GRBM_GUI_ACTIVE":      # summed over the 8 XCDs: the effective shader clock of the dispatch (MI355X_MICROARCH.md, DVFS)
                    for r in sorted(rs, key=lambda r: r["dur_us"]):
                        print(f"   {r['run']}: effective clock {r[n] / 8 / r['dur_us'] / 1e3:.3f} GHz")
                print(f"   corr(duration, {n}) = {cc:+.2f}   spread of the counter {100 * (v.max() - v.min()) / max(v.mean(), 1e-30):.2f} %   spread of the duration {100 * (d.max() - d.min()) / d.mean():.2f} %")
