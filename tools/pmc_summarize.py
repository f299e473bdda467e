#!/usr/bin/env python3
"""Sum rocprofv3 --pmc csv output per kernel and counter: {kernel: {counter: {sum, dispatches}}}."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    seen = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            c = row["Counter_Name"]
            d = out.setdefault(k, {}).setdefault(c, {"sum": 0.0, "dispatches": 0})
            d["sum"] += float(row["Counter_Value"])
            key = (k, c, row["Dispatch_Id"])
            if key not in seen:
                seen[key] = 1
                d["dispatches"] += 1
print(json.dumps(out, indent=1, sort_keys=True))
