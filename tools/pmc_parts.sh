#!/bin/bash
# VALU / SALU / LDS instructions per wavefront of the sweep kernel by proposal class (tools/sweep_parts.py under --pmc):
#   gpurun -- bash tools/pmc_parts.sh      -> gpurun_out/pmc_parts.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_parts; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
python3 tools/sweep_parts.py > "$OUT/warm.log" 2>&1
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d "$OUT/i" -o p --output-format csv -- python3 "$ROOT/tools/sweep_parts.py" > "$OUT/i.log" 2>&1)
python3 - "$OUT" <<'PY' | tee "$ROOT/gpurun_out/pmc_parts.txt"
import csv, glob, sys, collections
rows = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/i/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sweep" not in r["Kernel_Name"]: continue
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ds = sorted(rows)
print("k_sweep dispatches (last 7 = flags 0,0,1,2,4,7,7):")
for d in ds[-7:]:
    c = rows[d]; n = c.get("SQ_WAVES", 1) or 1
    print(d, {k: round(v / n) for k, v in c.items() if k != "SQ_WAVES"}, "waves", int(n))
PY
