#!/bin/bash
# BASELINE configs[4] at its stated size (200 k loci, 20 leaves, 13 populations, 4 bands, fixed ancient sample; library
# variant l) on one MI355X: bench line, rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes (run through gpurun):
#   bash tools/profile_config5.sh  ->  gpurun_out/config5_*.{json,csv}
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp
# every pass is ONE run into an EMPTY directory (the r02 summary merged two runs)
rm -rf $OUT/config5_kstats $OUT/config5_pmc
python3 tools/bench_config5.py > $OUT/config5_bench.json 2> $OUT/config5_bench.err
echo "bench rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/config5_kstats -o k --output-format csv -- python3 $ROOT/tools/bench_config5.py > $OUT/config5_kstats.log 2>&1)
echo "kstats rc=$?"
# the two dispatches of a launch point overlap (side stream, gph_engine.hip): per sweep, first start to last end
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/config5_kstats/*kernel_trace.csv")
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if r["Kernel_Name"].startswith("k_sweep")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    walls, sums = [], []
    for i in range(0, len(rows) - 1, 2):
        a, b = rows[i], rows[i + 1]
        walls.append(max(int(a["End_Timestamp"]), int(b["End_Timestamp"])) - int(a["Start_Timestamp"]))
        sums.append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]) + int(b["End_Timestamp"]) - int(b["Start_Timestamp"]))
    json.dump({"sweeps": len(walls), "k_sweep_wall_ms_per_sweep": sum(walls) / len(walls) / 1e6,
               "k_sweep_summed_dispatch_ms_per_sweep": sum(sums) / len(sums) / 1e6}, open("gpurun_out/config5_sweep_wall.json", "w"))
PY
find $OUT/config5_kstats -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 900 rocprofv3 --pmc $c -d $OUT/config5_pmc/$c -o p --output-format csv -- python3 $ROOT/tools/bench_config5.py > $OUT/config5_pmc_$c.log 2>&1)
  echo "$c rc=$?"
done
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES -d $OUT/config5_pmc/insts -o p --output-format csv -- python3 $ROOT/tools/bench_config5.py > $OUT/config5_pmc_insts.log 2>&1)
python3 tools/pmc_summarize.py $OUT/config5_pmc > $OUT/config5_pmc.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/config5_pmc.json"))
for k, s in d.items():
    if "FETCH_SIZE" in s and ("k_sweep" in k or "tau_eval" in k):
        f = s["FETCH_SIZE"]["sum"] * 2048 / s["FETCH_SIZE"]["dispatches"]; w = s["WRITE_SIZE"]["sum"] * 1024 / s["WRITE_SIZE"]["dispatches"]
        print(k, "fetch GB %.2f write GB %.2f per launch" % (f / 1e9, w / 1e9), {c: round(v["sum"] / s["SQ_WAVES"]["sum"]) for c, v in s.items() if c.startswith("SQ_")})
PY
