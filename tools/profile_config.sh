#!/bin/bash
# One BASELINE config other than the bench default at its stated size on one MI355X (run through gpurun from the repo root):
#   bash tools/profile_config.sh <synthetic config> <loci> <tag>
#     2 10000 config2   = BASELINE configs[1]: 10k loci x 8 leaves, 3-pop tree, no migration
#     3 40000 config3   = BASELINE configs[2]: 40k loci x 12 leaves, 3-pop tree + 2 bands
#   -> gpurun_out/<tag>_bench.json, <tag>_kstats/ (rocprofv3 --kernel-trace --stats), <tag>_pmc.json (separate --pmc passes)
# Every pass is ONE run of tools/bench_config5.py into an empty directory, the program directly after `--`.
set -u
CFG=$1; L=$2; TAG=$3
ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp
export GPH_BENCH_CONFIG=$CFG
rm -rf $OUT/${TAG}_kstats $OUT/${TAG}_pmc
python3 tools/bench_config5.py - $L > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kstats -o k --output-format csv -- python3 $ROOT/tools/bench_config5.py - $L > $OUT/${TAG}_kstats.log 2>&1)
echo "kstats rc=$?"
find $OUT/${TAG}_kstats -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c -d $OUT/${TAG}_pmc/$c -o p --output-format csv -- python3 $ROOT/tools/bench_config5.py - $L > $OUT/${TAG}_pmc_$c.log 2>&1)
  echo "$c rc=$?"
done
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES -d $OUT/${TAG}_pmc/insts -o p --output-format csv -- python3 $ROOT/tools/bench_config5.py - $L > $OUT/${TAG}_pmc_insts.log 2>&1)
python3 tools/pmc_summarize.py $OUT/${TAG}_pmc > $OUT/${TAG}_pmc.json
python3 - $TAG <<'PY'
import json, sys
tag = sys.argv[1]
d = json.load(open(f"gpurun_out/{tag}_pmc.json"))
b = json.load(open(f"gpurun_out/{tag}_bench.json"))
out = {}
for k, s in d.items():
    if "FETCH_SIZE" in s and ("k_sweep" in k or "tau_eval" in k or "mix_eval" in k):
        # MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 32-byte units x 2 on gfx950 -> x 2048 / 1024-byte... as in profile_config5.sh
        f = s["FETCH_SIZE"]["sum"] * 2048 / s["FETCH_SIZE"]["dispatches"]; w = s["WRITE_SIZE"]["sum"] * 1024 / s["WRITE_SIZE"]["dispatches"]
        out[k.split("(")[0]] = {"fetch_GB_per_launch": f / 1e9, "write_GB_per_launch": w / 1e9,
                                "per_wave": {c: round(v["sum"] / s["SQ_WAVES"]["sum"]) for c, v in s.items() if c.startswith("SQ_") and "SQ_WAVES" in s}}
sw = next((v for k, v in out.items() if "k_sweep" in k), None)
if sw:
    b["sweep_hbm_counter_bytes"] = (sw["fetch_GB_per_launch"] + sw["write_GB_per_launch"]) * 1e9
    b["sweep_hbm_counter_frac"] = b["sweep_hbm_counter_bytes"] / (b["sweep_ms"] * 1e-3) / 8e12
b["counters"] = out
json.dump(b, open(f"gpurun_out/{tag}_summary.json", "w"), indent=1)
print(json.dumps(b)[:1500])
PY
