#!/bin/bash
# instruction-cache counters of the benchmark run (one extra rocprofv3 --pmc pass): gpurun_out/pmc_icache.json
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_icache; mkdir -p "$OUT"; export TMPDIR=/tmp
python3 bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline > "$OUT/warm.json" 2> "$OUT/warm.err"
(cd /tmp && timeout 900 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH -d "$OUT/ic" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --preroll 0 --no-cpu-baseline > "$OUT/ic.log" 2>&1)
echo "rc=$?"; tail -3 "$OUT/ic.log"
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_IFETCH_LEVEL -d "$OUT/ic2" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --preroll 0 --no-cpu-baseline > "$OUT/ic2.log" 2>&1)
echo "rc2=$?"
python3 tools/pmc_summarize.py "$OUT" > "$ROOT/gpurun_out/pmc_icache.json"
