#!/bin/bash
# Everything the round's measurement section cites, in one GPU-box call (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>
#   gpurun_out/bench_<tag>.json          the bench line (with cpu_baseline)
#   gpurun_out/kstats_<tag>/             rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/pmc_<tag>.json            counter passes (tools/pmc_collect.sh)
set -u
TAG=${1:-run}
ROOT=$(pwd)
export TMPDIR=/tmp
python3 bench.py --steps 10 --warmup 3 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/kstats_$TAG" -o k --output-format csv -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$ROOT/gpurun_out/kstats_$TAG.log" 2>&1)
echo "kernel-trace rc=$?"
find gpurun_out/kstats_$TAG -name "*kernel_trace.csv" -size +8M -delete   # keep the stats, drop the raw trace if it is huge
bash tools/pmc_collect.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
echo "pmc rc=$?"
ls -la gpurun_out/kstats_$TAG/* | head
