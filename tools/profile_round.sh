#!/bin/bash
# Everything the round's measurement section cites, in one GPU-box call (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>
#   gpurun_out/bench_<tag>.json          the bench line (with cpu_baseline), default flags = what the driver runs
#   gpurun_out/kstats_<tag>/             rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/pmc_<tag>.json            counter passes (tools/pmc_collect.sh; FETCH_SIZE and WRITE_SIZE in their own passes)
#   gpurun_out/bench_<tag>_12500.json    the per-GPU share of the 100k-locus set at 8 GPUs, on this one GPU (strong-scaling proxy)
set -u
TAG=${1:-run}
ROOT=$(pwd)
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?"
python3 bench.py --steps 20 --warmup 5 --loci 12500 --no-cpu-baseline > gpurun_out/bench_${TAG}_12500.json 2> gpurun_out/bench_${TAG}_12500.err
echo "bench 12500 rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/kstats_$TAG" -o k --output-format csv -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$ROOT/gpurun_out/kstats_$TAG.log" 2>&1)
echo "kernel-trace rc=$?"
find gpurun_out/kstats_$TAG -name "*kernel_trace.csv" -size +8M -delete   # keep the stats, drop the raw trace if it is huge
bash tools/pmc_collect.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
echo "pmc rc=$?"
ls -la gpurun_out/kstats_$TAG/* | head
