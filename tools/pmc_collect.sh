#!/bin/bash
# Counter passes for the benchmark workload on the GPU box (run through gpurun from the repo root):
#   bash tools/pmc_collect.sh <tag>      -> gpurun_out/pmc_<tag>/*.csv + gpurun_out/pmc_<tag>.json
# every pass runs the chain 200 iterations first (as the bench does) and sums the counters over ALL k_sweep dispatches of the
# process; per-dispatch and per-wave figures divide by the dispatch / wave counts of the same pass
# Each pass is its own rocprofv3 run with --pmc only (no tracing flags), the program directly after `--`.
# FETCH_SIZE and WRITE_SIZE get separate passes (MI355X_MICROARCH.md, HBM section).
set -u
TAG=${1:-run}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline > "$OUT/warm.json" 2> "$OUT/warm.err"   # fills bench_cache/
pass() { # name counters...
  local name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --preroll 200 --no-cpu-baseline > "$OUT/$name.log" 2>&1)
  echo "pass $name rc=$?"
}
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
pass vmem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
pass int SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 tools/pmc_summarize.py "$OUT" > "$ROOT/gpurun_out/pmc_$TAG.json"
tail -c 1500 "$ROOT/gpurun_out/pmc_$TAG.json"
