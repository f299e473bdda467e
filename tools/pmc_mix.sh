#!/bin/bash
# dynamic VALU mix of k_sweep: two counter passes
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_mix; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
python3 bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline > $OUT/warm.json 2> $OUT/warm.err
pass() { local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --preroll 60 --no-cpu-baseline > "$OUT/$name.log" 2>&1); echo "pass $name rc=$?"; }
pass f64 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES
pass int SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVES
python3 tools/pmc_summarize.py $OUT > $ROOT/gpurun_out/pmc_mix.json
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_mix.json"))
for k,s in d.items():
    if "k_sweep" in k or "tau_eval" in k:
        n=s["SQ_WAVES"]["sum"]/ (2 if "SQ_WAVES" in s else 1)
        print(k, {c: round(v["sum"]/ (s["SQ_WAVES"]["sum"]/2)) for c,v in s.items()})
PY
