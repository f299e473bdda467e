#!/bin/bash
# quick instruction-mix pass: gpurun_out/pmc_insts.json
ROOT0=$(pwd)
set -u
LIBARG=${1:+--lib $ROOT0/$1}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_insts; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
python3 bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline $LIBARG > "$OUT/warm.json" 2> "$OUT/warm.err"
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES -d "$OUT/i" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --preroll 0 --no-cpu-baseline $LIBARG > "$OUT/i.log" 2>&1)
python3 tools/pmc_summarize.py "$OUT" > "$ROOT/gpurun_out/pmc_insts.json"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_insts.json"))
k=[x for x in d if "k_sweep" in x][0]; s=d[k]; n=s["SQ_WAVES"]["sum"]
print({c: round(v["sum"]/n) for c,v in s.items()})
PY
