#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_br; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
python3 bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline > "$OUT/warm.json" 2> "$OUT/warm.err"
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_NOT_TAKEN SQ_INSTS_CBRANCH_TAKEN SQ_WAVES SQ_INST_CYCLES_SALU SQ_INSTS_SALU -d "$OUT/i" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --preroll 0 --no-cpu-baseline > "$OUT/i.log" 2>&1)
tail -3 "$OUT/i.log"
python3 tools/pmc_summarize.py "$OUT" > "$ROOT/gpurun_out/pmc_br.json"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_br.json"))
k=[x for x in d if "k_sweep" in x][0]; s=d[k]; n=s["SQ_WAVES"]["sum"]
print({c: round(v["sum"]/n) for c,v in s.items()})
PY
