#!/bin/bash
# the evaluate kernels have "moods" between PROCESSES on one box (k_tau_eval 0.41 / 0.43 ms with one
# library).  N processes per counter set, each under `rocprofv3 --pmc` (nothing but --pmc; the program right after `--`): per
# process the mean duration of the k_tau_eval / k_mix_eval dispatches (the CSV's own timestamps) next to its L2 / fabric counters,
# so that a slow process can be told from a fast one by something other than its time.
#   bash tools/mood_counters.sh [runs per set, default 6]     (through gpurun)   ->   gpurun_out/mood_counters.txt
set -u
N=${1:-6}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/mood
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline > "$OUT/warm.json" 2> "$OUT/warm.err"
(cd /tmp && rocprofv3 -L > "$OUT/counters.txt" 2>&1)
have() { grep -q -w "$1" "$OUT/counters.txt"; }
pick() { local o=""; for c in "$@"; do if have "$c"; then o="$o $c"; fi; done; echo $o; }
SETA=$(pick TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum)
SETB=$(pick TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum)
SETC=$(pick TCC_BUSY_sum TCC_REQ_sum TCC_WRITEBACK_sum TCC_EA0_RD_UNCACHED_32B_sum)
SETD=$(pick GRBM_GUI_ACTIVE GRBM_COUNT)
echo "sets: A=[$SETA] B=[$SETB] C=[$SETC] D=[$SETD]"
for S in ${SETS:-A B C D}; do
  eval CS=\$SET$S
  [ -z "$CS" ] && continue
  for i in $(seq 1 $N); do
    (cd /tmp && timeout 600 rocprofv3 --pmc $CS -d "$OUT/$S$i" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --preroll 40 --no-cpu-baseline > "$OUT/$S$i.log" 2>&1)
    echo "set $S run $i rc=$?"
  done
done
python3 tools/mood_counters_report.py "$OUT" | tee "$ROOT/gpurun_out/mood_counters.txt"
