#!/bin/bash
# A/B timing of alternative builds of the library on the benchmark workload AFTER a pre-roll (the chain has left the
# prior-sampled start: migration events, deeper trees), run through gpurun:
#   [LOCI=12500] [LOG=gpurun_out/ab.log] bash tools/ab.sh bench_cache/a.so bench_cache/b.so ...
# every library is timed twice, interleaved (box-to-box and run-to-run noise is ~0.5 %)
export TMPDIR=/tmp
PRE=${PRE:-120}
LOG=${LOG:-gpurun_out/ab.log}
: > $LOG
for rep in 1 2; do
  for lib in "$@"; do
    timeout 600 python3 bench.py --lib "$lib" --steps 12 --warmup 3 --preroll $PRE ${LOCI:+--loci $LOCI} --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels']
print('$lib: %.1f M evals/s, %.3f ms/step, sweep %.3f ms, tau_eval %.3f, mix_eval %.3f, launches %.0f' % (l['value']/1e6, l['ms_per_step'], l['roofline']['avg_launch_ms'], k['tau_eval']['avg_ms'], k['mix_eval']['avg_ms'], l['config']['kernel_launches_per_iteration']))" | tee -a $LOG
  done
done
