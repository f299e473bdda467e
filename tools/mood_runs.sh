#!/bin/bash
# the evaluate kernels are bimodal between PROCESSES on one box (k_tau_eval 0.41 / 0.43 ms, k_mix_eval 0.51 / 0.55 ms): N runs of one
# library, one line each.   [ENVS="A=1 B=2"] bash tools/mood_runs.sh lib.so [runs]   (through gpurun)
lib=$1; n=${2:-8}
for i in $(seq 1 $n); do
  env $ENVS python3 bench.py --lib "$lib" --steps 8 --warmup 2 --preroll 60 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels']
print('$ENVS run $i: %.3f ms/step, sweep %.3f, tau_eval %.3f, mix_eval %.3f' % (l['ms_per_step'], l['roofline']['avg_launch_ms'], k['tau_eval']['avg_ms'], k['mix_eval']['avg_ms']))"
done
