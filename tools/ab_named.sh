#!/bin/bash
# like tools/ab.sh, with environment per run:  bash tools/ab_named.sh "ENV=.. lib.so" "lib2.so" ...
export TMPDIR=/tmp
PRE=${PRE:-120}
for rep in 1 2; do
  for spec in "$@"; do
    lib=${spec##* }; envs=${spec% *}; [ "$envs" = "$spec" ] && envs=""
    env $envs timeout 600 python3 bench.py --lib "$lib" --steps 12 --warmup 3 --preroll $PRE ${LOCI:+--loci $LOCI} --no-cpu-baseline 2>/dev/null < /dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels']
print('$spec: %.1f M evals/s, %.3f ms/step, sweep %.3f ms, tau_eval %.3f, mix_eval %.3f, accept %s' % (l['value']/1e6, l['ms_per_step'], l['roofline']['avg_launch_ms'], k['tau_eval']['avg_ms'], k['mix_eval']['avg_ms'], l['config']['accept_counts_timed'][:3]))"
  done
done
