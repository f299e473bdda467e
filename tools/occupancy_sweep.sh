#!/bin/bash
# occupancy experiments: extra dynamic LDS per kernel class (GPH_LDS_PAD="class:bytes,...", classes: 0 sweep, 1 tau_eval,
# 2 mix_eval) on the benchmark workload after a pre-roll -> gpurun_out/occupancy.log
export TMPDIR=/tmp
: > gpurun_out/occupancy.log
for pad in "" "1:1500,2:1500" "1:4000,2:4000" "1:7000,2:7000" "" "0:1500" ; do
  GPH_LDS_PAD="$pad" timeout 600 python3 bench.py --steps 12 --warmup 3 --preroll 120 --no-cpu-baseline 2>/dev/null < /dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels']
print('pad [%s]: %.3f ms/step, sweep %.3f ms, tau_eval %.4f, mix_eval %.4f' % ('$pad', l['ms_per_step'], l['roofline']['avg_launch_ms'], k['tau_eval']['avg_ms'], k['mix_eval']['avg_ms']))" | tee -a gpurun_out/occupancy.log
done
