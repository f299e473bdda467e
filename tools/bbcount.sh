#!/bin/bash
# Dynamic instruction profile of k_sweep by basic block (no PC sampling on this pool, no thread-trace decoder in the image):
#   1. here (build container):  bash tools/bbcount.sh build     -> bench_cache/bbcount.so + bench_cache/bbcount_dev.s.gz
#        the variant-s library with a counter per basic block patched into the kernel's assembly (tools/asm/patch_bbcount.py),
#        compiled with line tables so that every instruction carries its source line and inline chain
#   2. on the GPU:  gpurun -- 'python3 tools/bbcount_run.py'     -> gpurun_out/bbcount.json  (executions per block, one sweep)
#   3. here:  python3 tools/bbcount_report.py                    -> profiles/rNN_bbcount.txt  (instructions per locus and sweep by
#        source function / by class, the hottest blocks)
set -e
if [ "$1" = build ]; then
  rm -rf /tmp/gph_bb
  # [BB_KERNEL=<mangled name of another kernel whose second argument is GphDev>, with BB_WHAT=iteration BB_LAUNCHES=<launches per iteration> for steps 2 and 3]
  WORK=/tmp/gph_bb bash tools/asm/build_from_asm.sh bench_cache/bbcount.so tools/asm/patch_bbcount.py -DGPH_BBCOUNT -gline-tables-only
  gzip -c /tmp/gph_bb/dev.s > bench_cache/bbcount_dev.s.gz
  ls -la bench_cache/bbcount.so bench_cache/bbcount_dev.s.gz
fi
