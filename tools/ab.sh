#!/bin/bash
# A/B timing of alternative builds of the library on the benchmark workload (run through gpurun):
#   bash tools/ab.sh bench_cache/a.so bench_cache/b.so ...     -> gpurun_out/ab.log
# every library is timed twice, interleaved (box-to-box and run-to-run noise is ~0.5 %)
export TMPDIR=/tmp
: > gpurun_out/ab.log
for rep in 1 2; do
  for lib in "$@"; do
    timeout 600 python3 tools/bench_lib.py "$lib" 2>&1 | tail -1 | tee -a gpurun_out/ab.log
  done
done
