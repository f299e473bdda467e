#!/bin/bash
# PC-sampling profile of the benchmark workload (run through gpurun):  bash tools/pc_sample.sh <lib.so> [method] [interval]
# writes gpurun_out/pcs/*  (rocprofv3 --pc-sampling-beta-enabled; the program itself follows `--`: no wrapper exec)
cd /tmp && export TMPDIR=/tmp
lib=$1; method=${2:-stochastic}; interval=${3:-65536}
unit=cycles; [ "$method" = host_trap ] && unit=time
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pcs
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $method --pc-sampling-interval $interval \
  --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pcs -- python3 $GRAFT_REPO_ROOT/tools/bench_lib.py $lib 100000 2>&1 | tail -5
ls -la $GRAFT_REPO_ROOT/gpurun_out/pcs/*/* | head
