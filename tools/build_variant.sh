#!/bin/bash
# build an experimental variant of the `s` library into bench_cache/<name>.so:  tools/build_variant.sh name [-DFLAG ...]
name=$1; shift
S="g-phocs_amd/csrc/gph_engine.hip g-phocs_amd/csrc/gph_mcmc.cpp g-phocs_amd/csrc/gph_input.cpp g-phocs_amd/csrc/gph_program.cpp g-phocs_amd/csrc/gph_readtrace.cpp g-phocs_amd/csrc/gph_comm.cpp"
W=6
for a in "$@"; do case $a in -DGPH_SWEEP_WAVES=*) W=;; esac; done
exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-result -pthread -mllvm -disable-machine-licm -mllvm -structurizecfg-skip-uniform-regions -mllvm -amdgpu-sched-strategy=max-ilp \
  -DGPH_CAP_LEAVES=16 -DGPH_CAP_K=9 -DGPH_CAP_B=4 ${W:+-DGPH_SWEEP_WAVES=$W} "$@" $S -ldl -lrt -o bench_cache/$name.so
