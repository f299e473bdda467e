#!/bin/bash
# experimental build of library variant s (the benchmark's) with extra flags:  tools/build_s_variant.sh name [flags ...] -> bench_cache/<name>.so
name=$1; shift
S="g-phocs_amd/csrc/gph_engine.hip g-phocs_amd/csrc/gph_mcmc.cpp g-phocs_amd/csrc/gph_input.cpp g-phocs_amd/csrc/gph_program.cpp g-phocs_amd/csrc/gph_readtrace.cpp g-phocs_amd/csrc/gph_comm.cpp"
exec hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-result -pthread -mllvm -disable-machine-licm -mllvm -structurizecfg-skip-uniform-regions \
  -DGPH_CAP_LEAVES=16 -DGPH_CAP_K=9 -DGPH_CAP_B=4 -DGPH_SWEEP_WAVES=8 "$@" $S -ldl -lrt -o bench_cache/$name.so
