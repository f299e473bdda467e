#!/bin/bash
# device assembly of the engine (variant s by default):  tools/isa.sh out.s [extra -D flags]   (cross-compiles, no GPU)
out=$1; shift
src=${GPH_SRC:-g-phocs_amd/csrc/gph_engine.hip}
exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -mllvm -structurizecfg-skip-uniform-regions \
  -DGPH_CAP_LEAVES=16 -DGPH_CAP_K=9 -DGPH_CAP_B=4 -DGPH_SWEEP_WAVES=8 "$@" --cuda-device-only -S $src -o $out
