#!/bin/bash
# Build a variant-`s` library whose DEVICE code goes through an assembly-level rewrite (experiments on the instruction stream
# the compiler cannot be talked into):  tools/asm/build_from_asm.sh out.so patch.py [extra -D flags]
#   1. device assembly of gph_engine.hip (same flags as g-phocs_amd/__init__.py, variant s)   2. patch.py in.s out.s
#   3. assemble + link the code object, bundle it   4. host object with that bundle embedded   5. link with the host sources
set -e
out=$1; patch=$2; shift 2
LL=/opt/rocm/lib/llvm/bin
W=${WORK:-/tmp/gph_asm_$$}
mkdir -p $W
C=g-phocs_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -pthread -w -mllvm -disable-machine-licm -mllvm -structurizecfg-skip-uniform-regions -DGPH_CAP_LEAVES=16 -DGPH_CAP_K=9 -DGPH_CAP_B=4 -DGPH_SWEEP_WAVES=8"
hipcc $FL "$@" --cuda-device-only -S $C/gph_engine.hip -o $W/dev.s
python3 $patch $W/dev.s $W/dev_p.s
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev_p.s -o $W/dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.out $W/dev.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
hipcc $FL "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c $C/gph_engine.hip -o $W/host.o
hipcc $FL "$@" -shared $W/host.o $C/gph_mcmc.cpp $C/gph_input.cpp $C/gph_program.cpp $C/gph_readtrace.cpp $C/gph_comm.cpp -ldl -lrt -o $out
[ -n "$WORK" ] || rm -rf $W
