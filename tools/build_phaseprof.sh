#!/bin/bash
# Builds tools/libhexl_amd_phaseprof.so: the product sources with
# -DHEXL_AMD_PHASE_PROFILE (per-phase s_memtime stamps in block_pass).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
for f in ntt_kernels.hip eltwise_kernels.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DHEXL_AMD_PHASE_PROFILE \
    -Wno-unused-command-line-argument -I$ROOT/include -I$ROOT/hexl_amd/csrc \
    -c $ROOT/hexl_amd/csrc/$f -o $T/$f.o
done
for f in capi.cpp number_theory.cpp; do
  /opt/rocm/lib/llvm/bin/clang++ -x c++ -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/hexl_amd/csrc \
    -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -c $ROOT/hexl_amd/csrc/$f -o $T/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/libhexl_amd_phaseprof.so $T/*.o
rm -rf $T
