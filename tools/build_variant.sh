#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>: builds tools/libhexl_amd_<name>.so
# from the product sources with extra -D flags (developer A/B experiments, see
# HEXL_AMD_LIB in hexl_amd/__init__.py).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
# SRC_ROOT=<dir> builds another checkout's sources (e.g. a `git worktree` of an
# older commit) for same-box A/B timing.
ROOT=${SRC_ROOT:-$ROOT}
OUT=$(cd "$(dirname "$0")" && pwd)
NAME=$1; shift
T=$(mktemp -d)
for f in ntt_kernels.hip eltwise_kernels.hip keyswitch_kernels.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" \
    -Wno-unused-command-line-argument -I$ROOT/include -I$ROOT/hexl_amd/csrc \
    -c $ROOT/hexl_amd/csrc/$f -o $T/$f.o &
done
for f in capi.cpp number_theory.cpp workspace.cpp; do
  /opt/rocm/lib/llvm/bin/clang++ -x c++ -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/hexl_amd/csrc \
    -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -c $ROOT/hexl_amd/csrc/$f -o $T/$f.o &
done
wait || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libhexl_amd_$NAME.so $T/*.o
rm -rf $T
echo built tools/libhexl_amd_$NAME.so
