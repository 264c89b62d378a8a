// hexl/eltwise/eltwise-fma-mod.hpp -- (a * s + c) mod q on the GPU.
// Drop-in for hexl/include/hexl/eltwise/eltwise-fma-mod.hpp:22-24.
#pragma once
#include <stdint.h>

namespace intel {
namespace hexl {

/// result[i] = (arg1[i] * arg2 + arg3[i]) mod modulus in [0, modulus); arg3 may be
/// nullptr (no addend).  arg1, arg2, arg3 lie in [0, input_mod_factor * modulus),
/// input_mod_factor in {1, 2, 4, 8}, modulus < 2^61.
void EltwiseFMAMod(uint64_t* result, const uint64_t* arg1, uint64_t arg2, const uint64_t* arg3,
                   uint64_t n, uint64_t modulus, uint64_t input_mod_factor);

}  // namespace hexl
}  // namespace intel
