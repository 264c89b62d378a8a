// hexl/eltwise/eltwise-reduce-mod.hpp -- range reduction on the GPU.
// Drop-in for hexl/include/hexl/eltwise/eltwise-reduce-mod.hpp:24-26.
#pragma once
#include <stdint.h>

namespace intel {
namespace hexl {

/// result[i] = operand[i] reduced from [0, input_mod_factor * modulus) to
/// [0, output_mod_factor * modulus).  input_mod_factor in {modulus, 2, 4}
/// (== modulus means "any 64-bit word"), output_mod_factor in {1, 2}.
void EltwiseReduceMod(uint64_t* result, const uint64_t* operand, uint64_t n, uint64_t modulus,
                      uint64_t input_mod_factor, uint64_t output_mod_factor);

}  // namespace hexl
}  // namespace intel
