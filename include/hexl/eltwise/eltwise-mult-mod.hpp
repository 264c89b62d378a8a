// hexl/eltwise/eltwise-mult-mod.hpp -- (a * b) mod q on the GPU.
// Drop-in for hexl/include/hexl/eltwise/eltwise-mult-mod.hpp:23-25.
#pragma once
#include <stdint.h>

namespace intel {
namespace hexl {

/// result[i] = (operand1[i] * operand2[i]) mod modulus in [0, modulus).
/// Inputs lie in [0, input_mod_factor * modulus), input_mod_factor in {1, 2, 4},
/// input_mod_factor * modulus < 2^63.
void EltwiseMultMod(uint64_t* result, const uint64_t* operand1, const uint64_t* operand2,
                    uint64_t n, uint64_t modulus, uint64_t input_mod_factor);

}  // namespace hexl
}  // namespace intel
