// hexl/eltwise/eltwise-sub-mod.hpp -- (a - b) mod q on the GPU.
// Drop-in for hexl/include/hexl/eltwise/eltwise-sub-mod.hpp:22-37.
#pragma once
#include <stdint.h>

namespace intel {
namespace hexl {

/// result[i] = (operand1[i] - operand2[i]) mod modulus; inputs below modulus < 2^63.
void EltwiseSubMod(uint64_t* result, const uint64_t* operand1, const uint64_t* operand2,
                   uint64_t n, uint64_t modulus);

/// result[i] = (operand1[i] - operand2) mod modulus with a scalar operand2 < modulus.
void EltwiseSubMod(uint64_t* result, const uint64_t* operand1, uint64_t operand2, uint64_t n,
                   uint64_t modulus);

}  // namespace hexl
}  // namespace intel
