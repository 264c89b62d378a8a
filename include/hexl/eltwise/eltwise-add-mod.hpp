// hexl/eltwise/eltwise-add-mod.hpp -- (a + b) mod q on the GPU.
// Drop-in for hexl/include/hexl/eltwise/eltwise-add-mod.hpp:22-37.
#pragma once
#include <stdint.h>

namespace intel {
namespace hexl {

/// result[i] = (operand1[i] + operand2[i]) mod modulus, i < n.  Inputs must be
/// below modulus, modulus in (1, 2^63).  Pointers may be host or device memory
/// (detected per call); result may alias an operand.
void EltwiseAddMod(uint64_t* result, const uint64_t* operand1, const uint64_t* operand2,
                   uint64_t n, uint64_t modulus);

/// result[i] = (operand1[i] + operand2) mod modulus with a scalar operand2 < modulus.
void EltwiseAddMod(uint64_t* result, const uint64_t* operand1, uint64_t operand2, uint64_t n,
                   uint64_t modulus);

}  // namespace hexl
}  // namespace intel
