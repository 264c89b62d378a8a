// hexl/eltwise/eltwise-cmp-sub-mod.hpp -- conditional modular subtraction on the GPU.
// Drop-in for hexl/include/hexl/eltwise/eltwise-cmp-sub-mod.hpp:26-28.
#pragma once
#include <stdint.h>

#include "hexl/util/util.hpp"

namespace intel {
namespace hexl {

/// result[i] = cmp(operand1[i], bound) ? (operand1[i] mod modulus - diff) mod modulus
///                                     : operand1[i] mod modulus.
/// The comparison sees the unreduced word.  modulus > 1, 0 < diff < modulus.
void EltwiseCmpSubMod(uint64_t* result, const uint64_t* operand1, uint64_t n, uint64_t modulus,
                      CMPINT cmp, uint64_t bound, uint64_t diff);

}  // namespace hexl
}  // namespace intel
