// hexl/eltwise/eltwise-cmp-add.hpp -- conditional addition on the GPU.
// Drop-in for hexl/include/hexl/eltwise/eltwise-cmp-add.hpp:24-25.
#pragma once
#include <stdint.h>

#include "hexl/util/util.hpp"

namespace intel {
namespace hexl {

/// result[i] = cmp(operand1[i], bound) ? operand1[i] + diff : operand1[i]
/// (plain 64-bit addition).  n != 0, diff != 0.
void EltwiseCmpAdd(uint64_t* result, const uint64_t* operand1, uint64_t n, CMPINT cmp,
                   uint64_t bound, uint64_t diff);

}  // namespace hexl
}  // namespace intel
