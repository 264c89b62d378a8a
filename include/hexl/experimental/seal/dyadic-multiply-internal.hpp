// hexl/experimental/seal/dyadic-multiply-internal.hpp -- intel::hexl::internal::DyadicMultiply.
// Drop-in for hexl/include/hexl/experimental/seal/dyadic-multiply-internal.hpp:12-14; the public
// DyadicMultiply (dyadic-multiply.hpp) is the same operation -- both reach hexl_amd_dyadic_multiply.
#pragma once

#include <cstdint>

namespace intel {
namespace hexl {
namespace internal {

void DyadicMultiply(uint64_t* result, const uint64_t* operand1, const uint64_t* operand2,
                    uint64_t n, const uint64_t* moduli, uint64_t num_moduli);

}  // namespace internal
}  // namespace hexl
}  // namespace intel
