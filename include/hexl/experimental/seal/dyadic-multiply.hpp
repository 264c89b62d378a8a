// hexl/experimental/seal/dyadic-multiply.hpp -- RNS ciphertext product on the GPU.
// Drop-in for hexl/include/hexl/experimental/seal/dyadic-multiply.hpp:26-28.
#pragma once
#include <cstdint>

namespace intel {
namespace hexl {

/// (x[0], x[1]) * (y[0], y[1]) -> (x[0]*y[0], x[0]*y[1] + x[1]*y[0], x[1]*y[1]), every
/// polynomial n * num_moduli words in RNS form (modulus-major).  operand1/operand2 hold
/// two polynomials, result three; result may alias an operand.
void DyadicMultiply(uint64_t* result, const uint64_t* operand1, const uint64_t* operand2,
                    uint64_t n, const uint64_t* moduli, uint64_t num_moduli);

/// Additive extension (not in the reference): `num_pairs` ciphertext pairs with the same
/// moduli in one launch; operands hold num_pairs x 2 polynomials, result num_pairs x 3, all
/// DEVICE memory (hexl_amd_dyadic_multiply_batch).
void DyadicMultiplyBatch(uint64_t* result, const uint64_t* operand1, const uint64_t* operand2,
                         uint64_t num_pairs, uint64_t n, const uint64_t* moduli,
                         uint64_t num_moduli);

}  // namespace hexl
}  // namespace intel
