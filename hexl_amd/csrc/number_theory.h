// number_theory.h -- host scalar number theory for the plan builder.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace hexl_amd {
namespace nt {
typedef uint64_t u64;

u64 multiply_factor(u64 operand, u64 bit_shift, u64 modulus);
u64 multiply_mod(u64 x, u64 y, u64 modulus);
u64 pow_mod(u64 base, u64 exp, u64 modulus);
u64 inverse_mod(u64 x, u64 modulus);
bool is_power_of_two(u64 x);
u64 log2_floor(u64 x);
u64 reverse_bits(u64 x, u64 bit_width);
bool is_primitive_root(u64 root, u64 degree, u64 modulus);
u64 generate_primitive_root(u64 degree, u64 modulus);
u64 minimal_primitive_root(u64 degree, u64 modulus);
bool is_prime(u64 n);
size_t generate_primes(u64* out, size_t num_primes, size_t bit_size,
                       bool prefer_small, size_t ntt_size);
bool ntt_check_arguments(u64 degree, u64 modulus);

}  // namespace nt
}  // namespace hexl_amd
