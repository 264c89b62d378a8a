// number_theory.cpp -- host-side scalar number theory used to build NTT plans.
// Same results as the reference's helpers
// (hexl/number-theory/number-theory.cpp:13-261,
//  hexl/include/hexl/number-theory/number-theory.hpp:19-51), written for the
// plan builder: modular inverses by Fermat (the moduli are prime) except in
// the public inverse_mod, primitive roots by a deterministic search.
#include "number_theory.h"

namespace hexl_amd {
namespace nt {

typedef unsigned __int128 u128;

u64 multiply_factor(u64 operand, u64 bit_shift, u64 modulus) {
  // floor(operand * 2^bit_shift / modulus), low 64 bits
  // (number-theory.hpp:29-40; bit_shift in {32, 52, 64}; 63 is used for the
  // device tables of the Lazy arithmetic policy)
  u128 num = (u128)operand << bit_shift;
  return (u64)(num / modulus);
}

u64 multiply_mod(u64 x, u64 y, u64 modulus) {
  return (u64)(((u128)x * y) % modulus);
}

u64 pow_mod(u64 base, u64 exp, u64 modulus) {
  u64 acc = 1 % modulus;
  base %= modulus;
  for (; exp; exp >>= 1) {
    if (exp & 1) acc = multiply_mod(acc, base, modulus);
    base = multiply_mod(base, base, modulus);
  }
  return acc;
}

u64 inverse_mod(u64 x, u64 modulus) {
  // Extended Euclid; valid for any modulus coprime to x
  // (number-theory.cpp:13-42).  modulus == 1 -> 0.
  if (modulus == 1) return 0;
  __int128 r0 = modulus, r1 = x % modulus;
  __int128 t0 = 0, t1 = 1;
  while (r1 > 1) {
    __int128 k = r0 / r1;
    __int128 r2 = r0 - k * r1;
    __int128 t2 = t0 - k * t1;
    r0 = r1, r1 = r2;
    t0 = t1, t1 = t2;
  }
  if (t1 < 0) t1 += modulus;
  return (u64)t1;
}

bool is_power_of_two(u64 x) { return x && !(x & (x - 1)); }

u64 log2_floor(u64 x) { return 63 - __builtin_clzll(x); }

u64 reverse_bits(u64 x, u64 bit_width) {
  if (bit_width == 0) return 0;
  u64 r = 0;
  for (u64 i = 0; i < bit_width; ++i) r |= ((x >> i) & 1) << (bit_width - 1 - i);
  return r;
}

bool is_primitive_root(u64 root, u64 degree, u64 modulus) {
  // root^(degree/2) == -1 (number-theory.cpp:91-102)
  if (root == 0) return false;
  return pow_mod(root, degree / 2, modulus) == modulus - 1;
}

u64 generate_primitive_root(u64 degree, u64 modulus) {
  // The reference draws random candidates (number-theory.cpp:106-124); any
  // primitive degree-th root satisfies its contract.  Deterministic here.
  u64 cofactor = (modulus - 1) / degree;
  for (u64 g = 2; g < modulus && g < 100000; ++g) {
    u64 r = pow_mod(g, cofactor, modulus);
    if (is_primitive_root(r, degree, modulus)) return r;
  }
  return 0;
}

u64 minimal_primitive_root(u64 degree, u64 modulus) {
  // Smallest primitive degree-th root: the primitive roots are exactly the
  // odd powers of any one of them (number-theory.cpp:128-148).
  u64 g = generate_primitive_root(degree, modulus);
  if (g == 0) return 0;
  u64 g2 = multiply_mod(g, g, modulus);
  u64 best = g, cur = g;
  for (u64 i = 1; i < degree / 2 + 1; ++i) {
    cur = multiply_mod(cur, g2, modulus);
    if (cur < best) best = cur;
  }
  return best;
}

bool is_prime(u64 n) {
  // Deterministic Miller-Rabin for 64-bit n with the first 12 prime bases
  // (number-theory.cpp:166-212).
  static const u64 bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if (n < 2) return false;
  for (u64 a : bases) {
    if (n == a) return true;
    if (n % a == 0) return false;
  }
  u64 d = n - 1;
  int r = 0;
  while ((d & 1) == 0) d >>= 1, ++r;
  for (u64 a : bases) {
    u64 x = pow_mod(a, d, n);
    if (x == 1 || x == n - 1) continue;
    bool witness = true;
    for (int i = 1; i < r; ++i) {
      x = multiply_mod(x, x, n);
      if (x == n - 1) {
        witness = false;
        break;
      }
    }
    if (witness) return false;
  }
  return true;
}

size_t generate_primes(u64* out, size_t num_primes, size_t bit_size,
                       bool prefer_small, size_t ntt_size) {
  // primes == 1 (mod 2*ntt_size) in (2^bit_size, 2^(bit_size+1)), scanned
  // upward from the bottom or downward from the top
  // (number-theory.cpp:214-261).
  const int64_t lo = (int64_t(1) << bit_size) + 1;
  const int64_t hi = (int64_t(1) << (bit_size + 1)) - 1;
  const int64_t step = 2 * (int64_t)ntt_size;
  size_t found = 0;
  if (prefer_small) {
    for (int64_t c = lo; c < hi && found < num_primes; c += step)
      if (is_prime((u64)c)) out[found++] = (u64)c;
  } else {
    for (int64_t c = hi - (hi % step) + 1; c > lo && found < num_primes; c -= step)
      if (is_prime((u64)c)) out[found++] = (u64)c;
  }
  return found;
}

bool ntt_check_arguments(u64 degree, u64 modulus) {
  // NTT::CheckArguments (hexl/ntt/ntt-internal.cpp:171-186)
  if (!is_power_of_two(degree)) return false;
  if (degree < 2 || degree > (u64(1) << 20)) return false;
  if (modulus > (u64(1) << 62)) return false;
  if (modulus % (2 * degree) != 1) return false;
  return is_prime(modulus);
}

}  // namespace nt
}  // namespace hexl_amd
