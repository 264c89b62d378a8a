// hexl/number-theory/number-theory.hpp -- scalar modular arithmetic of the
// public API (reference: hexl/include/hexl/number-theory/number-theory.hpp:19-339,
// hexl/number-theory/number-theory.cpp:13-261).  The out-of-line functions are
// thin wrappers over the C-ABI of include/hexl_amd.h (host code, no GPU needed).
#pragma once
#include <stdint.h>

#include <limits>
#include <vector>

#include "hexl/util/check.hpp"
#include "hexl/util/compiler.hpp"

namespace intel {
namespace hexl {

/// Shoup / Barrett factor floor(operand * 2^bit_shift / modulus) for repeated
/// multiplication by `operand` (bit_shift in {32, 52, 64}).
class MultiplyFactor {
 public:
  MultiplyFactor() = default;
  MultiplyFactor(uint64_t operand, uint64_t bit_shift, uint64_t modulus) : m_operand(operand) {
    HEXL_CHECK(operand <= modulus, "operand " << operand << " must be less than modulus");
    HEXL_CHECK(bit_shift == 32 || bit_shift == 52 || bit_shift == 64,
               "Unsupported BitShift " << bit_shift);
    const uint128_t num = static_cast<uint128_t>(operand) << bit_shift;
    m_barrett_factor = static_cast<uint64_t>(num / modulus);
  }
  inline uint64_t BarrettFactor() const { return m_barrett_factor; }
  inline uint64_t Operand() const { return m_operand; }

 private:
  uint64_t m_operand;
  uint64_t m_barrett_factor;
};

inline bool IsPowerOfTwo(uint64_t num) { return num && !(num & (num - 1)); }
/// floor(log2(x))
inline uint64_t Log2(uint64_t x) { return MSB(x); }
inline bool IsPowerOfFour(uint64_t num) { return IsPowerOfTwo(num) && (Log2(num) % 2 == 0); }
/// Largest value representable in `bits` bits.
inline uint64_t MaximumValue(uint64_t bits) {
  HEXL_CHECK(bits <= 64, "MaximumValue requires bits <= 64; got " << bits);
  return bits == 64 ? (std::numeric_limits<uint64_t>::max)() : (1ULL << bits) - 1;
}

uint64_t ReverseBits(uint64_t x, uint64_t bit_width);
uint64_t InverseMod(uint64_t x, uint64_t modulus);
uint64_t MultiplyMod(uint64_t x, uint64_t y, uint64_t modulus);
uint64_t MultiplyMod(uint64_t x, uint64_t y, uint64_t y_precon, uint64_t modulus);
uint64_t AddUIntMod(uint64_t x, uint64_t y, uint64_t modulus);
uint64_t SubUIntMod(uint64_t x, uint64_t y, uint64_t modulus);
uint64_t PowMod(uint64_t base, uint64_t exp, uint64_t modulus);
bool IsPrimitiveRoot(uint64_t root, uint64_t degree, uint64_t modulus);
uint64_t GeneratePrimitiveRoot(uint64_t degree, uint64_t modulus);
uint64_t MinimalPrimitiveRoot(uint64_t degree, uint64_t modulus);
bool IsPrime(uint64_t n);
std::vector<uint64_t> GeneratePrimes(size_t num_primes, size_t bit_size,
                                     bool prefer_small_primes, size_t ntt_size = 1);

/// x * y mod modulus up to one extra modulus: result in [0, 2 * modulus).
template <int BitShift>
inline uint64_t MultiplyModLazy(uint64_t x, uint64_t y_operand, uint64_t y_barrett_factor,
                                uint64_t modulus) {
  HEXL_CHECK(y_operand < modulus, "y_operand must be less than modulus");
  HEXL_CHECK(modulus <= MaximumValue(BitShift), "Modulus exceeds bound");
  HEXL_CHECK(x <= MaximumValue(BitShift), "Operand exceeds bound");
  const uint64_t Q = MultiplyUInt64Hi<BitShift>(x, y_barrett_factor);
  return y_operand * x - Q * modulus;
}

template <int BitShift>
inline uint64_t MultiplyModLazy(uint64_t x, uint64_t y, uint64_t modulus) {
  HEXL_CHECK(BitShift == 64 || BitShift == 52, "Unsupported BitShift " << BitShift);
  return MultiplyModLazy<BitShift>(x, y, MultiplyFactor(y, BitShift, modulus).BarrettFactor(),
                                   modulus);
}

/// Sum and carry-out of two 64-bit words.
inline unsigned char AddUInt64(uint64_t operand1, uint64_t operand2, uint64_t* result) {
  *result = operand1 + operand2;
  return static_cast<unsigned char>(*result < operand1);
}

/// input mod modulus with q_barr = floor(2^64 / modulus); OutputModFactor == 2
/// skips the last conditional subtraction.
template <int OutputModFactor = 1>
uint64_t BarrettReduce64(uint64_t input, uint64_t modulus, uint64_t q_barr) {
  HEXL_CHECK(modulus != 0, "modulus == 0");
  const uint64_t r = input - MultiplyUInt64Hi<64>(input, q_barr) * modulus;
  if (OutputModFactor == 2) return r;
  return r >= modulus ? r - modulus : r;
}

/// x mod modulus for x < InputModFactor * modulus (InputModFactor in {1,2,4,8}).
template <int InputModFactor>
uint64_t ReduceMod(uint64_t x, uint64_t modulus, const uint64_t* twice_modulus = nullptr,
                   const uint64_t* four_times_modulus = nullptr) {
  static_assert(InputModFactor == 1 || InputModFactor == 2 || InputModFactor == 4 ||
                    InputModFactor == 8,
                "InputModFactor should be 1, 2, 4, or 8");
  if (InputModFactor >= 8) {
    HEXL_CHECK(four_times_modulus != nullptr, "four_times_modulus should not be nullptr");
    if (x >= *four_times_modulus) x -= *four_times_modulus;
  }
  if (InputModFactor >= 4) {
    HEXL_CHECK(twice_modulus != nullptr, "twice_modulus should not be nullptr");
    if (x >= *twice_modulus) x -= *twice_modulus;
  }
  if (InputModFactor >= 2) {
    if (x >= modulus) x -= modulus;
  }
  return x;
}

/// Montgomery reduction of T = T_hi * 2^BitShift + T_lo modulo q with R = 2^r.
template <int BitShift>
inline uint64_t MontgomeryReduce(uint64_t T_hi, uint64_t T_lo, uint64_t q, int r,
                                 uint64_t mod_R_msk, uint64_t inv_mod) {
  HEXL_CHECK(BitShift == 64 || BitShift == 52, "Unsupported BitShift " << BitShift);
  const uint64_t mfac = ((T_lo & mod_R_msk) * inv_mod) & mod_R_msk;
  uint64_t mq_hi, mq_lo;
  MultiplyUInt64(mfac, q, &mq_hi, &mq_lo);
  if (BitShift == 52) {
    mq_hi = (mq_hi << 12) | (mq_lo >> 52);
    mq_lo &= (1ULL << 52) - 1;
  }
  uint64_t t_lo = T_lo + mq_lo;
  const uint64_t carry = t_lo < T_lo ? 1 : 0;
  uint64_t t_hi = T_hi + mq_hi + carry;
  t_hi <<= (BitShift - r);
  t_lo >>= r;
  t_lo += t_hi;
  return t_lo >= q ? t_lo - q : t_lo;
}

/// x in [0, 2^r) with q * x == -1 mod 2^r (q odd), by Newton / Hensel lifting.
inline uint64_t HenselLemma2adicRoot(uint32_t r, uint64_t q) {
  uint64_t x = 1;  // q * 1 == -1 mod 2 for odd q
  for (uint32_t bits = 1; bits < r; bits *= 2) x = x * (2 + q * x);  // doubles the precision
  // the iteration above solves q*x == -1: start from x0 with q*x0 == -1 mod 2
  return r >= 64 ? x : (x & ((1ULL << r) - 1));
}

}  // namespace hexl
}  // namespace intel
