// hexl/util/compiler.hpp -- 128-bit arithmetic helpers (GNU / clang __int128),
// the counterparts of hexl/include/hexl/util/gcc.hpp:16-59 in the reference.
#pragma once
#include "hexl/util/check.hpp"
#include "hexl/util/defines.hpp"
#include "hexl/util/types.hpp"

namespace intel {
namespace hexl {

/// x * y as a 128-bit integer.
inline uint128_t MultiplyUInt64(uint64_t x, uint64_t y) {
  return static_cast<uint128_t>(x) * y;
}

/// x * y split into high and low words.
inline void MultiplyUInt64(uint64_t x, uint64_t y, uint64_t* prod_hi, uint64_t* prod_lo) {
  const uint128_t p = MultiplyUInt64(x, y);
  *prod_hi = static_cast<uint64_t>(p >> 64);
  *prod_lo = static_cast<uint64_t>(p);
}

/// (x * y) >> BitShift, low 64 bits.
template <int BitShift>
inline uint64_t MultiplyUInt64Hi(uint64_t x, uint64_t y) {
  return static_cast<uint64_t>(MultiplyUInt64(x, y) >> BitShift);
}

/// (input_hi * 2^64 + input_lo) mod modulus.
inline uint64_t BarrettReduce128(uint64_t input_hi, uint64_t input_lo, uint64_t modulus) {
  HEXL_CHECK(modulus != 0, "modulus == 0");
  const uint128_t v = (static_cast<uint128_t>(input_hi) << 64) | input_lo;
  return static_cast<uint64_t>(v % modulus);
}

/// Low 64 bits of floor((x1 * 2^64 + x0) / y).
inline uint64_t DivideUInt128UInt64Lo(uint64_t x1, uint64_t x0, uint64_t y) {
  const uint128_t v = (static_cast<uint128_t>(x1) << 64) | x0;
  return static_cast<uint64_t>(v / y);
}

/// Index of the most significant set bit, floor(log2(input)); input > 0.
inline uint64_t MSB(uint64_t input) { return 63 - static_cast<uint64_t>(__builtin_clzll(input)); }

#define HEXL_LOOP_UNROLL_4 _Pragma("GCC unroll 4")
#define HEXL_LOOP_UNROLL_8 _Pragma("GCC unroll 8")

}  // namespace hexl
}  // namespace intel
