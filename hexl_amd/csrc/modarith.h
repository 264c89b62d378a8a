// modarith.h -- word-sized modular arithmetic for gfx950 device code.
// Integer VALU only.  Measured issue cost on MI355X (tools/ubench.hip):
// v_mad_u64_u32 / v_mul_lo_u32 / v_mul_hi_u32 and every 64-bit add/compare
// ~4 cycles per wave, 32-bit add/cndmask/v_mad_u32_u24 ~2.
//
// Semantics follow the reference's scalar primitives
// (hexl/include/hexl/number-theory/number-theory.hpp:127-141 MultiplyModLazy,
// :195-205 BarrettReduce64, :214-258 ReduceMod; hexl/ntt/ntt-default.hpp:28-42
// and :112-125 for the two Harvey butterflies).  Two range policies:
//
//   Strict (any q < 2^62): the reference's invariants -- forward values in
//     [0,4q) with one conditional subtraction per butterfly, inverse values in
//     [0,2q).
//   Lazy (q < 2^56): the 64-bit word has >= 8 spare bits, so the forward
//     network never subtracts conditionally (values grow by at most 2q per
//     stage: < (4 + 2 log2 N) q < 2^62) and is reduced once at the end with a
//     single-word Barrett step; the inverse network skips the conditional
//     subtractions inside each register subtree (growth 2x per stage, at most
//     32q) and restores [0,2q) at subtree exit.  Because every multiplicand
//     is below 2^62 the Shoup factor is kept with 63 fractional bits and the
//     product is written as a chain of v_mad_u64_u32 that cannot overflow
//     (mul_lazy63): 10 multiplies and ~9 other VALU per butterfly instead of
//     ~37 instructions.
// Both produce the same canonical outputs; lazy outputs stay inside the
// reference's ranges ([0,4q) forward, [0,2q) inverse).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HX_HD __host__ __device__ __forceinline__
#define HX_D __device__ __forceinline__
#else
#define HX_HD inline
#define HX_D inline
#endif

namespace hexl_amd {

typedef uint64_t u64;
typedef uint32_t u32;

HX_HD u64 mul_hi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

// x - m if x >= m else x   (x < 2m)
HX_HD u64 csub(u64 x, u64 m) { return x >= m ? x - m : x; }

// Shoup / Harvey lazy product: x*W - floor(x*Wp / 2^64)*q  in [0, 2q) for ANY
// 64-bit x, W < q, Wp = floor(W * 2^64 / q).
HX_HD u64 mul_lazy(u64 x, u64 W, u64 Wp, u64 q) {
  u64 Q = mul_hi64(x, Wp);
  return x * W - Q * q;
}

// Per-modulus constants every kernel receives.
struct ModConst {
  u64 q;
  u64 two_q;
  u64 neg_q;    // 2^64 - q
  u64 barrett;  // floor(2^64 / q): single-word Barrett factor (lazy policy only)
};

#if defined(__HIP_DEVICE_COMPILE__)
#define HX_OPAQUE(v) asm("" : "+v"(v))
#else
#define HX_OPAQUE(v) (void)0
#endif

// Lazy-policy product for x < 2^62: W63 = floor(W * 2^63 / q) and
// Q = floor(2x * W63 / 2^64); x*W - Q*q lies in [0, 2q).  With 2x < 2^63 and
// W63 < 2^63 the middle column a1*b0 + a0*b1 + hi(a0*b0) stays below 2^64, so
// it is two chained v_mad_u64_u32 without carry handling.  The low 64 bits of
// x*W + Q*(-q) are two more chained mads for bits 0..63 of the low products and
// a four-mad chain whose low word is the sum of the cross terms.  HX_OPAQUE
// keeps the compiler from narrowing that chain to v_mul_lo_u32 + adds.
HX_HD u64 mul_lazy63(u64 x, u64 W, u64 W63, u64 neg_q) {
  const u64 x2 = x << 1;
  const u32 a0 = (u32)x2, a1 = (u32)(x2 >> 32);
  const u32 b0 = (u32)W63, b1 = (u32)(W63 >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
  const u32 m = __umulhi(a0, b0);
#else
  const u32 m = (u32)(((u64)a0 * b0) >> 32);
#endif
  u64 S = (u64)a1 * b0 + m;
  S = (u64)a0 * b1 + S;
  const u64 Q = (u64)a1 * b1 + (S >> 32);
  const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
  const u32 w0 = (u32)W, w1 = (u32)(W >> 32);
  const u32 q0 = (u32)Q, q1 = (u32)(Q >> 32);
  const u32 n0 = (u32)neg_q, n1 = (u32)(neg_q >> 32);
  u64 lo = (u64)x0 * w0;
  lo = (u64)q0 * n0 + lo;
  u64 c = (u64)x0 * w1;
  HX_OPAQUE(c);
  c = (u64)x1 * w0 + c;
  HX_OPAQUE(c);
  c = (u64)q0 * n1 + c;
  HX_OPAQUE(c);
  c = (u64)q1 * n0 + c;
  HX_OPAQUE(c);
  return lo + ((u64)(u32)c << 32);
}

struct Strict {  // any q < 2^62; tables hold floor(W * 2^64 / q)
  static constexpr bool kLazy = false;
  static HX_HD u64 mul(u64 x, u64 W, u64 Wp, const ModConst& m) {
    return mul_lazy(x, W, Wp, m.q);
  }
};
struct Lazy {  // q < 2^56; tables hold floor(W * 2^63 / q); multiplicands < 2^62
  static constexpr bool kLazy = true;
  static HX_HD u64 mul(u64 x, u64 W, u64 Wp, const ModConst& m) {
    return mul_lazy63(x, W, Wp, m.neg_q);
  }
};

// Forward (Cooley-Tukey) Harvey butterfly.
// Strict: x,y in [0,4q) -> [0,4q).  Lazy: x < B*q -> x', y' < (B+2)*q.
template <class A>
HX_HD void fwd_butterfly(u64& x, u64& y, u64 W, u64 Wp, const ModConst& m) {
  const u64 tx = A::kLazy ? x : csub(x, m.two_q);
  const u64 T = A::mul(y, W, Wp, m);
  x = tx + T;
  y = tx + m.two_q - T;
}

// End of the forward network: bring a value to [0,q) (canonical) or leave it in
// the reference's lazy range [0,4q).
template <class A>
HX_HD u64 fwd_finish(u64 x, const ModConst& m, bool canonical) {
  if (A::kLazy) {
    x = x - mul_hi64(x, m.barrett) * m.q;  // [0, 2q), BarrettReduce64<2>
    return canonical ? csub(x, m.q) : x;
  }
  return canonical ? csub(csub(x, m.two_q), m.q) : x;
}

// Inverse (Gentleman-Sande) Harvey butterfly at depth `k` of a register
// subtree (k = 0 for the first stage the subtree runs).
// Strict: x,y in [0,2q) -> [0,2q).  Lazy: x,y < 2q*2^k -> x' < 2q*2^(k+1),
// y' in [0,2q); the offset added before subtracting y is 2q*2^k.
template <class A>
HX_HD void inv_butterfly(u64& x, u64& y, u64 W, u64 Wp, const ModConst& m, int k) {
  const u64 s = x + y;
  const u64 d = x + (A::kLazy ? (m.two_q << k) : m.two_q) - y;
  x = A::kLazy ? s : csub(s, m.two_q);
  y = A::mul(d, W, Wp, m);
}

// Last inverse stage with N^{-1} folded in (ntt-radix-2.cpp:490-509):
// x' = (x+y) * n1, y' = (x-y) * n1W, both lazy in [0,2q).  The sum needs no
// conditional subtraction first because mul_lazy accepts any 64-bit input.
template <class A>
HX_HD void inv_butterfly_last(u64& x, u64& y, u64 n1, u64 n1p, u64 n1w, u64 n1wp,
                              const ModConst& m, int k) {
  const u64 s = x + y;
  const u64 d = x + (A::kLazy ? (m.two_q << k) : m.two_q) - y;
  x = A::mul(s, n1, n1p, m);
  y = A::mul(d, n1w, n1wp, m);
}

// Lazy policy, subtree exit: element with `lz` leading X-steps is < 2q * 2^lz.
template <int LZ>
HX_HD u64 inv_ladder(u64 x, const ModConst& m) {
#pragma unroll
  for (int t = LZ - 1; t >= 0; --t) x = csub(x, m.two_q << t);
  return x;
}

}  // namespace hexl_amd
