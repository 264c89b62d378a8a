// modarith.h -- word-sized modular arithmetic for gfx950 device code (and the
// host table builder).  Integer VALU only: a 64x64 high product is
// 1 v_mul_hi_u32 + 3 v_mad_u64_u32, a low product 2 v_mul_lo_u32 +
// 1 v_mad_u64_u32 (measured issue cost on MI355X: ~4 cycles per wave for each
// of those, 2 for adds -- tools/ubench.hip).
//
// Semantics follow the reference's scalar primitives
// (hexl/include/hexl/number-theory/number-theory.hpp:127-141 MultiplyModLazy,
// :195-205 BarrettReduce64, :214-258 ReduceMod; hexl/ntt/ntt-default.hpp:28-42
// and :112-125 for the two Harvey butterflies).
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

#if defined(__HIPCC__)
// Same result for q <= 2^55: the true value lies in [0, 2q) which is below
// 2^56, so only bits 0..55 of (x*W - Q*q) are needed.  Bits 32..55 of the two
// low products come from four 24x24-bit multiplies (v_mad_u32_u24, full rate)
// instead of four v_mul_lo_u32.
HX_D u64 mul_lazy_q55(u64 x, u64 W, u64 Wp, u64 q) {
  u64 Q = __umul64hi(x, Wp);
  u32 xl = (u32)x, xh = (u32)(x >> 32);
  u32 Wl = (u32)W, Wh = (u32)(W >> 32);
  u32 Ql = (u32)Q, Qh = (u32)(Q >> 32);
  u32 ql = (u32)q, qh = (u32)(q >> 32);
  u64 lo = (u64)xl * Wl - (u64)Ql * ql;  // wraps mod 2^64
  u32 cross = __umul24(xh, Wl) + __umul24(xl, Wh) - __umul24(Qh, ql) -
              __umul24(Ql, qh);  // exact mod 2^24
  u64 r = lo + ((u64)cross << 32);
  return r & ((1ULL << 56) - 1);
}
#endif

struct Q64 {  // generic modulus q < 2^62
  static HX_HD u64 mul(u64 x, u64 W, u64 Wp, u64 q) { return mul_lazy(x, W, Wp, q); }
};
#if defined(__HIPCC__)
struct Q55 {  // q <= 2^55 fast path
  static HX_D u64 mul(u64 x, u64 W, u64 Wp, u64 q) { return mul_lazy_q55(x, W, Wp, q); }
};
#endif

// Forward (Cooley-Tukey) Harvey butterfly, x,y in [0,4q) -> [0,4q)
template <class A>
HX_HD void fwd_butterfly(u64& x, u64& y, u64 W, u64 Wp, u64 q, u64 two_q) {
  u64 tx = csub(x, two_q);
  u64 T = A::mul(y, W, Wp, q);
  x = tx + T;
  y = tx + two_q - T;
}

// Inverse (Gentleman-Sande) Harvey butterfly, x,y in [0,2q) -> [0,2q)
template <class A>
HX_HD void inv_butterfly(u64& x, u64& y, u64 W, u64 Wp, u64 q, u64 two_q) {
  u64 s = x + y;
  u64 d = x + two_q - y;
  x = csub(s, two_q);
  y = A::mul(d, W, Wp, q);
}

// Last inverse stage with N^{-1} folded in (ntt-radix-2.cpp:490-509):
// x' = (x+y) * n1, y' = (x-y) * n1W, both lazy in [0,2q).  The sum needs no
// conditional subtraction first because mul_lazy accepts any 64-bit input.
template <class A>
HX_HD void inv_butterfly_last(u64& x, u64& y, u64 n1, u64 n1p, u64 n1w, u64 n1wp,
                              u64 q, u64 two_q) {
  u64 s = x + y;
  u64 d = x + two_q - y;
  x = A::mul(s, n1, n1p, q);
  y = A::mul(d, n1w, n1wp, q);
}

HX_HD u64 reduce_4q_to_q(u64 x, u64 q, u64 two_q) { return csub(csub(x, two_q), q); }

}  // namespace hexl_amd
