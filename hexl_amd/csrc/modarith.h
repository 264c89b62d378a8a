// modarith.h -- word-sized modular arithmetic for gfx950 device code.
// Integer VALU only.  Measured issue rate on MI355X (tools/ubench.hip, wall
// clock): a SIMD retires one simple wave64 VALU instruction (32-bit add, shift,
// cndmask, v_lshl_add_u64) per ~3 cycles and one multiply-class instruction
// (v_mad_u64_u32, v_mul_lo/hi_u32, v_mad_u32_u24) per ~5 cycles, and two waves
// per SIMD already saturate it.  The NTT kernels are bound by this rate, so the
// butterflies below are written for minimum instruction count.
//
// Semantics follow the reference's scalar primitives
// (hexl/include/hexl/number-theory/number-theory.hpp:127-141 MultiplyModLazy,
// :195-205 BarrettReduce64, :214-258 ReduceMod; hexl/ntt/ntt-default.hpp:28-42
// and :112-125 for the two Harvey butterflies).  Range policies of the 64-bit integer
// arithmetic (Small, Harvey60, Fp64 and the bounded members of the Lazy family -- Lazy32 and
// Lazy16, 2^56 <= q < 2^59 -- are described at their structs below):
//
//   Strict (any q < 2^62): the reference's invariants -- forward values in
//     [0,4q) with one conditional subtraction per butterfly, inverse values in
//     [0,2q); 64-bit Shoup factors; 23 instructions per butterfly (10 multiplies).
//   Lazy (q < 2^56): the 64-bit word has >= 8 spare bits.  Values are kept
//     DOUBLED (D = 2x) between the first load and the last store of a
//     transform, Shoup factors carry 63 fractional bits, and the quotient
//     estimate drops the lowest partial product, so a lazy product is
//     T2 = D*W - Q*2q in [0,6q) (= 2 * (x*W mod q + {0,1,2}q)) from 9 chained
//     v_mad_u64_u32 and 2 moves.  The forward network never subtracts
//     conditionally (doubled values grow by at most 6q per stage: at most
//     (8 + 6*17) q < 2^63) and is reduced once at the end; the inverse network
//     tracks the range of every element at compile time and reduces only where
//     the next use of a value would pass 2^63 (lazy_inverse.h).  14 (forward) /
//     15 (inverse) instructions per butterfly.
// All policies produce the same canonical outputs; lazy outputs stay inside the
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

// HX_OPAQUE(v): the value of v passes through an empty asm statement, so the compiler
// cannot reassociate or narrow the expression that produced it.
#if defined(__HIP_DEVICE_COMPILE__)
#define HX_OPAQUE(v) asm("" : "+v"(v))
#else
#define HX_OPAQUE(v) (void)0
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

// Same for x, m < 2^63 (every lazy-policy value): x + (-m) is one v_lshl_add_u64
// and the selection tests the sign of its high word -- no 64-bit compare, no
// carry chain.  neg_m = 2^64 - m.
HX_HD u64 csub_neg(u64 x, u64 neg_m) {
  const u64 t = x + neg_m;
  return (int32_t)(u32)(t >> 32) < 0 ? x : t;
}

// The same for any 64-bit x, m > 0: t = x + (-m) wraps above x exactly when x < m;
// one v_lshl_add_u64, one 64-bit compare, two v_cndmask.  neg_m = 2^64 - m.
HX_HD u64 csub_wrap(u64 x, u64 neg_m) {
  const u64 t = x + neg_m;
  return t <= x ? t : x;
}

// Shoup / Harvey lazy product: x*W - floor(x*Wp / 2^64)*q  in [0, 2q) for ANY
// 64-bit x, W < q, Wp = floor(W * 2^64 / q).
HX_HD u64 mul_lazy(u64 x, u64 W, u64 Wp, u64 q) {
  u64 Q = mul_hi64(x, Wp);
  return x * W - Q * q;
}

// Per-modulus constants every kernel receives (host: make_mod_const).
struct ModConst {
  u64 q;
  u64 two_q;
  u64 neg_q;  // 2^64 - q (Strict policy)
  u64 neg_two_q;  // 2^64 - 2q
  u64 barrett;    // floor(2^64 / q): single-word Barrett factor (reduce_any)
  // Harvey60 policy only:
  u64 four_q;
  u64 neg_four_q;  // 2^64 - 4q
  // Lazy policy only:
  u64 six_q;
  u32 fin_mul;    // floor(2^(31 + fin_shift) / q)
  u32 fin_shift;  // floor(log2 q)
  // Fp64 policy only:
  double qd;    // q
  double qinv;  // 1 / q, rounded to nearest
  // Rounding on load (KeySwitch: a forward transform whose input is the last RNS component,
  // rounded to the special modulus and brought to this one -- key-switch-internal.cpp:146-175;
  // set per launch by the multi-plan kernels, zero otherwise): the special modulus q_k, its
  // single-word Barrett factor and q_k / 2.
  u64 rnd_qk, rnd_barrett, rnd_half;
};

inline ModConst make_mod_const(u64 q) {  // host only
  ModConst m;
  m.q = q;
  m.two_q = q << 1;
  m.neg_q = 0 - q;
  m.neg_two_q = 0 - (q << 1);
  m.barrett = (u64)((((unsigned __int128)1) << 64) / q);
  m.four_q = q << 2;  // only meaningful (and only read) for q < 2^60 + 2^28 (Harvey60, Lazy16)
  m.neg_four_q = 0 - (q << 2);
  m.six_q = 6 * q;  // only meaningful (and only read) for q < 2^58 (Lazy, Lazy32: 6q < 2^61)
  u32 b = 0;
  while (b < 63 && (q >> (b + 1)) != 0) ++b;
  m.fin_shift = b;
  m.fin_mul = (u32)((((unsigned __int128)1) << (31 + b)) / q);
  m.qd = (double)q;  // exact for q < 2^53; only read for q < 2^50 (the Fp64 family)
  m.qinv = 1.0 / m.qd;
  m.rnd_qk = m.rnd_barrett = m.rnd_half = 0;
  return m;
}

// Scalars of the last inverse stage: n1 = N^-1 mod q, n1w = N^-1 * R[1]^-1, each with its
// Shoup companion (hexl/ntt/ntt-radix-2.cpp:490-497).
// c2, log_n, mont_mask: the multiply-free N^-1 scaling of the sum branch (scale_by_inverse_degree).
struct InvLast {
  u64 n1, n1p, n1w, n1wp;
  u64 c2;         // (q - 1) / N
  u32 log_n;      // log2 N
  u32 mont_mask;  // 2N - 1 where values are held doubled (Lazy, Harvey60), N - 1 otherwise
};

// s * N^-1 modulo q WITHOUT a modular product: a negacyclic modulus is q = 1 + N c2
// (q == 1 mod 2N, hexl/ntt/ntt-internal.cpp:171-186), so q == 1 (mod N) and one Montgomery
// step with radix N needs no multiplication by -q^-1:
//     m = -s mod N,   t = (s + m q) / N = (s + m) / N + m c2,   t N == s (mod q).
// Values held doubled (even numbers standing for their halves) take m = -s mod 2N: then 2N
// divides s + m (q == 1 mod 2N) and t is even again.  t < s / N + q (m / N): below 2q (4q
// doubled) whenever s < 2 N q -- every policy's s from N = 64 on.  A shift, a mask and ONE
// 32 x 32-bit multiply-add (+ a 32-bit multiply for the high word of c2) instead of the nine
// of a Shoup product: the sum branch of the last inverse stage (ntt-radix-2.cpp:490-509,
// X' = (X + Y) N^-1), 16 of the 96 products of a 5-stage strided subtree.
HX_HD u64 scale_by_inverse_degree(u64 s, const InvLast& il) {
  const u32 mq = (0u - (u32)s) & il.mont_mask;
  return ((s + mq) >> il.log_n) + (u64)mq * il.c2;
}

// x mod q for ANY 64-bit x: single-word Barrett with floor(2^64 / q)
// (hexl/eltwise/eltwise-reduce-mod.cpp:32-55 with input_mod_factor == modulus)
HX_HD u64 reduce_any(u64 x, u64 q, u64 barrett) {
  if (x < q) return x;
  return csub(x - mul_hi64(x, barrett) * q, q);
}
// The same without the early exit (for x < q the quotient estimate is 0 anyway): straight-line
// code for the transforms' load paths, where a divergent branch per element would cut the
// kernel into basic blocks.
HX_HD u64 reduce_any_straight(u64 x, u64 q, u64 barrett) {
  return csub(x - mul_hi64(x, barrett) * q, q);
}

#if defined(__HIP_DEVICE_COMPILE__)
HX_HD u32 mul_hi32(u32 a, u32 b) { return __umulhi(a, b); }
#else
HX_HD u32 mul_hi32(u32 a, u32 b) { return (u32)(((u64)a * b) >> 32); }
#endif

// S >> 32 as a 64-bit value (the addend of the quotient's last mad).  The mad that produced S
// left its high word in the ODD register of an aligned pair and the addend wants it in the
// EVEN one with a zero above: left to the compiler that is one v_mov into a pair whose upper
// half is a shared zero register, or -- under register pressure, 36 of the 44 butterflies of
// the 11-stage forward tile pass -- two (move down, clear up).  v_pk_mov_b32 builds the pair
// in ONE instruction with the zero as an inline constant: D.lo = S0[op_sel[0]] = S.hi,
// D.hi = S1[op_sel[1]] = 0.
HX_HD u64 high_word(u64 S) {
#if defined(__HIP_DEVICE_COMPILE__)
  u64 r;
  asm("v_pk_mov_b32 %0, %1, 0 op_sel:[1,0]" : "=v"(r) : "v"(S));
  return r;
#else
  return S >> 32;
#endif
}

// Lazy-policy product on doubled values.  D = 2x < 2^63, W < q < 2^56,
// W63 = floor(W * 2^63 / q).  Returns  acc + T2  (mod 2^64) where
//   T2 = D*W - Q*2q,   Q ~ floor(D * W63 / 2^64).
// With the exact quotient T2 = 2*(x*W - floor(x*W63/2^63)*q) lies in [0,4q)
// (x*W63/2^63 underestimates x*W/q by less than x/2^63 < 1/2).  EXACT == false
// leaves the partial product lo(D)*lo(W63) out of the quotient, which lowers Q by
// at most one more: T2 in [0,6q).  The middle column a1*b0 + a0*b1 (+ hi(a0*b0))
// stays below 2^64 because both factors are below 2^63, so the quotient is three
// chained v_mad_u64_u32; the low 64 bits of D*W + Q*(-2q) are two chained mads
// for the low x low products (seeded with `acc`, which makes the butterfly's
// addition free) and a four-mad chain whose low word is the sum of the cross
// terms.  HX_OPAQUE keeps the compiler from narrowing that chain to v_mul_lo_u32
// + adds.
template <bool EXACT>
HX_HD u64 mul_add_lazy2(u64 acc, u64 D, u64 W, u64 W63, u64 neg_two_q) {
  const u32 a0 = (u32)D, a1 = (u32)(D >> 32);
  const u32 b0 = (u32)W63, b1 = (u32)(W63 >> 32);
  u64 S = (u64)a1 * b0 + (EXACT ? mul_hi32(a0, b0) : 0u);
  if (EXACT) HX_OPAQUE(S);  // keeps hi(a0*b0) the first mad's addend (else: a separate 64-bit add)
  S = (u64)a0 * b1 + S;
  const u64 Q = (u64)a1 * b1 + high_word(S);
  const u32 w0 = (u32)W, w1 = (u32)(W >> 32);
  const u32 q0 = (u32)Q, q1 = (u32)(Q >> 32);
  const u32 n0 = (u32)neg_two_q, n1 = (u32)(neg_two_q >> 32);
  u64 lo = (u64)a0 * w0 + acc;
  lo = (u64)q0 * n0 + lo;
  u64 c = (u64)a0 * w1;
  HX_OPAQUE(c);
  c = (u64)a1 * w0 + c;
  HX_OPAQUE(c);
  c = (u64)q0 * n1 + c;
  HX_OPAQUE(c);
  c = (u64)q1 * n0 + c;
  HX_OPAQUE(c);
  u32 hi = (u32)(lo >> 32) + (u32)c;
  HX_OPAQUE(hi);  // one v_add_u32, not a 64-bit add of a shifted pair
  return ((u64)hi << 32) | (u32)lo;
}

// Strict-policy product for ANY 64-bit y: acc + (y*W - floor(y*Wp / 2^64)*q) mod 2^64,
// the exact Harvey lazy product in [0, 2q) (number-theory.hpp:127-141) written as
// a v_mad_u64_u32 chain.  The middle column a1*b0 + a0*b1 + hi(a0*b0) can exceed
// 64 bits here, so it is summed in two steps whose carries are explicit:
//   S1 = a1*b0 + hi(a0*b0)  (< 2^64),  (carry, T1) = a0*b1 + S1  (65 bits),
//   Q  = a1*b1 + (carry * 2^32 + hi32(T1)).
// 10 multiplies and ~8 other instructions instead of the ~30 of __umul64hi plus
// two 64-bit low products.
HX_HD u64 mul_add_strict(u64 acc, u64 y, u64 W, u64 Wp, u64 neg_q) {
  const u32 a0 = (u32)y, a1 = (u32)(y >> 32);
  const u32 b0 = (u32)Wp, b1 = (u32)(Wp >> 32);
  u64 S1 = (u64)a1 * b0 + mul_hi32(a0, b0);
  HX_OPAQUE(S1);
  u64 T1;
  const bool carry = __builtin_add_overflow((u64)a0 * b1, S1, &T1);  // mad with carry-out
  const u64 Q = (u64)a1 * b1 + ((T1 >> 32) | ((u64)carry << 32));
  const u32 w0 = (u32)W, w1 = (u32)(W >> 32);
  const u32 q0 = (u32)Q, q1 = (u32)(Q >> 32);
  const u32 n0 = (u32)neg_q, n1 = (u32)(neg_q >> 32);
  u64 lo = (u64)a0 * w0 + acc;
  lo = (u64)q0 * n0 + lo;
  u64 c = (u64)a0 * w1;
  HX_OPAQUE(c);
  c = (u64)a1 * w0 + c;
  HX_OPAQUE(c);
  c = (u64)q0 * n1 + c;
  HX_OPAQUE(c);
  c = (u64)q1 * n0 + c;
  HX_OPAQUE(c);
  u32 hi = (u32)(lo >> 32) + (u32)c;
  HX_OPAQUE(hi);
  return ((u64)hi << 32) | (u32)lo;
}

struct Strict {  // any q < 2^62; tables hold floor(W * 2^64 / q); plain values
  static constexpr bool kLazy = false;
  static constexpr bool kSmall = false;
  static constexpr bool kFp = false;
  static constexpr bool kH60 = false;
  static constexpr int kLimit = 0;
  static constexpr bool kExact = false;
  static constexpr int kFwdRun = 0, kInvRun = 0;
};
// The Lazy family: doubled values, 63-bit Shoup factors, no conditional subtraction per
// butterfly.  kLimit = what floor(2^63 / q) is at least for the moduli the member serves: every
// (doubled) value stays below kLimit * q.  kExact: the forward product keeps the lowest partial
// product (T2 < 4q instead of < 6q: one instruction more, a third less growth).
//   Lazy   (kLimit 128, q < 2^56)  forward: never reduced before the finish
//   Lazy32 (kLimit 32,  2^56 <= q < 2^58)  forward: the x operands of a stage lose 16q by one
//          sign-tested subtraction whenever the next growth would pass 32q (every other stage)
//   Lazy16 (kLimit 16,  2^58 <= q < 2^59)  the same with 8q and the exact product (alternate
//          stages): 17 instructions per forward butterfly where Harvey60 has 19
// The inverse network of every member is lazy_inverse.h's, scheduled for its limit.
template <int LIMIT, bool EXACT>
struct LazyT {
  static constexpr bool kLazy = true;
  static constexpr bool kSmall = false;
  static constexpr bool kFp = false;
  static constexpr bool kH60 = false;
  static constexpr int kLimit = LIMIT;
  static constexpr bool kExact = EXACT;
  static constexpr int kFwdRun = 0, kInvRun = 0;
};
typedef LazyT<128, false> Lazy;
typedef LazyT<32, false> Lazy32;
typedef LazyT<16, true> Lazy16;
// 2^56 <= q < 2^60 + 2^28 (where SEAL's and OpenFHE's 60-bit primes live, and the smallest
// primes above 2^60 that the reference's tests and benchmarks generate): the reference's
// Harvey invariants -- forward values in [0,4q), inverse values in [0,2q), one conditional
// subtraction per butterfly -- held on DOUBLED values (D = 2x < 8q) like the Lazy policy's,
// so that the product is its carry-free v_mad_u64_u32 chain with a 63-bit Shoup factor
// (exact quotient: T2 < 3q + q D / 2^63 < 4q) and the conditional subtraction a sign test
// (4q < 2^63): 19 (forward) / 20 (inverse) instructions per butterfly against Strict's 23.
// The chain's middle column a1*b0 + a0*b1 + hi(a0*b0) stays below 2^64 as long as
// a1 = D >> 32 <= 2^31 (b1 < 2^31): 8q <= 2^63 + 2^31, hence the bound.
struct Harvey60 {
  static constexpr bool kLazy = false;
  static constexpr bool kSmall = false;
  static constexpr bool kFp = false;
  static constexpr bool kH60 = true;
  static constexpr int kLimit = 0;
  static constexpr bool kExact = false;
  static constexpr int kFwdRun = 0, kInvRun = 0;
};
// q < 2^30 (the reference's 32-bit path, hexl/ntt/ntt-internal.cpp:218-226,
// :279-287): every value of the Strict invariants is below 4q < 2^32, so the
// arithmetic runs on the low words only -- Shoup factor floor(W * 2^32 / q), three
// 32-bit multiplies per butterfly instead of nine, conditional subtraction as one
// v_sub + v_min_u32.  Storage stays 64-bit (the API's), high words zero.
struct Small {
  static constexpr bool kLazy = false;
  static constexpr bool kSmall = true;
  static constexpr bool kFp = false;
  static constexpr bool kH60 = false;
  static constexpr int kLimit = 0;
  static constexpr bool kExact = false;
  static constexpr int kFwdRun = 0, kInvRun = 0;
};

// 2^30 <= q < 2^50 (the moduli the reference sends to its IFMA-52 / FP64-assisted
// kernels, hexl/ntt/ntt-internal.cpp:202-213, hexl/eltwise/eltwise-mult-mod.cpp:38-52;
// SURVEY 8f row 3): values are exact integers held in DOUBLES, residues are balanced
// (signed), and a twiddle is ONE word (W in (-q/2, q/2] as a double; no Shoup
// companion).  `v_fma_f64` issues at the rate of `v_mad_u64_u32`, and a product is
//     h = y*W (rounded),  l = fma(y, W, -h)  (exact: h + l = y*W),
//     k = rint(h * (1/q)),  T = fma(-k, q, h) + l  (= y*W - k*q exactly)
// six instructions against eleven: 8 per butterfly instead of 14 / 15
// (tools/ubench3: 17.4 ns against 25.8 ns per wave-butterfly).
//
// Exactness and range (every step is exact integer arithmetic as long as all values
// stay below 2^53 in magnitude; q < 2^50, |W| <= q/2, |y| = B q):
//   |y W / q - k| <= 1/2 (rint) + |y| 2^-53 (1/q and the product each rounded once)
//                    + |y| 2^-54 (l)      =>   |T| <= (0.5 + 0.1875 B) q,
//   h - k q is an integer below 2^53, so the fma is exact, and so is the sum with l.
// Forward (x' = x + T, y' = x - T): B' = 1.1875 B + 0.5 -- from 0.5 the bound after
// 1..7 stages is 1.09, 1.80, 2.64, 3.63, 4.81, 6.21, 7.88 (< 8 <= 2^53 / q): at most
// kFpFwdRun = 7 stages between two full reductions.  Inverse (x' = x + y,
// y' = (x - y) W): sums double, so at most kFpInvRun = 3 stages (4 q) between two.
// A full reduction v - rint(v / q) q leaves |v| <= (0.5 + 2^-48) q in three
// instructions.  tests/cpp/host_arith_check.cpp replays whole networks through these
// functions on the CPU and checks every intermediate bound.
// kFwdRun / kInvRun: the longest run of stages between two full reductions.  For q < 2^47 the
// same analysis has 2^53 / q >= 64 to work with and a per-unit error of 1.5 q 2^-53 < 0.0235:
// a forward pass of up to 14 stages grows a fully reduced value to < 9q (no reduction between
// the first load and the last store of a pass), an inverse run of 6 stages doubles it to
// 32q (differences of two such values stay below 2^53): Fp64L.  SEAL's default moduli up to
// N = 8192 (36- to 44-bit primes) are of this kind.
template <int FWD_RUN, int INV_RUN>
struct Fp64T {
  static constexpr bool kLazy = false;
  static constexpr bool kSmall = false;
  static constexpr bool kFp = true;
  static constexpr bool kH60 = false;
  static constexpr int kLimit = 0;
  static constexpr bool kExact = false;
  static constexpr int kFwdRun = FWD_RUN;
  static constexpr int kInvRun = INV_RUN;
};
typedef Fp64T<7, 3> Fp64;    // 2^47 <= q < 2^50
typedef Fp64T<24, 6> Fp64L;  // 2^30 <= q < 2^47
constexpr int kFpFwdRun = 7;
constexpr int kFpInvRun = 3;

HX_HD double fp_bits_to_double(u64 b) { return __builtin_bit_cast(double, b); }
HX_HD u64 fp_double_to_bits(double d) { return __builtin_bit_cast(u64, d); }

// integer x < 2^52 -> the double x: (2^52 + x) - 2^52, the integer dropped into the
// mantissa of 2^52
HX_HD double fp_from_u64(u64 x) { return fp_bits_to_double(x | 0x4330000000000000ULL) - 4503599627370496.0; }
// the double r, an integer in [0, 2^52) -> integer
HX_HD u64 fp_to_u64(double r) { return fp_double_to_bits(r + 4503599627370496.0) & 0x000FFFFFFFFFFFFFULL; }

// v - rint(v / q) q: |result| <= (0.5 + |v / q| 2^-51) q
HX_HD double fp_reduce(double v, const ModConst& m) {
  return __builtin_fma(-__builtin_rint(v * m.qinv), m.qd, v);
}
// y * W - k q, see above
HX_HD double fp_mul(double y, double W, const ModConst& m) {
  const double h = y * W;
  const double l = __builtin_fma(y, W, -h);
  const double k = __builtin_rint(h * m.qinv);
  return __builtin_fma(-k, m.qd, h) + l;
}
// any internal value -> canonical residue in [0, q) as an integer
HX_HD u64 fp_canonical(double v, const ModConst& m) {
  double r = fp_reduce(v, m);
  r = r < 0.0 ? r + m.qd : r;
  return fp_to_u64(r);
}

// end of a forward pass: fully reduced internal value, or (canonical) the residue in
// [0, q) as an integer -- one reduction serves both
HX_HD u64 fp_pass_end(double v, const ModConst& m, bool canonical) {
  const double r = fp_reduce(v, m);
  const double c = r < 0.0 ? r + m.qd : r;
  return canonical ? fp_to_u64(c) : fp_double_to_bits(r);
}

HX_HD void fwd_butterfly_fp(u64& x, u64& y, double W, const ModConst& m) {
  const double xd = fp_bits_to_double(x);
  const double t = fp_mul(fp_bits_to_double(y), W, m);
  x = fp_double_to_bits(xd + t);
  y = fp_double_to_bits(xd - t);
}
HX_HD void inv_butterfly_fp(u64& x, u64& y, double W, const ModConst& m) {
  const double xd = fp_bits_to_double(x), yd = fp_bits_to_double(y);
  x = fp_double_to_bits(xd + yd);
  y = fp_double_to_bits(fp_mul(xd - yd, W, m));
}
// last inverse stage: x' = (x + y) n1, y' = (x - y) n1w  (n1 = N^-1, n1w = N^-1 W, balanced)
HX_HD void inv_butterfly_last_fp(u64& x, u64& y, double n1, double n1w, const ModConst& m) {
  const double xd = fp_bits_to_double(x), yd = fp_bits_to_double(y);
  x = fp_double_to_bits(fp_mul(xd + yd, n1, m));
  y = fp_double_to_bits(fp_mul(xd - yd, n1w, m));
}

// x*W - floor(x*Wp / 2^32)*q in [0, 2q) for any 32-bit x; W < q < 2^30, Wp = floor(W 2^32 / q)
HX_HD u32 mul_small(u32 x, u32 W, u32 Wp, u32 q) { return x * W - mul_hi32(x, Wp) * q; }
// x - m if x >= m else x, as min(x, x - m) on unsigned words
HX_HD u32 csub32(u32 x, u32 m) {
  const u32 t = x - m;
  return t < x ? t : x;
}

// Value as held inside a transform <-> value in the caller's buffer.
template <class A>
HX_HD u64 to_internal(u64 x, const ModConst& m) {
  if (A::kFp) return fp_double_to_bits(fp_reduce(fp_from_u64(x), m));  // x < 4q < 2^52
  return (A::kLazy || A::kH60) ? x << 1 : x;
}
// Fp64: full reduction of an internal value (a no-op for the integer policies)
template <class A>
HX_HD u64 fp_bound(u64 v, const ModConst& m) {
  if (A::kFp) return fp_double_to_bits(fp_reduce(fp_bits_to_double(v), m));
  return v;
}

// Forward (Cooley-Tukey) Harvey butterfly.
// Strict: x,y in [0,4q) -> [0,4q).
// Lazy (doubled): x, y < B*q -> x', y' < (B+6)*q;  y' = 2x + 6q - x'.
template <class A>
HX_HD void fwd_butterfly(u64& x, u64& y, u64 W, u64 Wp, const ModConst& m) {
  if (A::kLazy) {
    if (A::kExact) {  // x < B q -> x', y' < (B + 4) q;  y' = 2x + 4q - x'
      const u64 xs = mul_add_lazy2<true>(x, y, W, Wp, m.neg_two_q);
      y = (x << 1) + m.four_q - xs;
      x = xs;
    } else {
      const u64 xs = mul_add_lazy2<false>(x, y, W, Wp, m.neg_two_q);
      y = (x << 1) + m.six_q - xs;
      x = xs;
    }
  } else if (A::kH60) {  // doubled: x, y < 8q -> < 8q
    const u64 tx = csub_neg(x, m.neg_four_q);                       // < 4q
    const u64 xs = mul_add_lazy2<true>(tx, y, W, Wp, m.neg_two_q);  // tx + T2, T2 < 4q
    y = (tx << 1) + m.four_q - xs;                                  // tx + 4q - T2
    x = xs;
  } else if (A::kSmall) {
    const u32 tx = csub32((u32)x, (u32)m.two_q);
    const u32 T = mul_small((u32)y, (u32)W, (u32)Wp, (u32)m.q);
    x = tx + T;
    y = tx + (u32)m.two_q - T;
  } else {
    const u64 tx = csub_wrap(x, m.neg_two_q);
    const u64 xs = mul_add_strict(tx, y, W, Wp, m.neg_q);
    y = (tx << 1) + m.two_q - xs;  // = tx + 2q - T (mod 2^64; the true value is < 4q)
    x = xs;
  }
}

// Lazy policy: a doubled value D < 2^(b+8), b = floor(log2 q) >= 32 (the Lazy policy serves
// 2^32 <= q < 2^56, choose_policy), brought to [0,4q).  The quotient D / 2q fits in 7 bits and is
// estimated from the top 8 bits of D with one 32-bit multiply: Qe = hi32((D >> b) *
// floor(2^(31+b) / q)) is floor(D / 2q) or one less.  D >> b is a 32-bit shift of the high
// word; D - Qe*2q is one mad for the low product and the carry into the high word plus a
// 32-bit multiply-add for Qe * hi32(-2q): 5 instructions.
HX_HD u64 lazy_estimate_reduce(u64 D, const ModConst& m) {
  const u32 s = (u32)(D >> 32) >> (m.fin_shift - 32);
  const u32 qe = mul_hi32(s, m.fin_mul);
  const u32 n0 = (u32)m.neg_two_q, n1 = (u32)(m.neg_two_q >> 32);
  const u64 lo = (u64)qe * n0 + D;
  u32 t = qe * n1;
  HX_OPAQUE(t);  // v_mul_lo_u32 + v_add_u32 (left alone: a second 64-bit mad between two moves)
  const u32 hi = (u32)(lo >> 32) + t;
  return ((u64)hi << 32) | (u32)lo;
}

// End of the forward network: bring a value to [0,q) (canonical) or leave it in
// the reference's lazy range [0,4q).
// Lazy: D = 2x < 2^(b+8) with b = floor(log2 q) (x < 55q); the quotient x/q fits
// in 7 bits, so it is estimated from the top 8 bits of x with one 32-bit
// multiply: Qe = hi32((D >> b) * floor(2^(31+b)/q)) is floor(x/q) or one less,
// r2 = D - Qe*2q in [0,4q).
template <class A>
HX_HD u64 fwd_finish(u64 x, const ModConst& m, bool canonical) {
  if (A::kFp) return fp_canonical(fp_bits_to_double(x), m);  // also a legal lazy output
  if (A::kLazy) {
    u64 r2 = lazy_estimate_reduce(x, m);
    if (canonical) r2 = csub_neg(r2, m.neg_two_q);
    return r2 >> 1;
  }
  if (A::kH60) {  // doubled value < 8q
    if (canonical) x = csub_neg(csub_neg(x, m.neg_four_q), m.neg_two_q);
    return x >> 1;
  }
  if (A::kSmall) return canonical ? csub32(csub32((u32)x, (u32)m.two_q), (u32)m.q) : x;
  return canonical ? csub(csub(x, m.two_q), m.q) : x;
}

// Inverse (Gentleman-Sande) Harvey butterfly (hexl/ntt/ntt-default.hpp:112-125) of the policies
// that keep the reference's invariant -- every value in [0,2q) (Harvey60: doubled, [0,4q)) with
// one conditional subtraction per butterfly.  The Lazy policy's inverse network, which subtracts
// only where its compile-time range bookkeeping says it must, is lazy_inverse.h.
template <class A>
HX_HD void inv_butterfly(u64& x, u64& y, u64 W, u64 Wp, const ModConst& m) {
  static_assert(!A::kLazy && !A::kFp, "Lazy: lazy_inverse.h; Fp64: inv_butterfly_fp");
  const u64 s = x + y;
  if (A::kH60) {  // doubled: x, y < 4q -> < 4q
    const u64 d = x + m.four_q - y;  // < 8q
    x = csub_neg(s, m.neg_four_q);
    y = mul_add_lazy2<true>(0, d, W, Wp, m.neg_two_q);
  } else if (A::kSmall) {
    const u32 d = (u32)x + (u32)m.two_q - (u32)y;
    x = csub32((u32)s, (u32)m.two_q);
    y = mul_small(d, (u32)W, (u32)Wp, (u32)m.q);
  } else {
    const u64 d = x + m.two_q - y;
    x = csub_wrap(s, m.neg_two_q);
    y = mul_add_strict(0, d, W, Wp, m.neg_q);
  }
}

// Last inverse stage with N^{-1} folded in (ntt-radix-2.cpp:490-509):
// x' = (x+y) * n1, y' = (x-y) * n1W.  The sum needs no conditional subtraction
// first because the lazy products accept it as it is.  Strict: [0,2q).  Harvey60: the doubled
// results lie in [0,4q).
// MONT: the sum branch through scale_by_inverse_degree (N >= 64).
template <class A, bool MONT = false>
HX_HD void inv_butterfly_last(u64& x, u64& y, const InvLast& il, const ModConst& m) {
  static_assert(!A::kLazy && !A::kFp, "Lazy: lazy_inverse.h; Fp64: inv_butterfly_last_fp");
  const u64 s = x + y;
  if (A::kH60) {  // s, d < 8q < 2^63
    const u64 d = x + m.four_q - y;
    x = MONT ? scale_by_inverse_degree(s, il) : mul_add_lazy2<true>(0, s, il.n1, il.n1p, m.neg_two_q);
    y = mul_add_lazy2<true>(0, d, il.n1w, il.n1wp, m.neg_two_q);
  } else if (A::kSmall) {
    const u32 d = (u32)x + (u32)m.two_q - (u32)y;
    x = mul_small((u32)s, (u32)il.n1, (u32)il.n1p, (u32)m.q);
    y = mul_small(d, (u32)il.n1w, (u32)il.n1wp, (u32)m.q);
  } else {
    const u64 d = x + m.two_q - y;
    x = MONT ? scale_by_inverse_degree(s, il) : mul_add_strict(0, s, il.n1, il.n1p, m.neg_q);
    y = mul_add_strict(0, d, il.n1w, il.n1wp, m.neg_q);
  }
}

// End of the inverse network: v is an output of inv_butterfly_last.
template <class A>
HX_HD u64 inv_finish(u64 v, const ModConst& m, bool canonical) {
  if (A::kFp) return fp_canonical(fp_bits_to_double(v), m);
  if (A::kLazy || A::kH60) {
    if (canonical) v = csub_neg(v, m.neg_two_q);
    return v >> 1;
  }
  if (A::kSmall) return canonical ? csub32((u32)v, (u32)m.q) : v;
  return canonical ? csub(v, m.q) : v;
}

}  // namespace hexl_amd
