/*
 * hexl_oracle.c -- CPU restatement of the intel/hexl NTT + Eltwise hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under hexl_amd/ (the product) may include,
 * link, import or call this file.  Its only callers are tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke(), and there only as the checker.
 *
 * What it is: a plain-C, scalar, single-thread restatement of the reference's
 * *native* (non-AVX) algorithms, function by function, each citing the
 * reference file:line it follows (paths relative to the reference root).
 *
 * Pinning: the real reference cannot be compiled in this image under the
 * build rules (every translation unit includes the cmake-generated
 * hexl/util/defines.hpp and the git-fetched third-party cpuinfo_x86.h;
 * see DESIGN.md "Oracle"), so this oracle is pinned against every
 * known-answer vector the reference's own tests hold for this path
 * (tests/golden/hexl_kat.json, checked by tests/test_oracle_kat.py) and
 * against an independent O(N^2) big-integer evaluation of the transform's
 * mathematical definition.
 */
#include "hexl_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------ */
/* 128-bit helpers: hexl/include/hexl/util/gcc.hpp:16-59                      */
/* ------------------------------------------------------------------------ */

/* gcc.hpp:50-54 MultiplyUInt64Hi<64> */
uint64_t ho_mul_hi64(uint64_t x, uint64_t y) {
  return (uint64_t)(((u128)x * (u128)y) >> 64);
}

/* gcc.hpp:20-28 BarrettReduce128 (the reference uses a true 128-bit %) */
uint64_t ho_reduce128(uint64_t hi, uint64_t lo, uint64_t modulus) {
  u128 n = ((u128)hi << 64) | (u128)lo;
  return (uint64_t)(n % modulus);
}

/* gcc.hpp:31-37 DivideUInt128UInt64Lo */
uint64_t ho_divide_u128_u64_lo(uint64_t x1, uint64_t x0, uint64_t y) {
  u128 n = ((u128)x1 << 64) | (u128)x0;
  return (uint64_t)(n / y);
}

/* gcc.hpp:57-59 MSB == floor(log2(x)); integer form of the reference's log2l */
uint64_t ho_msb(uint64_t x) {
  uint64_t r = 0;
  while (x >>= 1) ++r;
  return r;
}

/* ------------------------------------------------------------------------ */
/* Scalar number theory                                                      */
/* ------------------------------------------------------------------------ */

/* hexl/include/hexl/number-theory/number-theory.hpp:29-40 MultiplyFactor:
 * floor((operand << bit_shift) / modulus), bit_shift in {32, 52, 64}. */
uint64_t ho_multiply_factor(uint64_t operand, uint64_t bit_shift,
                            uint64_t modulus) {
  uint64_t op_hi = (bit_shift == 64) ? operand : (operand >> (64 - bit_shift));
  uint64_t op_lo = (bit_shift == 64) ? 0 : (operand << bit_shift);
  return ho_divide_u128_u64_lo(op_hi, op_lo, modulus);
}

/* hexl/number-theory/number-theory.cpp:13-42 InverseMod (extended Euclid on
 * signed 64-bit cofactors, made positive at the end). */
uint64_t ho_inverse_mod(uint64_t input, uint64_t modulus) {
  uint64_t a = input % modulus;
  if (modulus == 1) return 0;
  int64_t m0 = (int64_t)modulus;
  int64_t y = 0, x = 1;
  while (a > 1) {
    int64_t q = (int64_t)(a / modulus);
    int64_t t = (int64_t)modulus;
    modulus = a % modulus;
    a = (uint64_t)t;
    t = y;
    y = x - q * y;
    x = t;
  }
  if (x < 0) x += m0;
  return (uint64_t)x;
}

/* number-theory.cpp:44-52 MultiplyMod(x, y, modulus) */
uint64_t ho_multiply_mod(uint64_t x, uint64_t y, uint64_t modulus) {
  u128 p = (u128)x * (u128)y;
  return ho_reduce128((uint64_t)(p >> 64), (uint64_t)p, modulus);
}

/* number-theory.cpp:54-59 MultiplyMod(x, y, y_precon, modulus) */
uint64_t ho_multiply_mod_precon(uint64_t x, uint64_t y, uint64_t y_precon,
                                uint64_t modulus) {
  uint64_t q = ho_mul_hi64(x, y_precon);
  q = x * y - q * modulus;
  return q >= modulus ? q - modulus : q;
}

/* number-theory.hpp:127-141 MultiplyModLazy<64>: result in [0, 2q) */
uint64_t ho_multiply_mod_lazy64(uint64_t x, uint64_t y_operand,
                                uint64_t y_barrett_factor, uint64_t modulus) {
  uint64_t Q = ho_mul_hi64(x, y_barrett_factor);
  return y_operand * x - Q * modulus;
}

/* number-theory.cpp:61-66 */
uint64_t ho_add_uint_mod(uint64_t x, uint64_t y, uint64_t modulus) {
  uint64_t sum = x + y;
  return (sum >= modulus) ? (sum - modulus) : sum;
}

/* number-theory.cpp:68-73 */
uint64_t ho_sub_uint_mod(uint64_t x, uint64_t y, uint64_t modulus) {
  uint64_t diff = (x + modulus) - y;
  return (diff >= modulus) ? (diff - modulus) : diff;
}

/* number-theory.cpp:76-87 PowMod */
uint64_t ho_pow_mod(uint64_t base, uint64_t exp, uint64_t modulus) {
  base %= modulus;
  uint64_t result = 1;
  while (exp > 0) {
    if (exp & 1) result = ho_multiply_mod(result, base, modulus);
    base = ho_multiply_mod(base, base, modulus);
    exp >>= 1;
  }
  return result;
}

/* number-theory.cpp:91-102 IsPrimitiveRoot: root^(degree/2) == -1 */
int ho_is_primitive_root(uint64_t root, uint64_t degree, uint64_t modulus) {
  if (root == 0) return 0;
  return ho_pow_mod(root, degree / 2, modulus) == (modulus - 1);
}

/* number-theory.cpp:106-124 GeneratePrimitiveRoot draws random trial values;
 * number-theory.cpp:128-148 MinimalPrimitiveRoot then takes the minimum over
 * root * (root^2)^i, i = 0..degree-1, i.e. over EVERY primitive degree-th
 * root, so its result does not depend on which root the random search found.
 * The restatement therefore searches candidates 2, 3, 4, ... deterministically
 * and applies the same minimisation loop. */
uint64_t ho_minimal_primitive_root(uint64_t degree, uint64_t modulus) {
  uint64_t quotient = (modulus - 1) / degree;
  uint64_t root = 0;
  for (uint64_t cand = 2; cand < modulus; ++cand) {
    uint64_t r = ho_pow_mod(cand, quotient, modulus);
    if (ho_is_primitive_root(r, degree, modulus)) {
      root = r;
      break;
    }
  }
  if (root == 0) return 0;
  uint64_t generator_sq = ho_multiply_mod(root, root, modulus);
  uint64_t current = root;
  uint64_t min_root = root;
  for (uint64_t i = 0; i < degree; ++i) {
    if (current < min_root) min_root = current;
    current = ho_multiply_mod(current, generator_sq, modulus);
  }
  return min_root;
}

/* number-theory.cpp:150-163 ReverseBits */
uint64_t ho_reverse_bits(uint64_t x, uint64_t bit_width) {
  if (bit_width == 0) return 0;
  uint64_t rev = 0;
  for (uint64_t i = bit_width; i > 0; i--) {
    rev |= ((x & 1) << (i - 1));
    x >>= 1;
  }
  return rev;
}

/* number-theory.cpp:166-212 IsPrime: Miller-Rabin, 12 fixed bases */
int ho_is_prime(uint64_t n) {
  static const uint64_t as[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (int k = 0; k < 12; ++k) {
    if (n == as[k]) return 1;
    if (n % as[k] == 0) return 0;
  }
  uint64_t r = 63;
  while (r > 0) {
    uint64_t two_pow_r = (1ULL << r);
    if ((n - 1) % two_pow_r == 0) break;
    --r;
  }
  uint64_t d = (n - 1) / (1ULL << r);
  for (int k = 0; k < 12; ++k) {
    uint64_t x = ho_pow_mod(as[k], d, n);
    if (x == 1 || x == n - 1) continue;
    int prime = 0;
    for (uint64_t i = 1; i < r; ++i) {
      x = ho_pow_mod(x, 2, n);
      if (x == n - 1) {
        prime = 1;
        break;
      }
    }
    if (!prime) return 0;
  }
  return 1;
}

/* number-theory.cpp:214-261 GeneratePrimes: primes == 1 mod 2*ntt_size in
 * (2^bit_size, 2^(bit_size+1)), ascending from the bottom or descending from
 * the top.  Returns the number found (== num_primes on success). */
size_t ho_generate_primes(uint64_t* out, size_t num_primes, size_t bit_size,
                          int prefer_small_primes, size_t ntt_size) {
  int64_t lower = (1LL << bit_size) + 1LL;
  int64_t upper = (1LL << (bit_size + 1LL)) - 1LL;
  int64_t two_n = 2 * (int64_t)ntt_size;
  int64_t cand = prefer_small_primes ? lower : upper - (upper % two_n) + 1;
  int64_t step = (prefer_small_primes ? 1 : -1) * two_n;
  size_t found = 0;
  while (prefer_small_primes ? (cand < upper) : (cand > lower)) {
    if (ho_is_prime((uint64_t)cand)) {
      out[found++] = (uint64_t)cand;
      if (found == num_primes) return found;
    }
    cand += step;
  }
  return found;
}

/* number-theory.hpp:214-258 ReduceMod<InputModFactor> */
static inline uint64_t reduce_mod_k(uint64_t x, uint64_t q, uint64_t k) {
  if (k >= 8 && x >= 4 * q) x -= 4 * q;
  if (k >= 4 && x >= 2 * q) x -= 2 * q;
  if (k >= 2 && x >= q) x -= q;
  return x;
}

/* number-theory.hpp:195-205 BarrettReduce64<OutputModFactor> */
static inline uint64_t barrett_reduce64(uint64_t input, uint64_t modulus,
                                        uint64_t q_barr, int out_mf) {
  uint64_t q = ho_mul_hi64(input, q_barr);
  uint64_t r = input - q * modulus;
  if (out_mf == 2) return r;
  return (r >= modulus) ? r - modulus : r;
}

/* ------------------------------------------------------------------------ */
/* NTT tables: hexl/ntt/ntt-internal.cpp:54-169                               */
/* ------------------------------------------------------------------------ */

/* root_pows[bitrev(i)] = w^i (:60-72); precon = floor(W * 2^64 / q) (:113-126);
 * inv_root_pows in stage order: [0] = 1, then the inverses of
 * root_pows[m .. 2m-1] for m = N/2, N/4, ..., 1 (:143-154). */
void ho_ntt_tables(uint64_t n, uint64_t q, uint64_t w, uint64_t* root_pows,
                   uint64_t* precon_root_pows, uint64_t* inv_root_pows,
                   uint64_t* precon_inv_root_pows) {
  uint64_t bits = ho_msb(n);
  uint64_t* inv_br = (uint64_t*)malloc(n * sizeof(uint64_t));
  root_pows[0] = 1;
  inv_br[0] = ho_inverse_mod(1, q);
  uint64_t prev = 0;
  for (uint64_t i = 1; i < n; ++i) {
    uint64_t idx = ho_reverse_bits(i, bits);
    root_pows[idx] = ho_multiply_mod(root_pows[prev], w, q);
    inv_br[idx] = ho_inverse_mod(root_pows[idx], q);
    prev = idx;
  }
  inv_root_pows[0] = inv_br[0];
  uint64_t idx = 1;
  for (uint64_t m = n >> 1; m > 0; m >>= 1)
    for (uint64_t i = 0; i < m; ++i) inv_root_pows[idx++] = inv_br[m + i];
  for (uint64_t i = 0; i < n; ++i) {
    precon_root_pows[i] = ho_multiply_factor(root_pows[i], 64, q);
    precon_inv_root_pows[i] = ho_multiply_factor(inv_root_pows[i], 64, q);
  }
  free(inv_br);
}

/* ------------------------------------------------------------------------ */
/* NTT kernels: hexl/ntt/ntt-radix-2.cpp, butterflies hexl/ntt/ntt-default.hpp */
/* ------------------------------------------------------------------------ */

/* ntt-default.hpp:28-42 FwdButterflyRadix2: X, Y in [0,4q) -> [0,4q) */
static inline void fwd_butterfly(uint64_t* xr, uint64_t* yr, uint64_t x,
                                 uint64_t y, uint64_t W, uint64_t Wp,
                                 uint64_t q, uint64_t two_q) {
  uint64_t tx = (x >= two_q) ? x - two_q : x;
  uint64_t T = ho_multiply_mod_lazy64(y, W, Wp, q);
  *xr = tx + T;
  *yr = tx + two_q - T;
}

/* ntt-default.hpp:112-125 InvButterflyRadix2: X, Y in [0,2q) -> [0,2q) */
static inline void inv_butterfly(uint64_t* xr, uint64_t* yr, uint64_t x,
                                 uint64_t y, uint64_t W, uint64_t Wp,
                                 uint64_t q, uint64_t two_q) {
  uint64_t tx = x + y;
  uint64_t ty = x + two_q - y;
  *xr = (tx >= two_q) ? tx - two_q : tx;
  *yr = ho_multiply_mod_lazy64(ty, W, Wp, q);
}

/* ntt-radix-2.cpp:17-261 ForwardTransformToBitReverseRadix2.  The reference
 * unrolls by butterfly gap; the loop nest below is the same iteration space:
 * m = 1,2,..,n/2 groups; group i uses W = root_pows[m + i] (:128-129); the
 * first pass reads `operand` and writes `result` (:40-117), later passes run
 * in place; out_mf == 1 ends with ReduceMod<4> (:254-260). */
void ho_ntt_forward_radix2(uint64_t* result, const uint64_t* operand,
                           uint64_t n, uint64_t q, const uint64_t* root_pows,
                           const uint64_t* precon_root_pows, uint64_t in_mf,
                           uint64_t out_mf) {
  (void)in_mf;
  uint64_t two_q = q << 1;
  uint64_t t = n >> 1;
  const uint64_t* src = operand;
  for (uint64_t m = 1; m < n; m <<= 1) {
    uint64_t offset = 0;
    for (uint64_t i = 0; i < m; ++i) {
      uint64_t W = root_pows[m + i];
      uint64_t Wp = precon_root_pows[m + i];
      for (uint64_t j = 0; j < t; ++j) {
        uint64_t a = offset + j, b = a + t;
        fwd_butterfly(&result[a], &result[b], src[a], src[b], W, Wp, q, two_q);
      }
      offset += (t << 1);
    }
    t >>= 1;
    src = result;
  }
  if (out_mf == 1)
    for (uint64_t i = 0; i < n; ++i) result[i] = reduce_mod_k(result[i], q, 4);
}

/* ntt-radix-2.cpp:330-519 InverseTransformFromBitReverseRadix2: stages
 * t = 1 .. n/4 walk inv_root_pows sequentially (:349-482); n == 2
 * out-of-place copies first (:486-488); the last stage folds N^{-1}
 * (:490-509); out_mf == 1 ends with ReduceMod<2> (:511-518). */
void ho_ntt_inverse_radix2(uint64_t* result, const uint64_t* operand,
                           uint64_t n, uint64_t q,
                           const uint64_t* inv_root_pows,
                           const uint64_t* precon_inv_root_pows, uint64_t in_mf,
                           uint64_t out_mf) {
  (void)in_mf;
  uint64_t two_q = q << 1;
  uint64_t n_div_2 = n >> 1;
  uint64_t t = 1;
  uint64_t root_index = 1;
  const uint64_t* src = operand;
  for (uint64_t m = n_div_2; m > 1; m >>= 1) {
    uint64_t offset = 0;
    for (uint64_t i = 0; i < m; ++i, ++root_index) {
      uint64_t W = inv_root_pows[root_index];
      uint64_t Wp = precon_inv_root_pows[root_index];
      for (uint64_t j = 0; j < t; ++j) {
        uint64_t a = offset + j, b = a + t;
        inv_butterfly(&result[a], &result[b], src[a], src[b], W, Wp, q, two_q);
      }
      offset += (t << 1);
    }
    t <<= 1;
    src = result;
  }
  if (result != operand && n == 2) memcpy(result, operand, n * sizeof(uint64_t));

  uint64_t W = inv_root_pows[n - 1];
  uint64_t inv_n = ho_inverse_mod(n, q);
  uint64_t inv_n_precon = ho_multiply_factor(inv_n, 64, q);
  uint64_t inv_n_w = ho_multiply_mod(inv_n, W, q);
  uint64_t inv_n_w_precon = ho_multiply_factor(inv_n_w, 64, q);
  uint64_t* X = result;
  uint64_t* Y = X + n_div_2;
  for (uint64_t j = 0; j < n_div_2; ++j) {
    uint64_t tx = ho_add_uint_mod(X[j], Y[j], two_q);
    uint64_t ty = X[j] + two_q - Y[j];
    X[j] = ho_multiply_mod_lazy64(tx, inv_n, inv_n_precon, q);
    Y[j] = ho_multiply_mod_lazy64(ty, inv_n_w, inv_n_w_precon, q);
  }
  if (out_mf == 1)
    for (uint64_t i = 0; i < n; ++i) result[i] = reduce_mod_k(result[i], q, 2);
}

/* ------------------------------------------------------------------------ */
/* hexl/ntt/ntt-radix-4.cpp: the reference's second native implementation.  A */
/* radix-4 butterfly is four radix-2 butterflies (ntt-default.hpp:62-100,      */
/* :127-157), so every value equals the radix-2 network's; the reference's     */
/* tests assert radix-4 == radix-2 (test/test-ntt.cpp:406-478).                 */
/* ------------------------------------------------------------------------ */

static int is_power_of_four(uint64_t n) {
  return n && !(n & (n - 1)) && (n & 0x5555555555555555ULL);
}

/* ntt-default.hpp:62-100 FwdButterflyRadix4 on (x0, x1, x2, x3), in place */
static inline void fwd_butterfly4(uint64_t* x0, uint64_t* x1, uint64_t* x2, uint64_t* x3,
                                  const uint64_t* W, const uint64_t* Wp, uint64_t i1,
                                  uint64_t q, uint64_t two_q) {
  const uint64_t i2 = 2 * i1, i3 = 2 * i1 + 1;
  fwd_butterfly(x0, x2, *x0, *x2, W[i1], Wp[i1], q, two_q);
  fwd_butterfly(x1, x3, *x1, *x3, W[i1], Wp[i1], q, two_q);
  fwd_butterfly(x0, x1, *x0, *x1, W[i2], Wp[i2], q, two_q);
  fwd_butterfly(x2, x3, *x2, *x3, W[i3], Wp[i3], q, two_q);
}

/* ntt-radix-4.cpp:17-400 ForwardTransformToBitReverseRadix4 */
void ho_ntt_forward_radix4(uint64_t* result, const uint64_t* operand, uint64_t n, uint64_t q,
                           const uint64_t* root_pows, const uint64_t* precon_root_pows,
                           uint64_t in_mf, uint64_t out_mf) {
  (void)in_mf;
  const uint64_t two_q = q << 1;
  const int pow4 = is_power_of_four(n);
  uint64_t m_start, t;
  if (!pow4) { /* :47-71 a radix-2 step first when log2(n) is odd */
    t = n >> 1;
    for (uint64_t j = 0; j < t; ++j)
      fwd_butterfly(&result[j], &result[j + t], operand[j], operand[j + t], root_pows[1],
                    precon_root_pows[1], q, two_q);
    m_start = 2;
    t = n >> 3;
  } else { /* :75-199 the first radix-4 pass reads the operand */
    t = n >> 2;
    for (uint64_t j = 0; j < t; ++j) {
      uint64_t a = operand[j], b = operand[j + t], c = operand[j + 2 * t], d = operand[j + 3 * t];
      fwd_butterfly4(&a, &b, &c, &d, root_pows, precon_root_pows, 1, q, two_q);
      result[j] = a;
      result[j + t] = b;
      result[j + 2 * t] = c;
      result[j + 3 * t] = d;
    }
    m_start = 4;
    t >>= 2;
  }
  for (uint64_t m = m_start; m < n; m <<= 2) { /* :204-384 */
    for (uint64_t i = 0; i < m; ++i) {
      uint64_t* x = result + i * 4 * t;
      for (uint64_t j = 0; j < t; ++j)
        fwd_butterfly4(&x[j], &x[j + t], &x[j + 2 * t], &x[j + 3 * t], root_pows,
                       precon_root_pows, m + i, q, two_q);
    }
    t >>= 2;
  }
  if (out_mf == 1) /* :386-397 */
    for (uint64_t i = 0; i < n; ++i) result[i] = reduce_mod_k(result[i], q, 4);
}

/* ntt-default.hpp:127-157 InvButterflyRadix4 */
static inline void inv_butterfly4(uint64_t* x0, uint64_t* x1, uint64_t* x2, uint64_t* x3,
                                  const uint64_t* W, const uint64_t* Wp, uint64_t i1,
                                  uint64_t i2, uint64_t i3, uint64_t q, uint64_t two_q) {
  inv_butterfly(x0, x1, *x0, *x1, W[i1], Wp[i1], q, two_q);
  inv_butterfly(x2, x3, *x2, *x3, W[i2], Wp[i2], q, two_q);
  inv_butterfly(x0, x2, *x0, *x2, W[i3], Wp[i3], q, two_q);
  inv_butterfly(x1, x3, *x1, *x3, W[i3], Wp[i3], q, two_q);
}

/* ntt-radix-4.cpp:402-700 InverseTransformFromBitReverseRadix4 */
void ho_ntt_inverse_radix4(uint64_t* result, const uint64_t* operand, uint64_t n, uint64_t q,
                           const uint64_t* inv_root_pows, const uint64_t* precon_inv_root_pows,
                           uint64_t in_mf, uint64_t out_mf) {
  (void)in_mf;
  const uint64_t two_q = q << 1, n_div_2 = n >> 1;
  const int pow4 = is_power_of_four(n);
  const uint64_t* src = operand;
  if (pow4) { /* :423-442 a radix-2 step first when log2(n) is even */
    for (uint64_t j = 0; j < n_div_2; ++j)
      inv_butterfly(&result[2 * j], &result[2 * j + 1], operand[2 * j], operand[2 * j + 1],
                    inv_root_pows[1 + j], precon_inv_root_pows[1 + j], q, two_q);
    src = result;
  }
  uint64_t t = pow4 ? 2 : 1;
  uint64_t w1 = 1 + (pow4 ? n_div_2 : 0);               /* :447 */
  uint64_t w3 = n_div_2 + 1 + (pow4 ? (n >> 2) : 0);    /* :448 */
  for (uint64_t m = n >> (pow4 ? 3 : 2); m > 0; m >>= 2) { /* :452-650 */
    for (uint64_t i = 0; i < m; ++i) {
      const uint64_t off = i * 4 * t;
      const uint64_t i1 = w1++, i2 = w1++, i3 = w3++;
      for (uint64_t j = 0; j < t; ++j) {
        uint64_t a = src[off + j], b = src[off + j + t], c = src[off + j + 2 * t],
                 d = src[off + j + 3 * t];
        inv_butterfly4(&a, &b, &c, &d, inv_root_pows, precon_inv_root_pows, i1, i2, i3, q, two_q);
        result[off + j] = a;
        result[off + j + t] = b;
        result[off + j + 2 * t] = c;
        result[off + j + 3 * t] = d;
      }
    }
    src = result;
    t <<= 2;
    w1 += m;
    w3 += m / 2;
  }
  if (result != operand && n == 2) memcpy(result, operand, n * sizeof(uint64_t)); /* :654-656 */
  /* :659-680 the last stage with N^-1 folded in, as in the radix-2 transform */
  const uint64_t W = inv_root_pows[n - 1];
  const uint64_t inv_n = ho_inverse_mod(n, q);
  const uint64_t inv_n_precon = ho_multiply_factor(inv_n, 64, q);
  const uint64_t inv_n_w = ho_multiply_mod(inv_n, W, q);
  const uint64_t inv_n_w_precon = ho_multiply_factor(inv_n_w, 64, q);
  uint64_t* X = result;
  uint64_t* Y = X + n_div_2;
  for (uint64_t j = 0; j < n_div_2; ++j) {
    const uint64_t tx = ho_add_uint_mod(X[j], Y[j], two_q);
    const uint64_t ty = X[j] + two_q - Y[j];
    X[j] = ho_multiply_mod_lazy64(tx, inv_n, inv_n_precon, q);
    Y[j] = ho_multiply_mod_lazy64(ty, inv_n_w, inv_n_w_precon, q);
  }
  if (out_mf == 1) /* :682-690 */
    for (uint64_t i = 0; i < n; ++i) result[i] = reduce_mod_k(result[i], q, 2);
}

/* ntt-radix-2.cpp:263-293 ReferenceForwardTransformToBitReverse */
void ho_ntt_forward_reference(uint64_t* operand, uint64_t n, uint64_t q,
                              const uint64_t* root_pows) {
  uint64_t t = n >> 1;
  for (uint64_t m = 1; m < n; m <<= 1) {
    uint64_t offset = 0;
    for (uint64_t i = 0; i < m; ++i) {
      uint64_t W = root_pows[m + i];
      for (uint64_t j = offset; j < offset + t; ++j) {
        uint64_t tx = operand[j];
        uint64_t wy = ho_multiply_mod(operand[j + t], W, q);
        operand[j] = ho_add_uint_mod(tx, wy, q);
        operand[j + t] = ho_sub_uint_mod(tx, wy, q);
      }
      offset += (t << 1);
    }
    t >>= 1;
  }
}

/* ntt-radix-2.cpp:295-328 ReferenceInverseTransformFromBitReverse */
void ho_ntt_inverse_reference(uint64_t* operand, uint64_t n, uint64_t q,
                              const uint64_t* inv_root_pows) {
  uint64_t t = 1;
  uint64_t root_index = 1;
  for (uint64_t m = n >> 1; m >= 1; m >>= 1) {
    uint64_t offset = 0;
    for (uint64_t i = 0; i < m; ++i, ++root_index) {
      uint64_t W = inv_root_pows[root_index];
      for (uint64_t j = offset; j < offset + t; ++j) {
        uint64_t x = operand[j], y = operand[j + t];
        operand[j] = ho_add_uint_mod(x, y, q);
        operand[j + t] = ho_multiply_mod(W, ho_sub_uint_mod(x, y, q), q);
      }
      offset += (t << 1);
    }
    t <<= 1;
  }
  uint64_t inv_n = ho_inverse_mod(n, q);
  for (uint64_t i = 0; i < n; ++i)
    operand[i] = ho_multiply_mod(operand[i], inv_n, q);
}

/* ------------------------------------------------------------------------ */
/* Element-wise ops (native paths)                                           */
/* ------------------------------------------------------------------------ */

/* hexl/eltwise/eltwise-add-mod.cpp:16-43 */
void ho_eltwise_add_mod(uint64_t* result, const uint64_t* a, const uint64_t* b,
                        uint64_t n, uint64_t q) {
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t sum = a[i] + b[i];
    result[i] = (sum >= q) ? sum - q : sum;
  }
}

/* eltwise-add-mod.cpp:45-69 (scalar form compares against q - b) */
void ho_eltwise_add_mod_scalar(uint64_t* result, const uint64_t* a, uint64_t b,
                               uint64_t n, uint64_t q) {
  uint64_t diff = q - b;
  for (uint64_t i = 0; i < n; ++i)
    result[i] = (a[i] >= diff) ? a[i] - diff : a[i] + b;
}

/* hexl/eltwise/eltwise-sub-mod.cpp:15-42 */
void ho_eltwise_sub_mod(uint64_t* result, const uint64_t* a, const uint64_t* b,
                        uint64_t n, uint64_t q) {
  for (uint64_t i = 0; i < n; ++i)
    result[i] = (a[i] >= b[i]) ? a[i] - b[i] : a[i] + q - b[i];
}

/* eltwise-sub-mod.cpp:44-65 */
void ho_eltwise_sub_mod_scalar(uint64_t* result, const uint64_t* a, uint64_t b,
                               uint64_t n, uint64_t q) {
  for (uint64_t i = 0; i < n; ++i)
    result[i] = (a[i] >= b) ? a[i] - b : a[i] + q - b;
}

/* hexl/eltwise/eltwise-mult-mod-internal.hpp:34-100 EltwiseMultModNative:
 * generalised Barrett with alpha = 62, beta = -2. */
void ho_eltwise_mult_mod(uint64_t* result, const uint64_t* a,
                         const uint64_t* b, uint64_t n, uint64_t q,
                         uint64_t in_mf) {
  const int64_t beta = -2;
  const int64_t alpha = 62;
  uint64_t ceil_log_mod = ho_msb(q) + 1;
  uint64_t prod_right_shift = ceil_log_mod + beta;
  uint64_t barr_lo =
      ho_multiply_factor(1ULL << (ceil_log_mod + alpha - 64), 64, q);
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t x = reduce_mod_k(a[i], q, in_mf);
    uint64_t y = reduce_mod_k(b[i], q, in_mf);
    u128 prod = (u128)x * (u128)y;
    uint64_t prod_hi = (uint64_t)(prod >> 64), prod_lo = (uint64_t)prod;
    uint64_t c1 = (prod_lo >> prod_right_shift) +
                  (prod_hi << (64 - prod_right_shift));
    uint64_t q_hat = ho_mul_hi64(c1, barr_lo);
    uint64_t Z = prod_lo - q_hat * q;
    result[i] = (Z >= q) ? (Z - q) : Z;
  }
}

/* hexl/eltwise/eltwise-fma-mod-internal.hpp:12-39 EltwiseFMAModNative */
void ho_eltwise_fma_mod(uint64_t* result, const uint64_t* arg1, uint64_t arg2,
                        const uint64_t* arg3, uint64_t n, uint64_t q,
                        uint64_t in_mf) {
  arg2 = reduce_mod_k(arg2, q, in_mf);
  uint64_t precon = ho_multiply_factor(arg2, 64, q);
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t v1 = reduce_mod_k(arg1[i], q, in_mf);
    uint64_t r = ho_multiply_mod_precon(v1, arg2, precon, q);
    if (arg3) {
      uint64_t v3 = reduce_mod_k(arg3[i], q, in_mf);
      r = ho_add_uint_mod(r, v3, q);
    }
    result[i] = r;
  }
}

/* hexl/eltwise/eltwise-reduce-mod.cpp:16-79 EltwiseReduceModNative and the
 * in == out copy of the public wrapper (:94-99).  in_mf == q means "arbitrary
 * 64-bit input": single-word Barrett. */
void ho_eltwise_reduce_mod(uint64_t* result, const uint64_t* operand,
                           uint64_t n, uint64_t q, uint64_t in_mf,
                           uint64_t out_mf) {
  if (in_mf == out_mf && operand != result) {
    for (uint64_t i = 0; i < n; ++i) result[i] = operand[i];
    return;
  }
  uint64_t barrett_factor = ho_multiply_factor(1, 64, q);
  uint64_t two_q = q << 1;
  if (in_mf == q) {
    for (uint64_t i = 0; i < n; ++i)
      result[i] = (operand[i] >= q)
                      ? barrett_reduce64(operand[i], q, barrett_factor,
                                         (int)out_mf)
                      : operand[i];
  }
  if (in_mf == 2)
    for (uint64_t i = 0; i < n; ++i) result[i] = reduce_mod_k(operand[i], q, 2);
  if (in_mf == 4) {
    if (out_mf == 1)
      for (uint64_t i = 0; i < n; ++i)
        result[i] = reduce_mod_k(operand[i], q, 4);
    if (out_mf == 2)
      for (uint64_t i = 0; i < n; ++i)
        result[i] = reduce_mod_k(operand[i], two_q, 2);
  }
}

/* hexl/include/hexl/util/util.hpp:16-25 enum class CMPINT and
 * hexl/util/util-internal.hpp:16-41 Compare(): EQ 0, LT 1, LE 2, FALSE 3, NE 4,
 * NLT 5, NLE 6, TRUE 7 (anything else: true, :39-40). */
static int compare_cmpint(int cmp, uint64_t lhs, uint64_t rhs) {
  switch (cmp) {
    case 0: return lhs == rhs;
    case 1: return lhs < rhs;
    case 2: return lhs <= rhs;
    case 3: return 0;
    case 4: return lhs != rhs;
    case 5: return lhs >= rhs;
    case 6: return lhs > rhs;
    case 7: return 1;
  }
  return 1;
}

/* hexl/eltwise/eltwise-cmp-add.cpp:32-106 EltwiseCmpAddNative:
 * result[i] = cmp(operand1[i], bound) ? operand1[i] + diff : operand1[i]
 * (plain 64-bit addition, no modulus). */
void ho_eltwise_cmp_add(uint64_t* result, const uint64_t* operand1, uint64_t n,
                        int cmp, uint64_t bound, uint64_t diff) {
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t op = operand1[i];
    result[i] = compare_cmpint(cmp, op, bound) ? op + diff : op;
  }
}

/* hexl/eltwise/eltwise-cmp-sub-mod.cpp:47-66 EltwiseCmpSubModNative: the
 * comparison sees the unreduced word, the word is then reduced with a true
 * `% modulus` (:60) and diff is subtracted mod modulus where the comparison
 * held (SubUIntMod, number-theory.cpp:68-73).  modulus > 1 (any value, not
 * necessarily prime), 0 < diff < modulus. */
void ho_eltwise_cmp_sub_mod(uint64_t* result, const uint64_t* operand1,
                            uint64_t n, uint64_t modulus, int cmp,
                            uint64_t bound, uint64_t diff) {
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t op = operand1[i];
    int op_cmp = compare_cmpint(cmp, op, bound);
    op %= modulus;
    if (op_cmp) op = ho_sub_uint_mod(op, diff, modulus);
    result[i] = op;
  }
}

/* hexl/experimental/seal/dyadic-multiply-internal.cpp:17-74 DyadicMultiply:
 * (x0, x1) * (y0, y1) -> (x0*y0, x0*y1 + x1*y0, x1*y1) per RNS modulus, each
 * polynomial n * num_moduli words.  Same call order as the reference (result
 * may alias operand1/operand2: poly 2 is written first, poly 0 last) and the
 * same tiling: tiles of min(n, 512) coefficients, n / tile whole tiles (:33-34;
 * a remainder, only possible when n > 512 is not a multiple of 512, is left
 * untouched exactly as there). */
void ho_dyadic_multiply(uint64_t* result, const uint64_t* operand1,
                        const uint64_t* operand2, uint64_t n,
                        const uint64_t* moduli, uint64_t num_moduli) {
  uint64_t poly_size = n * num_moduli;
  uint64_t tile_size = n < 512 ? n : 512;
  uint64_t num_tiles = n / tile_size;
  uint64_t* temp = (uint64_t*)malloc(tile_size * sizeof(uint64_t));
  for (uint64_t i = 0; i < num_moduli; ++i) {
    for (uint64_t tile = 0; tile < num_tiles; ++tile) {
      uint64_t p0 = i * n + tile_size * tile;
      uint64_t p1 = p0 + poly_size;
      uint64_t p2 = p0 + 2 * poly_size;
      ho_eltwise_mult_mod(result + p2, operand1 + p1, operand2 + p1, tile_size,
                          moduli[i], 1);
      ho_eltwise_mult_mod(temp, operand1 + p1, operand2 + p0, tile_size,
                          moduli[i], 1);
      ho_eltwise_mult_mod(result + p1, operand1 + p0, operand2 + p1, tile_size,
                          moduli[i], 1);
      ho_eltwise_add_mod(result + p1, temp, result + p1, tile_size, moduli[i]);
      ho_eltwise_mult_mod(result + p0, operand1 + p0, operand2 + p0, tile_size,
                          moduli[i], 1);
    }
  }
  free(temp);
}

/* ------------------------------------------------------------------------ */
/* Convenience used by tests / bench: a complete plan in one allocation.     */
/* ------------------------------------------------------------------------ */

ho_ntt* ho_ntt_create(uint64_t n, uint64_t q, uint64_t root) {
  ho_ntt* p = (ho_ntt*)calloc(1, sizeof(ho_ntt));
  p->n = n;
  p->q = q;
  /* ntt-internal.cpp:51-52: default root = MinimalPrimitiveRoot(2N, q) */
  p->w = root ? root : ho_minimal_primitive_root(2 * n, q);
  p->root_pows = (uint64_t*)malloc(4 * n * sizeof(uint64_t));
  p->precon_root_pows = p->root_pows + n;
  p->inv_root_pows = p->root_pows + 2 * n;
  p->precon_inv_root_pows = p->root_pows + 3 * n;
  ho_ntt_tables(n, q, p->w, p->root_pows, p->precon_root_pows,
                p->inv_root_pows, p->precon_inv_root_pows);
  return p;
}

void ho_ntt_destroy(ho_ntt* p) {
  if (!p) return;
  free(p->root_pows);
  free(p);
}

void ho_ntt_forward(const ho_ntt* p, uint64_t* result, const uint64_t* operand,
                    uint64_t in_mf, uint64_t out_mf) {
  ho_ntt_forward_radix2(result, operand, p->n, p->q, p->root_pows,
                        p->precon_root_pows, in_mf, out_mf);
}

void ho_ntt_inverse(const ho_ntt* p, uint64_t* result, const uint64_t* operand,
                    uint64_t in_mf, uint64_t out_mf) {
  ho_ntt_inverse_radix2(result, operand, p->n, p->q, p->inv_root_pows,
                        p->precon_inv_root_pows, in_mf, out_mf);
}

/* Batched helpers: `batch` polynomials back to back (stride n). */
void ho_ntt_forward_batch(const ho_ntt* p, uint64_t* result,
                          const uint64_t* operand, uint64_t batch,
                          uint64_t in_mf, uint64_t out_mf) {
  for (uint64_t b = 0; b < batch; ++b)
    ho_ntt_forward(p, result + b * p->n, operand + b * p->n, in_mf, out_mf);
}

void ho_ntt_inverse_batch(const ho_ntt* p, uint64_t* result,
                          const uint64_t* operand, uint64_t batch,
                          uint64_t in_mf, uint64_t out_mf) {
  for (uint64_t b = 0; b < batch; ++b)
    ho_ntt_inverse(p, result + b * p->n, operand + b * p->n, in_mf, out_mf);
}

/* hexl/experimental/seal/key-switch-internal.cpp:25-201 KeySwitch (CKKS path,
 * root_of_unity_powers_ptr == nullptr).  Restated step by step with the same
 * intermediate ranges; GetNTT(n, q) (ntt-cache.hpp:27-53) becomes a plan per
 * modulus built up front.
 *   n = coeff_count; t_target_iter: decomp x n words (NTT form);
 *   k_switch_keys[j]: key_component_count x key_modulus_size x n words;
 *   result: key_component_count x decomp x n words, accumulated into. */
void ho_key_switch(uint64_t* result, const uint64_t* t_target_iter, uint64_t n,
                   uint64_t decomp_modulus_size, uint64_t key_modulus_size,
                   uint64_t rns_modulus_size, uint64_t key_component_count,
                   const uint64_t* moduli, const uint64_t* const* k_switch_keys,
                   const uint64_t* modswitch_factors) {
  const uint64_t D = decomp_modulus_size, K = key_modulus_size,
                 R = rns_modulus_size, C = key_component_count;
  ho_ntt** plans = (ho_ntt**)calloc(K, sizeof(ho_ntt*));
  for (uint64_t i = 0; i < K; ++i) plans[i] = ho_ntt_create(n, moduli[i], 0);

  /* :38-56 copy of the target, back to coefficient form per decomposition
   * modulus: ComputeInverse(2, 1) */
  uint64_t* t_target = (uint64_t*)malloc(D * n * sizeof(uint64_t));
  memcpy(t_target, t_target_iter, D * n * sizeof(uint64_t));
  uint64_t* t_ntt = (uint64_t*)calloc(n, sizeof(uint64_t));
  for (uint64_t j = 0; j < D; ++j)
    ho_ntt_inverse(plans[j], t_target + j * n, t_target + j * n, 2, 1);

  uint64_t* t_poly_prod = (uint64_t*)calloc(C * n * R, sizeof(uint64_t));
  unsigned __int128* acc =
      (unsigned __int128*)malloc(C * n * sizeof(unsigned __int128));

  for (uint64_t i = 0; i < R; ++i) { /* :61-131 */
    uint64_t key_index = (i == D ? K - 1 : i);
    uint64_t qk = moduli[key_index];
    for (uint64_t x = 0; x < C * n; ++x) acc[x] = 0;
    for (uint64_t j = 0; j < D; ++j) {
      const uint64_t* t_operand;
      if (i == j) {
        t_operand = t_target_iter + j * n; /* :72-73 already in NTT form */
      } else {
        if (moduli[j] <= qk) { /* :77-80 */
          memcpy(t_ntt, t_target + j * n, n * sizeof(uint64_t));
        } else { /* :82-86 ReduceMod(in = modulus, out = 1) */
          ho_eltwise_reduce_mod(t_ntt, t_target + j * n, n, qk, qk, 1);
        }
        ho_ntt_forward(plans[key_index], t_ntt, t_ntt, 4, 4); /* :89 lazy */
        t_operand = t_ntt;
      }
      /* :94-115 128-bit multiply-accumulate, no reduction */
      for (uint64_t k = 0; k < C; ++k)
        for (uint64_t l = 0; l < n; ++l)
          acc[k * n + l] += (unsigned __int128)t_operand[l] *
                            k_switch_keys[j][n * key_index + k * K * n + l];
    }
    /* :122-130 BarrettReduce128 == exact remainder (util/gcc.hpp:20-28) */
    for (uint64_t k = 0; k < C; ++k)
      for (uint64_t l = 0; l < n; ++l)
        t_poly_prod[i * n + n * R * k + l] = (uint64_t)(acc[k * n + l] % qk);
  }

  for (uint64_t kc = 0; kc < C; ++kc) { /* :134-197 */
    uint64_t* prod = t_poly_prod + kc * n * R;
    uint64_t* t_last = prod + D * n;
    uint64_t qk = moduli[K - 1];
    uint64_t qk_half = qk >> 1;
    ho_ntt_inverse(plans[K - 1], t_last, t_last, 2, 2); /* :140-141 */
    uint64_t bf_k = ho_multiply_factor(1, 64, qk);
    for (uint64_t l = 0; l < n; ++l) /* :146-151 */
      t_last[l] = barrett_reduce64(t_last[l] + qk_half, qk, bf_k, 1);
    for (uint64_t i = 0; i < D; ++i) {
      uint64_t qi = moduli[i];
      if (qk > qi) { /* :159-167 */
        ho_eltwise_reduce_mod(t_ntt, t_last, n, qi, qi, 1);
      } else {
        memcpy(t_ntt, t_last, n * sizeof(uint64_t));
      }
      uint64_t bf_i = ho_multiply_factor(1, 64, qi);
      uint64_t fix = qi - barrett_reduce64(qk_half, qi, bf_i, 1); /* :170-175 */
      for (uint64_t l = 0; l < n; ++l) t_ntt[l] += fix;
      ho_ntt_forward(plans[i], t_ntt, t_ntt, 4, 4); /* :178 */
      uint64_t qi_lazy = qi << 2;                   /* :180 */
      uint64_t* t_ith = prod + i * n;
      for (uint64_t l = 0; l < n; ++l) /* :183-186 */
        t_ith[l] = t_ith[l] + qi_lazy - t_ntt[l];
      ho_eltwise_fma_mod(t_ith, t_ith, modswitch_factors[i], NULL, n, qi,
                         8); /* :189-190 */
      uint64_t* data = result + n * (D * kc + i);
      ho_eltwise_add_mod(data, data, t_ith, n, qi); /* :195-196 */
    }
  }
  free(acc);
  free(t_poly_prod);
  free(t_ntt);
  free(t_target);
  for (uint64_t i = 0; i < K; ++i) ho_ntt_destroy(plans[i]);
  free(plans);
}

void ho_ntt_forward_batch_avx512(const ho_ntt* p, uint64_t* result,
                                 const uint64_t* operand, uint64_t batch,
                                 uint64_t in_mf, uint64_t out_mf) {
  for (uint64_t b = 0; b < batch; ++b)
    ho_ntt_forward_radix2_avx512(result + b * p->n, operand + b * p->n, p->n,
                                 p->q, p->root_pows, p->precon_root_pows,
                                 in_mf, out_mf);
}

void ho_ntt_inverse_batch_avx512(const ho_ntt* p, uint64_t* result,
                                 const uint64_t* operand, uint64_t batch,
                                 uint64_t in_mf, uint64_t out_mf) {
  for (uint64_t b = 0; b < batch; ++b)
    ho_ntt_inverse_radix2_avx512(result + b * p->n, operand + b * p->n, p->n,
                                 p->q, p->inv_root_pows,
                                 p->precon_inv_root_pows, in_mf, out_mf);
}

/* splitmix64 stream shared by tests, bench and the device-side generator:
 * coefficient i of polynomial `seed` = next() mod bound. */
void ho_fill_splitmix(uint64_t* out, uint64_t n, uint64_t seed,
                      uint64_t bound) {
  uint64_t s = seed;
  for (uint64_t i = 0; i < n; ++i) {
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    out[i] = bound ? z % bound : z;
  }
}
