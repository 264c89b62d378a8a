/* hexl_oracle_avx512.c -- TEST INFRASTRUCTURE ONLY (see hexl_oracle.c).
 *
 * AVX-512 variant of the CPU restatement's two transforms, used only as the
 * `cpu_baseline` of bench.py so that the reported CPU number is an 8-lane SIMD
 * one like the reference's production path (hexl/ntt/fwd-ntt-avx512.cpp,
 * inv-ntt-avx512.cpp), not a scalar one.  It is NOT a copy of that path: the
 * butterfly network, twiddle indexing and ranges are exactly those of
 * ho_ntt_forward_radix2 / ho_ntt_inverse_radix2 (the reference's native
 * algorithm, ntt-radix-2.cpp:17-261, :330-519), executed 8 butterflies at a
 * time: the stages whose butterfly gap is >= 8 straight from memory, the gap-4 /
 * 2 / 1 stages on two vectors at a time with in-register permutations and
 * per-lane twiddles, and -- as the reference's production path does above N =
 * 1024 (fwd-ntt-avx512.cpp:384-403, inv-ntt-avx512.cpp:318-345) -- depth first
 * once a sub-transform fits the L1 data cache (blocks of 2048 coefficients =
 * 16 KiB).  Every intermediate is the same value as in the scalar code, so
 * outputs are bit-identical to it for every (in_mf, out_mf)
 * (tests/test_oracle_kat.py::test_avx512_variant_matches_scalar).
 */
#include <immintrin.h>
#include <string.h>

#include "hexl_oracle.h"

#define HO_AVX512 __attribute__((target("avx512f,avx512dq")))

int ho_has_avx512(void) {
  return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
}

/* high 64 bits of the 8 products a[i] * b[i] (MultiplyUInt64Hi<64>, util/gcc.hpp:50-54) */
HO_AVX512 static inline __m512i mulhi_epu64(__m512i a, __m512i b) {
  const __m512i mask = _mm512_set1_epi64(0xffffffffLL);
  __m512i a_hi = _mm512_srli_epi64(a, 32), b_hi = _mm512_srli_epi64(b, 32);
  __m512i lo_lo = _mm512_mul_epu32(a, b);
  __m512i hi_lo = _mm512_mul_epu32(a_hi, b);
  __m512i lo_hi = _mm512_mul_epu32(a, b_hi);
  __m512i hi_hi = _mm512_mul_epu32(a_hi, b_hi);
  __m512i t = _mm512_add_epi64(hi_lo, _mm512_srli_epi64(lo_lo, 32));
  __m512i u = _mm512_add_epi64(lo_hi, _mm512_and_si512(t, mask));
  return _mm512_add_epi64(_mm512_add_epi64(hi_hi, _mm512_srli_epi64(t, 32)),
                          _mm512_srli_epi64(u, 32));
}

/* MultiplyModLazy<64> (number-theory.hpp:127-141) on 8 lanes, scalar W */
HO_AVX512 static inline __m512i mul_lazy8(__m512i y, __m512i W, __m512i Wp, __m512i q) {
  __m512i Q = mulhi_epu64(y, Wp);
  return _mm512_sub_epi64(_mm512_mullo_epi64(y, W), _mm512_mullo_epi64(Q, q));
}

/* x - m if x >= m else x (x, m < 2^63): unsigned min of x and the wrapped difference */
HO_AVX512 static inline __m512i csub8(__m512i x, __m512i m) {
  return _mm512_min_epu64(x, _mm512_sub_epi64(x, m));
}

static inline uint64_t mul_lazy1(uint64_t y, uint64_t W, uint64_t Wp, uint64_t q) {
  return y * W - ho_mul_hi64(y, Wp) * q;
}

#define HO_BLOCK 2048u /* coefficients of a depth-first block */

HO_AVX512 static inline __m512i idx8(long long a, long long b, long long c, long long d,
                                     long long e, long long f, long long g, long long h) {
  return _mm512_setr_epi64(a, b, c, d, e, f, g, h);
}

/* Per-lane twiddles of a stage with gap t in {1, 2, 4}: the 8 lanes of a vector pair
 * cover 8 / t consecutive groups, each twiddle repeated t times. */
HO_AVX512 static inline __m512i lane_twiddles(const uint64_t* w, uint64_t t) {
  if (t == 1) return _mm512_loadu_si512((const void*)w);
  if (t == 2)
    return _mm512_permutexvar_epi64(idx8(0, 0, 1, 1, 2, 2, 3, 3),
                                    _mm512_zextsi256_si512(_mm256_loadu_si256((const void*)w)));
  return _mm512_permutexvar_epi64(idx8(0, 0, 0, 0, 1, 1, 1, 1),
                                  _mm512_zextsi128_si512(_mm_loadu_si128((const void*)w)));
}

/* Two consecutive vectors A, B (16 coefficients) of a stage with gap t in {1, 2, 4}:
 * X = the first elements of the 8 butterflies, Y = the second ones; and back. */
HO_AVX512 static inline void split_xy(__m512i A, __m512i B, uint64_t t, __m512i* X, __m512i* Y) {
  if (t == 4) {
    *X = _mm512_permutex2var_epi64(A, idx8(0, 1, 2, 3, 8, 9, 10, 11), B);
    *Y = _mm512_permutex2var_epi64(A, idx8(4, 5, 6, 7, 12, 13, 14, 15), B);
  } else if (t == 2) {
    *X = _mm512_permutex2var_epi64(A, idx8(0, 1, 4, 5, 8, 9, 12, 13), B);
    *Y = _mm512_permutex2var_epi64(A, idx8(2, 3, 6, 7, 10, 11, 14, 15), B);
  } else {
    *X = _mm512_permutex2var_epi64(A, idx8(0, 2, 4, 6, 8, 10, 12, 14), B);
    *Y = _mm512_permutex2var_epi64(A, idx8(1, 3, 5, 7, 9, 11, 13, 15), B);
  }
}
HO_AVX512 static inline void merge_xy(__m512i X, __m512i Y, uint64_t t, __m512i* A, __m512i* B) {
  if (t == 4) {
    *A = _mm512_permutex2var_epi64(X, idx8(0, 1, 2, 3, 8, 9, 10, 11), Y);
    *B = _mm512_permutex2var_epi64(X, idx8(4, 5, 6, 7, 12, 13, 14, 15), Y);
  } else if (t == 2) {
    *A = _mm512_permutex2var_epi64(X, idx8(0, 1, 8, 9, 2, 3, 10, 11), Y);
    *B = _mm512_permutex2var_epi64(X, idx8(4, 5, 12, 13, 6, 7, 14, 15), Y);
  } else {
    *A = _mm512_permutex2var_epi64(X, idx8(0, 8, 1, 9, 2, 10, 3, 11), Y);
    *B = _mm512_permutex2var_epi64(X, idx8(4, 12, 5, 13, 6, 14, 7, 15), Y);
  }
}

/* One forward stage (m groups in the whole transform, gap t = n / 2m) over the
 * coefficients [first, first + count) -- whole groups --, reading src, writing dst. */
HO_AVX512 static void fwd_stage(uint64_t* dst, const uint64_t* src, uint64_t first, uint64_t count,
                                uint64_t m, uint64_t t, uint64_t q, const uint64_t* W,
                                const uint64_t* Wp) {
  const uint64_t two_q = q << 1;
  const __m512i vq = _mm512_set1_epi64((long long)q), v2q = _mm512_set1_epi64((long long)two_q);
  const uint64_t g0 = first / (2 * t), groups = count / (2 * t);
  if (t >= 8) {
    for (uint64_t g = 0; g < groups; ++g) {
      const __m512i vW = _mm512_set1_epi64((long long)W[m + g0 + g]);
      const __m512i vWp = _mm512_set1_epi64((long long)Wp[m + g0 + g]);
      const uint64_t off = first + g * 2 * t;
      for (uint64_t j = 0; j < t; j += 8) {
        const uint64_t a = off + j, b = a + t;
        __m512i X = _mm512_loadu_si512((const void*)(src + a));
        __m512i Y = _mm512_loadu_si512((const void*)(src + b));
        __m512i tx = csub8(X, v2q);
        __m512i T = mul_lazy8(Y, vW, vWp, vq);
        _mm512_storeu_si512((void*)(dst + a), _mm512_add_epi64(tx, T));
        _mm512_storeu_si512((void*)(dst + b), _mm512_sub_epi64(_mm512_add_epi64(tx, v2q), T));
      }
    }
    return;
  }
  uint64_t e = 0; /* coefficients done */
  if (count >= 16) {
    for (; e + 16 <= count; e += 16) {
      const uint64_t g = g0 + e / (2 * t);
      __m512i A = _mm512_loadu_si512((const void*)(src + first + e));
      __m512i B = _mm512_loadu_si512((const void*)(src + first + e + 8));
      __m512i X, Y;
      split_xy(A, B, t, &X, &Y);
      const __m512i vW = lane_twiddles(W + m + g, t), vWp = lane_twiddles(Wp + m + g, t);
      __m512i tx = csub8(X, v2q);
      __m512i T = mul_lazy8(Y, vW, vWp, vq);
      merge_xy(_mm512_add_epi64(tx, T), _mm512_sub_epi64(_mm512_add_epi64(tx, v2q), T), t, &A, &B);
      _mm512_storeu_si512((void*)(dst + first + e), A);
      _mm512_storeu_si512((void*)(dst + first + e + 8), B);
    }
  }
  for (; e < count; e += 2 * t) { /* fewer than 16 coefficients: scalar */
    const uint64_t g = g0 + e / (2 * t);
    for (uint64_t j = 0; j < t; ++j) {
      const uint64_t a = first + e + j, b = a + t;
      const uint64_t x = src[a], y = src[b];
      const uint64_t tx = (x >= two_q) ? x - two_q : x;
      const uint64_t T = mul_lazy1(y, W[m + g], Wp[m + g], q);
      dst[a] = tx + T;
      dst[b] = tx + two_q - T;
    }
  }
}

HO_AVX512 void ho_ntt_forward_radix2_avx512(uint64_t* result, const uint64_t* operand,
                                            uint64_t n, uint64_t q,
                                            const uint64_t* root_pows,
                                            const uint64_t* precon_root_pows, uint64_t in_mf,
                                            uint64_t out_mf) {
  (void)in_mf;
  const uint64_t two_q = q << 1;
  const __m512i vq = _mm512_set1_epi64((long long)q), v2q = _mm512_set1_epi64((long long)two_q);
  const uint64_t* src = operand;
  /* breadth first while a group is larger than a block ... */
  uint64_t m = 1, t = n >> 1;
  for (; m < n && 2 * t > HO_BLOCK; m <<= 1, t >>= 1) {
    fwd_stage(result, src, 0, n, m, t, q, root_pows, precon_root_pows);
    src = result;
  }
  /* ... then every block through all remaining stages while it sits in L1 */
  if (m < n) {
    const uint64_t block = 2 * t;
    for (uint64_t first = 0; first < n; first += block) {
      const uint64_t* s = src;
      for (uint64_t mm = m, tt = t; mm < n; mm <<= 1, tt >>= 1) {
        fwd_stage(result, s, first, block, mm, tt, q, root_pows, precon_root_pows);
        s = result;
      }
    }
  }
  if (out_mf == 1) {
    uint64_t i = 0;
    for (; i + 8 <= n; i += 8) {
      __m512i v = _mm512_loadu_si512((const void*)(result + i));
      _mm512_storeu_si512((void*)(result + i), csub8(csub8(v, v2q), vq));
    }
    for (; i < n; ++i) {
      uint64_t v = result[i];
      if (v >= two_q) v -= two_q;
      if (v >= q) v -= q;
      result[i] = v;
    }
  }
}

/* One inverse stage (m groups, gap t = n / 2m; the stage's first twiddle is
 * inv_root_pows[root0], group g uses root0 + g) over [first, first + count). */
HO_AVX512 static void inv_stage(uint64_t* dst, const uint64_t* src, uint64_t first, uint64_t count,
                                uint64_t root0, uint64_t t, uint64_t q, const uint64_t* W,
                                const uint64_t* Wp) {
  const uint64_t two_q = q << 1;
  const __m512i vq = _mm512_set1_epi64((long long)q), v2q = _mm512_set1_epi64((long long)two_q);
  const uint64_t g0 = first / (2 * t), groups = count / (2 * t);
  if (t >= 8) {
    for (uint64_t g = 0; g < groups; ++g) {
      const __m512i vW = _mm512_set1_epi64((long long)W[root0 + g0 + g]);
      const __m512i vWp = _mm512_set1_epi64((long long)Wp[root0 + g0 + g]);
      const uint64_t off = first + g * 2 * t;
      for (uint64_t j = 0; j < t; j += 8) {
        const uint64_t a = off + j, b = a + t;
        __m512i X = _mm512_loadu_si512((const void*)(src + a));
        __m512i Y = _mm512_loadu_si512((const void*)(src + b));
        __m512i s = _mm512_add_epi64(X, Y);
        __m512i d = _mm512_sub_epi64(_mm512_add_epi64(X, v2q), Y);
        _mm512_storeu_si512((void*)(dst + a), csub8(s, v2q));
        _mm512_storeu_si512((void*)(dst + b), mul_lazy8(d, vW, vWp, vq));
      }
    }
    return;
  }
  uint64_t e = 0;
  if (count >= 16) {
    for (; e + 16 <= count; e += 16) {
      const uint64_t g = root0 + g0 + e / (2 * t);
      __m512i A = _mm512_loadu_si512((const void*)(src + first + e));
      __m512i B = _mm512_loadu_si512((const void*)(src + first + e + 8));
      __m512i X, Y;
      split_xy(A, B, t, &X, &Y);
      const __m512i vW = lane_twiddles(W + g, t), vWp = lane_twiddles(Wp + g, t);
      __m512i s = _mm512_add_epi64(X, Y);
      __m512i d = _mm512_sub_epi64(_mm512_add_epi64(X, v2q), Y);
      merge_xy(csub8(s, v2q), mul_lazy8(d, vW, vWp, vq), t, &A, &B);
      _mm512_storeu_si512((void*)(dst + first + e), A);
      _mm512_storeu_si512((void*)(dst + first + e + 8), B);
    }
  }
  for (; e < count; e += 2 * t) {
    const uint64_t g = root0 + g0 + e / (2 * t);
    for (uint64_t j = 0; j < t; ++j) {
      const uint64_t a = first + e + j, b = a + t;
      const uint64_t x = src[a], y = src[b];
      const uint64_t s = x + y, d = x + two_q - y;
      dst[a] = (s >= two_q) ? s - two_q : s;
      dst[b] = mul_lazy1(d, W[g], Wp[g], q);
    }
  }
}

HO_AVX512 void ho_ntt_inverse_radix2_avx512(uint64_t* result, const uint64_t* operand,
                                            uint64_t n, uint64_t q,
                                            const uint64_t* inv_root_pows,
                                            const uint64_t* precon_inv_root_pows,
                                            uint64_t in_mf, uint64_t out_mf) {
  (void)in_mf;
  const uint64_t two_q = q << 1;
  const __m512i vq = _mm512_set1_epi64((long long)q), v2q = _mm512_set1_epi64((long long)two_q);
  const uint64_t n_div_2 = n >> 1;
  /* the stage with m groups (gap t = n / 2m) starts at root index 1 + (n/2 - m) * 2 ... i.e.
   * the stages m = n/2, n/4, ..., 2 use n/2, n/4, ... consecutive twiddles from index 1 on */
  const uint64_t* src = operand;
  const uint64_t block = n < HO_BLOCK ? n : HO_BLOCK;
  /* depth first: every block through the stages whose groups fit in it (not the root stage) */
  uint64_t m_after = n_div_2, t_after = 1, root_after = 1;
  for (uint64_t first = 0; first < n; first += block) {
    const uint64_t* s = src;
    uint64_t m = n_div_2, t = 1, root = 1;
    for (; m > 1 && 2 * t <= block; m >>= 1, t <<= 1) {
      inv_stage(result, s, first, block, root, t, q, inv_root_pows, precon_inv_root_pows);
      s = result;
      root += m;
    }
    m_after = m;
    t_after = t;
    root_after = root;
  }
  if (m_after < n_div_2) src = result;
  /* breadth first for the stages whose groups span several blocks */
  for (uint64_t m = m_after, t = t_after, root = root_after; m > 1; m >>= 1, t <<= 1) {
    inv_stage(result, src, 0, n, root, t, q, inv_root_pows, precon_inv_root_pows);
    src = result;
    root += m;
  }
  if (result != operand && n == 2) memcpy(result, operand, n * sizeof(uint64_t));

  const uint64_t W = inv_root_pows[n - 1];
  const uint64_t inv_n = ho_inverse_mod(n, q);
  const uint64_t inv_n_p = ho_multiply_factor(inv_n, 64, q);
  const uint64_t inv_n_w = ho_multiply_mod(inv_n, W, q);
  const uint64_t inv_n_w_p = ho_multiply_factor(inv_n_w, 64, q);
  uint64_t* X = result;
  uint64_t* Y = X + n_div_2;
  uint64_t j = 0;
  if (n_div_2 >= 8) {
    const __m512i v1 = _mm512_set1_epi64((long long)inv_n), v1p = _mm512_set1_epi64((long long)inv_n_p);
    const __m512i vw = _mm512_set1_epi64((long long)inv_n_w), vwp = _mm512_set1_epi64((long long)inv_n_w_p);
    for (; j + 8 <= n_div_2; j += 8) {
      __m512i x = _mm512_loadu_si512((const void*)(X + j));
      __m512i y = _mm512_loadu_si512((const void*)(Y + j));
      __m512i tx = csub8(_mm512_add_epi64(x, y), v2q); /* AddUIntMod(x, y, 2q) */
      __m512i ty = _mm512_sub_epi64(_mm512_add_epi64(x, v2q), y);
      _mm512_storeu_si512((void*)(X + j), mul_lazy8(tx, v1, v1p, vq));
      _mm512_storeu_si512((void*)(Y + j), mul_lazy8(ty, vw, vwp, vq));
    }
  }
  for (; j < n_div_2; ++j) {
    const uint64_t tx = ho_add_uint_mod(X[j], Y[j], two_q);
    const uint64_t ty = X[j] + two_q - Y[j];
    X[j] = mul_lazy1(tx, inv_n, inv_n_p, q);
    Y[j] = mul_lazy1(ty, inv_n_w, inv_n_w_p, q);
  }
  if (out_mf == 1) {
    uint64_t i = 0;
    for (; i + 8 <= n; i += 8) {
      __m512i v = _mm512_loadu_si512((const void*)(result + i));
      _mm512_storeu_si512((void*)(result + i), csub8(v, vq));
    }
    for (; i < n; ++i) result[i] = result[i] >= q ? result[i] - q : result[i];
  }
}
