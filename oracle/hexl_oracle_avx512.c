/* hexl_oracle_avx512.c -- TEST INFRASTRUCTURE ONLY (see hexl_oracle.c).
 *
 * AVX-512 variant of the CPU restatement's two transforms, used only as the
 * `cpu_baseline` of bench.py so that the reported CPU number is an 8-lane SIMD
 * one like the reference's production path (hexl/ntt/fwd-ntt-avx512.cpp,
 * inv-ntt-avx512.cpp), not a scalar one.  It is NOT a copy of that path: the
 * butterfly network, twiddle indexing and ranges are exactly those of
 * ho_ntt_forward_radix2 / ho_ntt_inverse_radix2 (the reference's native
 * algorithm, ntt-radix-2.cpp:17-261, :330-519), with the stages whose butterfly
 * gap is >= 8 executed 8 butterflies at a time and the three gap-4/2/1 stages
 * left scalar.  Every intermediate is the same value as in the scalar code, so
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

HO_AVX512 void ho_ntt_forward_radix2_avx512(uint64_t* result, const uint64_t* operand,
                                            uint64_t n, uint64_t q,
                                            const uint64_t* root_pows,
                                            const uint64_t* precon_root_pows, uint64_t in_mf,
                                            uint64_t out_mf) {
  (void)in_mf;
  const uint64_t two_q = q << 1;
  const __m512i vq = _mm512_set1_epi64((long long)q), v2q = _mm512_set1_epi64((long long)two_q);
  uint64_t t = n >> 1;
  const uint64_t* src = operand;
  for (uint64_t m = 1; m < n; m <<= 1) {
    uint64_t offset = 0;
    for (uint64_t i = 0; i < m; ++i) {
      const uint64_t W = root_pows[m + i], Wp = precon_root_pows[m + i];
      if (t >= 8) {
        const __m512i vW = _mm512_set1_epi64((long long)W), vWp = _mm512_set1_epi64((long long)Wp);
        for (uint64_t j = 0; j < t; j += 8) {
          const uint64_t a = offset + j, b = a + t;
          __m512i X = _mm512_loadu_si512((const void*)(src + a));
          __m512i Y = _mm512_loadu_si512((const void*)(src + b));
          __m512i tx = csub8(X, v2q);
          __m512i T = mul_lazy8(Y, vW, vWp, vq);
          _mm512_storeu_si512((void*)(result + a), _mm512_add_epi64(tx, T));
          _mm512_storeu_si512((void*)(result + b),
                              _mm512_sub_epi64(_mm512_add_epi64(tx, v2q), T));
        }
      } else {
        for (uint64_t j = 0; j < t; ++j) {
          const uint64_t a = offset + j, b = a + t;
          const uint64_t x = src[a], y = src[b];
          const uint64_t tx = (x >= two_q) ? x - two_q : x;
          const uint64_t T = mul_lazy1(y, W, Wp, q);
          result[a] = tx + T;
          result[b] = tx + two_q - T;
        }
      }
      offset += (t << 1);
    }
    t >>= 1;
    src = result;
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

HO_AVX512 void ho_ntt_inverse_radix2_avx512(uint64_t* result, const uint64_t* operand,
                                            uint64_t n, uint64_t q,
                                            const uint64_t* inv_root_pows,
                                            const uint64_t* precon_inv_root_pows,
                                            uint64_t in_mf, uint64_t out_mf) {
  (void)in_mf;
  const uint64_t two_q = q << 1;
  const __m512i vq = _mm512_set1_epi64((long long)q), v2q = _mm512_set1_epi64((long long)two_q);
  const uint64_t n_div_2 = n >> 1;
  uint64_t t = 1, root_index = 1;
  const uint64_t* src = operand;
  for (uint64_t m = n_div_2; m > 1; m >>= 1) {
    uint64_t offset = 0;
    for (uint64_t i = 0; i < m; ++i, ++root_index) {
      const uint64_t W = inv_root_pows[root_index], Wp = precon_inv_root_pows[root_index];
      if (t >= 8) {
        const __m512i vW = _mm512_set1_epi64((long long)W), vWp = _mm512_set1_epi64((long long)Wp);
        for (uint64_t j = 0; j < t; j += 8) {
          const uint64_t a = offset + j, b = a + t;
          __m512i X = _mm512_loadu_si512((const void*)(src + a));
          __m512i Y = _mm512_loadu_si512((const void*)(src + b));
          __m512i s = _mm512_add_epi64(X, Y);
          __m512i d = _mm512_sub_epi64(_mm512_add_epi64(X, v2q), Y);
          _mm512_storeu_si512((void*)(result + a), csub8(s, v2q));
          _mm512_storeu_si512((void*)(result + b), mul_lazy8(d, vW, vWp, vq));
        }
      } else {
        for (uint64_t j = 0; j < t; ++j) {
          const uint64_t a = offset + j, b = a + t;
          const uint64_t x = src[a], y = src[b];
          const uint64_t s = x + y, d = x + two_q - y;
          result[a] = (s >= two_q) ? s - two_q : s;
          result[b] = mul_lazy1(d, W, Wp, q);
        }
      }
      offset += (t << 1);
    }
    t <<= 1;
    src = result;
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
