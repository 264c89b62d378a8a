// keyswitch_kernels.hip -- the element-wise stages of KeySwitch
// (hexl/experimental/seal/key-switch-internal.cpp:25-201) fused into four
// streaming kernels; the transforms between them are the batched NTT kernels of
// ntt_kernels.hip.  All intermediates stay in device memory: the reference's
// per-modulus loops of InvNTT -> ReduceMod -> FwdNTT(4,4) -> 128-bit MAC ->
// BarrettReduce128 -> FMAMod(8) -> AddMod each round-trip a cache-resident
// polynomial; here every stage between two transforms is one kernel.
//
// Results are canonical ([0, q_i)), hence bit-identical to the reference's even
// though the lazy forward transforms return different representatives.
#include <hip/hip_runtime.h>

#include "internal.h"
#include "modarith.h"

namespace hexl_amd {

// x mod q for any 64-bit x (modarith.h)
__device__ __forceinline__ u64 full_reduce(u64 x, u64 q, u64 barrett) {
  return reduce_any(x, q, barrett);
}

// Work decomposition of every kernel: blockIdx.x strides over the n coefficients,
// blockIdx.y selects the modulus / operand, blockIdx.z the (target, key component).
// No pointer is __restrict__: nothing here is in place, but the buffers are slices of
// one workspace.

// :77-89 operands of the products: for RNS index i (key modulus q[i]) and every
// decomposition modulus j != i the coefficient-form target, reduced to q[i] where
// moduli[j] is larger.  blockIdx.y = s (see internal.h), blockIdx.z = target.
__global__ void __launch_bounds__(256)
ks_gather_kernel(u64* ntt_buf, const u64* t_target, KsDims d, KsGatherAll g) {
  const u32 D = d.decomp, s = blockIdx.y, tgt = blockIdx.z;
  u32 i, j;
  if (s < D * (D - 1)) {
    i = s / (D - 1);
    const u32 r = s - i * (D - 1);
    j = r < i ? r : r + 1;
  } else {
    i = D;
    j = s - D * (D - 1);
  }
  const u64* src = t_target + ((u64)tgt * D + j) * d.n;
  u64* dst = ntt_buf + ((u64)tgt * D * D + s) * d.n;
  const bool reduce = (g.reduce_mask[i] >> j) & 1;
  const u64 q = g.q[i], barrett = g.barrett[i];
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 l = (u64)blockIdx.x * 256 + threadIdx.x; l < d.n; l += stride) {
    const u64 v = src[l];
    dst[l] = reduce ? full_reduce(v, q, barrett) : v;
  }
}

// :94-130 multiply with the keys, accumulate in 128 bits, reduce once.
// blockIdx.y = RNS index i, blockIdx.z = (tile of kMacTargets targets) * ceil(C / 2) + pair
// of key components.  A thread applies the key words of its coefficient -- for two key
// components -- to kMacTargets targets: the keys (D * C * (D + 1) polynomials, the largest
// operand of the whole KeySwitch) are read once per tile of targets instead of once per
// target, and a target's operands once per pair of components instead of once per component.
// (2, 4 and 8 targets per tile measured equal within 2 % of the whole call at 256 targets, n = 16384,
// round 6: the keys come out of the L2 / Infinity Cache either way)
constexpr int kMacTargets = 4;
__global__ void __launch_bounds__(256)
ks_mac_kernel(u64* prod, const u64* t_target_iter, const u64* ntt_buf, KsDims d, KsMacAll m) {
  const u32 D = d.decomp, i = blockIdx.y;
  const u32 pairs = (d.components + 1) >> 1;
  const u32 tile = blockIdx.z / pairs, k0 = (blockIdx.z - tile * pairs) * 2;
  const bool two = k0 + 1 < d.components;
  const u32 t0 = tile * kMacTargets;
  const u32 nt = d.targets - t0 < (u32)kMacTargets ? d.targets - t0 : (u32)kMacTargets;
  const u64 n = d.n;
  const u64 key_off0 = ((u64)k0 * d.key_moduli + m.key_index[i]) * n;
  const u64 key_off1 = key_off0 + (two ? (u64)d.key_moduli * n : 0);
  const u64 buf_off = i < D ? (u64)i * (D - 1) : (u64)D * (D - 1);  // first operand of index i
  const u64 q = m.q[i], barrett = m.barrett[i], two64 = m.two64_mod_q[i], mu = m.mu[i];
  const u32 shift = m.shift[i];
  const u64 hi_limit = m.hi_limit[i];
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 l = (u64)blockIdx.x * 256 + threadIdx.x; l < n; l += stride) {
    u64 lo[2][kMacTargets], hi[2][kMacTargets];
#pragma unroll
    for (int t = 0; t < kMacTargets; ++t) lo[0][t] = hi[0][t] = lo[1][t] = hi[1][t] = 0;
    for (u32 j = 0; j < D; ++j) {
      const u64 b0 = m.keys[j][key_off0 + l];
      const u64 b1 = m.keys[j][key_off1 + l];
      // operand j of RNS index i: the target's own NTT-form polynomial (j == i) or the
      // transformed product operand
      const u64 slot = (j == i) ? 0 : buf_off + (j < i ? j : (i < D ? j - 1 : j));
#pragma unroll
      for (int t = 0; t < kMacTargets; ++t) {
        if ((u32)t < nt) {
          const u64 tgt = t0 + t;
          const u64 a = (j == i) ? t_target_iter[(tgt * D + j) * n + l]
                                 : ntt_buf[(tgt * D * D + slot) * n + l];
          u64 plo = a * b0, phi = __umul64hi(a, b0);
          lo[0][t] += plo;
          hi[0][t] += phi + (lo[0][t] < plo);
          plo = a * b1;
          phi = __umul64hi(a, b1);
          lo[1][t] += plo;
          hi[1][t] += phi + (lo[1][t] < plo);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c == 0 || two) {
#pragma unroll
        for (int t = 0; t < kMacTargets; ++t) {
          if ((u32)t < nt) {
            if (hi[c][t] < hi_limit) {
              // Generalised Barrett with alpha = 62, beta = -2 (MultOp, eltwise_kernels.hip;
              // eltwise-mult-mod-internal.hpp:50-93) on the whole sum S = hi 2^64 + lo: with n = bits(q),
              // mu = floor(2^(n + 62) / q), c1 = S >> (n - 2) < 2^63, the estimate floor(c1 mu / 2^64) is
              // the quotient or one less for every S < 2^(n + 61) (S/q - estimate < 2^(gamma - alpha) +
              // 2^(beta + 1) + 1 <= 2 with gamma = 61) -- hi < 2^(n - 3), checked per element, so operands
              // outside the ranges the reference assumes still get the exact path below.  In range (a < 4q
              // lazy transform outputs, b < q) S < D 2^(2n + 2): every sum of D <= 16 products up to 55-bit
              // moduli.  One estimate, one multiply, one conditional subtraction instead of two single-word
              // reductions, a 128-bit product and a second estimate; the same canonical residue as
              // BarrettReduce128 (util/gcc.hpp:20-28).
              const u64 c1 = (lo[c][t] >> shift) | (hi[c][t] << (64 - shift));  // (hi_limit > 0: 2 <= shift <= 62)
              const u64 r = lo[c][t] - __umul64hi(c1, mu) * q;                   // in [0, 2q)
              prod[(((u64)i * d.targets + t0 + t) * d.components + k0 + c) * n + l] = csub(r, q);
              continue;
            }
            // (hi * 2^64 + lo) mod q exactly (BarrettReduce128, util/gcc.hpp:20-28)
            const u64 r1 = full_reduce(hi[c][t], q, barrett);
            const u64 r2 = full_reduce(lo[c][t], q, barrett);
            const u64 plo = r1 * two64, phi = __umul64hi(r1, two64);
            const u64 c1 = shift ? ((plo >> shift) | (phi << (64 - shift))) : plo;
            const u64 r = csub(plo - __umul64hi(c1, mu) * q, q);
            prod[(((u64)i * d.targets + t0 + t) * d.components + k0 + c) * n + l] = csub(r + r2, q);
          }
        }
      }
    }
  }
}

// :146-175 round the last RNS component (add q_k / 2, reduce mod q_k), bring it to
// every decomposition modulus and add the correction.
// blockIdx.y = i < D, blockIdx.z = target * C + k.
__global__ void __launch_bounds__(256)
ks_round_kernel(u64* tbuf, const u64* prod, KsDims d, KsRound r) {
  const u32 D = d.decomp, i = blockIdx.y;
  const KsRoundMod mi = r.mod[i];
  const u64* t_last = prod + ((u64)D * d.targets * d.components + blockIdx.z) * d.n;
  u64* dst = tbuf + ((u64)blockIdx.z * D + i) * d.n;
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 l = (u64)blockIdx.x * 256 + threadIdx.x; l < d.n; l += stride) {
    const u64 x = t_last[l] + r.qk_half;
    const u64 v = csub(x - __umul64hi(x, r.barrett_k) * r.qk, r.qk);
    dst[l] = (mi.reduce ? full_reduce(v, mi.q, mi.barrett) : v) + mi.fix;
  }
}

// :180-196 (ct mod q_i - ct mod q_k) * q_k^-1 mod q_i, accumulated into the result.
// blockIdx.y = i < D, blockIdx.z = target * C + k.
__global__ void __launch_bounds__(256)
ks_finish_kernel(u64* result, const u64* prod, const u64* tbuf, KsDims d, KsFinish f) {
  const u32 D = d.decomp, i = blockIdx.y;
  const KsFinishMod mi = f.mod[i];
  u64* data = result + ((u64)blockIdx.z * D + i) * d.n;
  const u64* p = prod + ((u64)i * d.targets * d.components + blockIdx.z) * d.n;
  const u64* t = tbuf + ((u64)blockIdx.z * D + i) * d.n;
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 l = (u64)blockIdx.x * 256 + threadIdx.x; l < d.n; l += stride) {
    u64 x = p[l] + (mi.q << 2) - t[l];  // < 8q
    x = csub(x, mi.q << 2);
    x = csub(x, mi.q << 1);
    x = csub(x, mi.q);
    const u64 r = csub(mul_lazy(x, mi.s, mi.sp, mi.q), mi.q);  // FMAMod, input_mod_factor 8
    data[l] = csub(data[l] + r, mi.q);                          // AddMod
  }
}

static unsigned ks_grid(u64 n) {
  const u64 b = (n + 255) / 256;
  return (unsigned)(b < 65535 ? (b ? b : 1) : 65535);
}

hipError_t ks_gather_launch(u64* ntt_buf, const u64* t_target, const KsDims& d,
                            const KsGatherAll& g, hipStream_t st) {
  hipLaunchKernelGGL(ks_gather_kernel, dim3(ks_grid(d.n), d.decomp * d.decomp, d.targets),
                     dim3(256), 0, st, ntt_buf, t_target, d, g);
  return hipGetLastError();
}
hipError_t ks_mac_launch(u64* prod, const u64* t_target_iter, const u64* ntt_buf, const KsDims& d,
                         const KsMacAll& m, hipStream_t st) {
  hipLaunchKernelGGL(ks_mac_kernel,
                     dim3(ks_grid(d.n), d.decomp + 1,
                          ((d.targets + kMacTargets - 1) / kMacTargets) * ((d.components + 1) / 2)),
                     dim3(256), 0, st, prod, t_target_iter, ntt_buf, d, m);
  return hipGetLastError();
}
hipError_t ks_round_launch(u64* tbuf, const u64* prod, const KsDims& d, const KsRound& r,
                           hipStream_t st) {
  hipLaunchKernelGGL(ks_round_kernel, dim3(ks_grid(d.n), d.decomp, d.targets * d.components),
                     dim3(256), 0, st, tbuf, prod, d, r);
  return hipGetLastError();
}
hipError_t ks_finish_launch(u64* result, const u64* prod, const u64* tbuf, const KsDims& d,
                            const KsFinish& f, hipStream_t st) {
  hipLaunchKernelGGL(ks_finish_kernel, dim3(ks_grid(d.n), d.decomp, d.targets * d.components),
                     dim3(256), 0, st, result, prod, tbuf, d, f);
  return hipGetLastError();
}

}  // namespace hexl_amd
