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

// x mod q for any 64-bit x: single-word Barrett, floor(2^64 / q)
// (eltwise-reduce-mod.cpp:32-55 with input_mod_factor == modulus)
__device__ __forceinline__ u64 full_reduce(u64 x, u64 q, u64 barrett) {
  if (x < q) return x;
  return csub(x - __umul64hi(x, barrett) * q, q);
}

// :77-89 operands of the products for RNS index i: slot s holds decomposition
// modulus jmap[s], reduced to the key modulus where that one is smaller.
__global__ void __launch_bounds__(256)
ks_gather_kernel(u64* out, const u64* t_target, u64 n, KsGather g) {
  const u32 s = blockIdx.y;
  const u64* src = t_target + (u64)g.jmap[s] * n;
  u64* dst = out + (u64)s * n;
  const bool reduce = (g.reduce_mask >> s) & 1;
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 l = (u64)blockIdx.x * 256 + threadIdx.x; l < n; l += stride) {
    const u64 v = src[l];
    dst[l] = reduce ? full_reduce(v, g.q, g.barrett) : v;
  }
}

// :94-130 multiply with the keys, accumulate in 128 bits, reduce once.
// blockIdx.y = key component k.
__global__ void __launch_bounds__(256)
ks_mac_kernel(u64* prod, const u64* t_target_iter, const u64* ntt_buf, u64 n, KsMac m) {
  const u32 k = blockIdx.y;
  const u64 key_off = (u64)k * m.key_component_stride + m.key_index_offset;
  u64* dst = prod + (u64)k * m.prod_component_stride + m.prod_offset;
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 l = (u64)blockIdx.x * 256 + threadIdx.x; l < n; l += stride) {
    u64 lo = 0, hi = 0;
    for (u32 j = 0; j < m.decomp; ++j) {
      const u64 a = (j == m.self) ? t_target_iter[(u64)j * n + l]
                                  : ntt_buf[(u64)m.slot[j] * n + l];
      const u64 b = m.keys[j][key_off + l];
      const u64 plo = a * b, phi = __umul64hi(a, b);
      lo += plo;
      hi += phi + (lo < plo);
    }
    // (hi * 2^64 + lo) mod q exactly (BarrettReduce128, util/gcc.hpp:20-28)
    const u64 r1 = full_reduce(hi, m.q, m.barrett);
    const u64 r2 = full_reduce(lo, m.q, m.barrett);
    const u64 plo = r1 * m.two64_mod_q, phi = __umul64hi(r1, m.two64_mod_q);
    const u64 c1 = m.shift ? ((plo >> m.shift) | (phi << (64 - m.shift))) : plo;
    const u64 r = csub(plo - __umul64hi(c1, m.mu) * m.q, m.q);
    dst[l] = csub(r + r2, m.q);
  }
}

// :146-175 round the last RNS component (add q_k / 2, reduce mod q_k), bring it to
// every decomposition modulus and add the correction; blockIdx.y = i.
__global__ void __launch_bounds__(256)
ks_round_kernel(u64* tbuf, const u64* t_last, u64 n, KsRound r) {
  const KsRoundMod mi = r.mod[blockIdx.y];
  u64* dst = tbuf + (u64)blockIdx.y * n;
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 l = (u64)blockIdx.x * 256 + threadIdx.x; l < n; l += stride) {
    const u64 x = t_last[l] + r.qk_half;
    const u64 v = csub(x - __umul64hi(x, r.barrett_k) * r.qk, r.qk);
    dst[l] = (mi.reduce ? full_reduce(v, mi.q, mi.barrett) : v) + mi.fix;
  }
}

// :180-196 (ct mod q_i - ct mod q_k) * q_k^-1 mod q_i, accumulated into the result;
// blockIdx.y = i.
__global__ void __launch_bounds__(256)
ks_finish_kernel(u64* result, const u64* prod, const u64* tbuf, u64 n, KsFinish f) {
  const KsFinishMod mi = f.mod[blockIdx.y];
  u64* data = result + (u64)blockIdx.y * n;
  const u64* p = prod + (u64)blockIdx.y * n;
  const u64* t = tbuf + (u64)blockIdx.y * n;
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 l = (u64)blockIdx.x * 256 + threadIdx.x; l < n; l += stride) {
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

hipError_t ks_gather_launch(u64* out, const u64* t_target, u64 n, u32 slots, const KsGather& g,
                            hipStream_t st) {
  if (slots == 0) return hipSuccess;
  hipLaunchKernelGGL(ks_gather_kernel, dim3(ks_grid(n), slots), dim3(256), 0, st, out, t_target, n,
                     g);
  return hipGetLastError();
}
hipError_t ks_mac_launch(u64* prod, const u64* t_target_iter, const u64* ntt_buf, u64 n,
                         u32 components, const KsMac& m, hipStream_t st) {
  hipLaunchKernelGGL(ks_mac_kernel, dim3(ks_grid(n), components), dim3(256), 0, st, prod,
                     t_target_iter, ntt_buf, n, m);
  return hipGetLastError();
}
hipError_t ks_round_launch(u64* tbuf, const u64* t_last, u64 n, u32 decomp, const KsRound& r,
                           hipStream_t st) {
  hipLaunchKernelGGL(ks_round_kernel, dim3(ks_grid(n), decomp), dim3(256), 0, st, tbuf, t_last, n,
                     r);
  return hipGetLastError();
}
hipError_t ks_finish_launch(u64* result, const u64* prod, const u64* tbuf, u64 n, u32 decomp,
                            const KsFinish& f, hipStream_t st) {
  hipLaunchKernelGGL(ks_finish_kernel, dim3(ks_grid(n), decomp), dim3(256), 0, st, result, prod,
                     tbuf, n, f);
  return hipGetLastError();
}

}  // namespace hexl_amd
