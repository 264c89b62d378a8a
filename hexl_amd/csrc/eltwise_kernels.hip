// eltwise_kernels.hip -- element-wise modular arithmetic on uint64 vectors for
// gfx950.  Streaming, HBM-bound kernels: 16 bytes per lane per access, one
// pair of elements per thread with a grid-stride fallback for very large n.
//
// Replaces (results canonical in [0, q), hence bit-identical):
//   EltwiseAddMod     hexl/eltwise/eltwise-add-mod.cpp:16-113
//   EltwiseSubMod     hexl/eltwise/eltwise-sub-mod.cpp:15-110
//   EltwiseMultMod    hexl/eltwise/eltwise-mult-mod-internal.hpp:34-100
//   EltwiseFMAMod     hexl/eltwise/eltwise-fma-mod-internal.hpp:12-39
//   EltwiseReduceMod  hexl/eltwise/eltwise-reduce-mod.cpp:16-123
//   EltwiseCmpAdd     hexl/eltwise/eltwise-cmp-add.cpp:16-106
//   EltwiseCmpSubMod  hexl/eltwise/eltwise-cmp-sub-mod.cpp:18-66
//   DyadicMultiply    hexl/experimental/seal/dyadic-multiply-internal.cpp:17-74
//                     (one fused kernel: 56 B per coefficient instead of the 120 B
//                     of its five element-wise calls)
// Algorithmic bytes per element: 24 (two inputs + one output) or 16.
#include <hip/hip_runtime.h>

#include "internal.h"
#include "modarith.h"

namespace hexl_amd {

// x in [0, k*q) -> [0, q) by conditional subtraction, k in {1,2,4,8}
// (number-theory.hpp:214-258 ReduceMod<k>)
__device__ __forceinline__ u64 reduce_k(u64 x, u64 q, u64 k) {
  if (k >= 8) x = csub(x, q << 2);
  if (k >= 4) x = csub(x, q << 1);
  if (k >= 2) x = csub(x, q);
  return x;
}

struct AddOp {
  u64 q;
  __device__ __forceinline__ u64 operator()(u64 a, u64 b) const { return csub(a + b, q); }
};
struct AddScalarOp {  // eltwise-add-mod.cpp:45-69: compare against q - b
  u64 b, diff;
  __device__ __forceinline__ u64 operator()(u64 a, u64) const {
    return a >= diff ? a - diff : a + b;
  }
};
struct SubOp {
  u64 q;
  __device__ __forceinline__ u64 operator()(u64 a, u64 b) const {
    return a >= b ? a - b : a + q - b;
  }
};
struct SubScalarOp {
  u64 q, b;
  __device__ __forceinline__ u64 operator()(u64 a, u64) const {
    return a >= b ? a - b : a + q - b;
  }
};
// Generalised Barrett, alpha = 62, beta = -2 (eltwise-mult-mod-internal.hpp:50-93):
// c1 = floor(x*y / 2^(ceil(log q) - 2)); q_hat = hi64(c1 * mu); Z = lo64(x*y) - q_hat*q.
struct MultOp {
  u64 q, mu, in_mf;
  u32 shift;  // ceil(log2 q) - 2
  __device__ __forceinline__ u64 operator()(u64 a, u64 b) const {
    const u64 x = reduce_k(a, q, in_mf);
    const u64 y = reduce_k(b, q, in_mf);
    const u64 lo = x * y;
    const u64 hi = __umul64hi(x, y);
    const u64 c1 = shift ? ((lo >> shift) | (hi << (64 - shift))) : lo;
    const u64 q_hat = __umul64hi(c1, mu);
    return csub(lo - q_hat * q, q);
  }
};
// (a * s + c) mod q with s reduced on the host and its Shoup factor sp
// (eltwise-fma-mod-internal.hpp:12-39; MultiplyMod number-theory.cpp:54-59)
template <bool HAS_C>
struct FmaOp {
  u64 q, s, sp, in_mf;
  __device__ __forceinline__ u64 operator()(u64 a, u64 c) const {
    const u64 x = reduce_k(a, q, in_mf);
    u64 r = csub(mul_lazy(x, s, sp, q), q);
    if (HAS_C) r = csub(r + reduce_k(c, q, in_mf), q);
    return r;
  }
};
// eltwise-reduce-mod.cpp:16-79.  mode 0: arbitrary word -> [0,q) (Barrett);
// 1: arbitrary word -> [0,2q); 2: [0,2q)->[0,q); 3: [0,4q)->[0,q);
// 4: [0,4q)->[0,2q); 5: copy.
struct ReduceOp {
  u64 q, barrett;  // floor(2^64 / q)
  int mode;
  __device__ __forceinline__ u64 barrett1(u64 x) const {
    if (x < q) return x;
    const u64 r = x - __umul64hi(x, barrett) * q;
    return r;
  }
  __device__ __forceinline__ u64 operator()(u64 a, u64) const {
    switch (mode) {
      case 0: {
        u64 r = barrett1(a);
        return a < q ? a : csub(r, q);
      }
      case 1:
        return barrett1(a);
      case 2:
        return csub(a, q);
      case 3:
        return csub(csub(a, q << 1), q);
      case 4:
        return csub(a, q << 1);
      default:
        return a;
    }
  }
};
// Fused ReduceMod(q -> 1) on both vector inputs + FMAMod (BASELINE config 5).
template <bool HAS_C>
struct ReduceFmaOp {
  u64 q, s, sp, barrett;
  __device__ __forceinline__ u64 full(u64 x) const {
    if (x < q) return x;
    return csub(x - __umul64hi(x, barrett) * q, q);
  }
  __device__ __forceinline__ u64 operator()(u64 a, u64 c) const {
    u64 r = csub(mul_lazy(full(a), s, sp, q), q);
    if (HAS_C) r = csub(r + full(c), q);
    return r;
  }
};

// CMPINT (hexl/include/hexl/util/util.hpp:16-25) and Compare
// (hexl/util/util-internal.hpp:16-41).  `cmp` is uniform: the switch is scalar.
__device__ __forceinline__ bool compare_cmpint(int cmp, u64 lhs, u64 rhs) {
  switch (cmp) {
    case 0: return lhs == rhs;
    case 1: return lhs < rhs;
    case 2: return lhs <= rhs;
    case 3: return false;
    case 4: return lhs != rhs;
    case 5: return lhs >= rhs;
    case 6: return lhs > rhs;
    default: return true;
  }
}
// eltwise-cmp-add.cpp:32-106: plain 64-bit addition where the comparison holds
struct CmpAddOp {
  u64 bound, diff;
  int cmp;
  __device__ __forceinline__ u64 operator()(u64 a, u64) const {
    return compare_cmpint(cmp, a, bound) ? a + diff : a;
  }
};
// eltwise-cmp-sub-mod.cpp:47-66: the comparison sees the unreduced word; the
// word is then reduced with a true `% modulus` (any modulus > 1, any 64-bit
// input: single-word Barrett with floor(2^64/m) is off by at most one) and diff
// subtracted as SubUIntMod does (number-theory.cpp:68-73, same wrapping
// expression).
struct CmpSubModOp {
  u64 m, barrett, bound, diff;
  int cmp;
  __device__ __forceinline__ u64 operator()(u64 a, u64) const {
    u64 r = a - __umul64hi(a, barrett) * m;
    r = r >= m ? r - m : r;
    if (compare_cmpint(cmp, a, bound)) {
      const u64 d = (r + m) - diff;
      r = d >= m ? d - m : d;
    }
    return r;
  }
};

typedef u64 u64x2 __attribute__((ext_vector_type(2)));

template <class Op, bool HAS_B>
__global__ void __launch_bounds__(256)
eltwise_vec2(u64* res, const u64* a, const u64* b, u64 npairs, Op op) {
  // (no __restrict__: the result may alias an operand -- in place is the common use;
  // a thread loads its inputs before it stores)
  const u64x2* a2 = reinterpret_cast<const u64x2*>(a);
  const u64x2* b2 = reinterpret_cast<const u64x2*>(b);
  u64x2* r2 = reinterpret_cast<u64x2*>(res);
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < npairs; i += stride) {
    // nontemporal 16-byte accesses: the vectors are streamed once
    const u64x2 va = __builtin_nontemporal_load(a2 + i);
    u64x2 vb = {0, 0};
    if (HAS_B) vb = __builtin_nontemporal_load(b2 + i);
    u64x2 vr;
    vr.x = op(va.x, vb.x);
    vr.y = op(va.y, vb.y);
    __builtin_nontemporal_store(vr, r2 + i);
  }
}

template <class Op, bool HAS_B>
__global__ void __launch_bounds__(256)
eltwise_scalar(u64* res, const u64* a, const u64* b, u64 begin, u64 n, Op op) {
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 i = begin + (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const u64 vb = HAS_B ? b[i] : 0;
    res[i] = op(a[i], vb);
  }
}

static unsigned grid_for(u64 items) {
  const u64 blocks = (items + 255) / 256;
  const u64 cap = 1u << 20;
  return (unsigned)(blocks < cap ? (blocks ? blocks : 1) : cap);
}

template <class Op, bool HAS_B>
static hipError_t run(const EltArgs& g, const u64* b, Op op, hipStream_t st) {
  if (g.n == 0) return hipSuccess;
  const uintptr_t mask = (uintptr_t)g.result | (uintptr_t)g.a | (HAS_B ? (uintptr_t)b : 0);
  u64 done = 0;
  ScopedKernelTimer timer("eltwise", st);
  if ((mask & 15) == 0 && g.n >= 2) {
    const u64 npairs = g.n / 2;
    hipLaunchKernelGGL((eltwise_vec2<Op, HAS_B>), dim3(grid_for(npairs)), dim3(256), 0, st,
                       g.result, g.a, b, npairs, op);
    done = npairs * 2;
  }
  if (done < g.n) {
    hipLaunchKernelGGL((eltwise_scalar<Op, HAS_B>), dim3(grid_for(g.n - done)), dim3(256), 0,
                       st, g.result, g.a, b, done, g.n, op);
  }
  return hipGetLastError();
}

static u64 host_floor_2_64_over(u64 num_shifted_operand, u64 q) {
  // floor(operand * 2^64 / q)
  return (u64)((((unsigned __int128)num_shifted_operand) << 64) / q);
}

static u64 host_reduce_k(u64 x, u64 q, u64 k) {
  if (k >= 8 && x >= 4 * q) x -= 4 * q;
  if (k >= 4 && x >= 2 * q) x -= 2 * q;
  if (k >= 2 && x >= q) x -= q;
  return x;
}

static MultOp make_mult_op(u64 q, u64 in_mf);

hipError_t eltwise_launch(EltOp op, const EltArgs& g, hipStream_t st) {
  const u64 q = g.q;
  switch (op) {
    case ELT_ADD:
      return run<AddOp, true>(g, g.b, AddOp{q}, st);
    case ELT_ADD_SCALAR:
      return run<AddScalarOp, false>(g, nullptr, AddScalarOp{g.scalar, q - g.scalar}, st);
    case ELT_SUB:
      return run<SubOp, true>(g, g.b, SubOp{q}, st);
    case ELT_SUB_SCALAR:
      return run<SubScalarOp, false>(g, nullptr, SubScalarOp{q, g.scalar}, st);
    case ELT_MULT:
      return run<MultOp, true>(g, g.b, make_mult_op(q, g.in_mf), st);
    case ELT_FMA: {
      const u64 s = host_reduce_k(g.scalar, q, g.in_mf);
      const u64 sp = host_floor_2_64_over(s, q);
      if (g.b) return run<FmaOp<true>, true>(g, g.b, FmaOp<true>{q, s, sp, g.in_mf}, st);
      return run<FmaOp<false>, false>(g, nullptr, FmaOp<false>{q, s, sp, g.in_mf}, st);
    }
    case ELT_REDUCE: {
      int mode;
      if (g.in_mf == g.out_mf)
        mode = 5;  // copy (eltwise-reduce-mod.cpp:94-99); in place: no-op
      else if (g.in_mf == q)
        mode = g.out_mf == 1 ? 0 : 1;
      else if (g.in_mf == 2)
        mode = 2;
      else
        mode = g.out_mf == 1 ? 3 : 4;
      if (mode == 5 && g.result == g.a) return hipSuccess;
      return run<ReduceOp, false>(g, nullptr, ReduceOp{q, host_floor_2_64_over(1, q), mode},
                                  st);
    }
    case ELT_REDUCE_FMA: {
      const u64 s = g.scalar % q;
      const u64 sp = host_floor_2_64_over(s, q);
      const u64 bar = host_floor_2_64_over(1, q);
      if (g.b)
        return run<ReduceFmaOp<true>, true>(g, g.b, ReduceFmaOp<true>{q, s, sp, bar}, st);
      return run<ReduceFmaOp<false>, false>(g, nullptr, ReduceFmaOp<false>{q, s, sp, bar}, st);
    }
    case ELT_CMP_ADD:
      return run<CmpAddOp, false>(g, nullptr, CmpAddOp{g.bound, g.scalar, g.cmp}, st);
    case ELT_CMP_SUB_MOD:
      return run<CmpSubModOp, false>(
          g, nullptr, CmpSubModOp{q, host_floor_2_64_over(1, q), g.bound, g.scalar, g.cmp}, st);
  }
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// DyadicMultiply: (x0, x1) * (y0, y1) -> (x0 y0, x0 y1 + x1 y0, x1 y1) per modulus
// ---------------------------------------------------------------------------
constexpr int kDyadicModuliPerLaunch = 32;
struct DyadicModuli {
  MultOp mult[kDyadicModuliPerLaunch];
};

// blockIdx.y = modulus within this launch; every thread reads its four inputs
// before it writes its three outputs, so the result may alias either operand
// (the in-place forms of the reference's tests).
// VEC2: two adjacent coefficients per thread, 16 bytes per lane per access like the other
// streaming kernels (seven streams of 8-byte accesses ran at 0.63 of the HBM peak where
// MultMod's 16-byte ones run at 0.78); needs 16-byte aligned buffers and an even n (every
// polynomial then starts on a 16-byte boundary).
template <bool VEC2>
__global__ void __launch_bounds__(256)
dyadic_multiply_kernel(u64* res, const u64* x, const u64* y, u64 n, u64 n_proc, u64 poly_size,
                       u64 first_modulus, DyadicModuli mods) {
  // blockIdx.z = ciphertext pair of a batched call: operands are 2, results 3 polynomials
  x += (u64)blockIdx.z * 2 * poly_size;
  y += (u64)blockIdx.z * 2 * poly_size;
  res += (u64)blockIdx.z * 3 * poly_size;
  const MultOp op = mods.mult[blockIdx.y];
  const u64 base = (first_modulus + blockIdx.y) * n;
  const u64 stride = (u64)gridDim.x * 256;
  if constexpr (VEC2) {
    const u64 pairs = n_proc >> 1;  // (n_proc is even with n)
    for (u64 e = (u64)blockIdx.x * 256 + threadIdx.x; e < pairs; e += stride) {
      const u64 p0 = base + 2 * e, p1 = p0 + poly_size, p2 = p1 + poly_size;
      const ulonglong2 x0 = *reinterpret_cast<const ulonglong2*>(x + p0);
      const ulonglong2 x1 = *reinterpret_cast<const ulonglong2*>(x + p1);
      const ulonglong2 y0 = *reinterpret_cast<const ulonglong2*>(y + p0);
      const ulonglong2 y1 = *reinterpret_cast<const ulonglong2*>(y + p1);
      ulonglong2 r0, r1, r2;
      r2.x = op(x1.x, y1.x);
      r2.y = op(x1.y, y1.y);
      r1.x = csub(op(x0.x, y1.x) + op(x1.x, y0.x), op.q);
      r1.y = csub(op(x0.y, y1.y) + op(x1.y, y0.y), op.q);
      r0.x = op(x0.x, y0.x);
      r0.y = op(x0.y, y0.y);
      *reinterpret_cast<ulonglong2*>(res + p2) = r2;
      *reinterpret_cast<ulonglong2*>(res + p1) = r1;
      *reinterpret_cast<ulonglong2*>(res + p0) = r0;
    }
  } else {
    for (u64 e = (u64)blockIdx.x * 256 + threadIdx.x; e < n_proc; e += stride) {
      const u64 p0 = base + e, p1 = p0 + poly_size, p2 = p1 + poly_size;
      const u64 x0 = x[p0], x1 = x[p1], y0 = y[p0], y1 = y[p1];
      const u64 r2 = op(x1, y1);
      const u64 t = op(x1, y0);
      const u64 r1 = csub(op(x0, y1) + t, op.q);
      const u64 r0 = op(x0, y0);
      res[p2] = r2;
      res[p1] = r1;
      res[p0] = r0;
    }
  }
}

static MultOp make_mult_op(u64 q, u64 in_mf) {
  const u32 ceil_log = 64 - __builtin_clzll(q);  // floor(log2 q) + 1
  const u32 shift = ceil_log - 2;
  const u64 mu = host_floor_2_64_over(1ull << (ceil_log + 62 - 64), q);
  return MultOp{q, mu, in_mf, shift};
}

hipError_t dyadic_multiply_launch(u64* result, const u64* op1, const u64* op2, u64 n,
                                  const u64* moduli, u64 num_moduli, u64 pairs,
                                  hipStream_t st) {
  // dyadic-multiply-internal.cpp:33-34: whole tiles of min(n, 512) coefficients
  const u64 tile = n < 512 ? n : 512;
  const u64 n_proc = (n / tile) * tile;
  const u64 poly_size = n * num_moduli;
  ScopedKernelTimer timer("dyadic_multiply", st);
  for (u64 first = 0; first < num_moduli; first += kDyadicModuliPerLaunch) {
    const u64 count =
        num_moduli - first < kDyadicModuliPerLaunch ? num_moduli - first : kDyadicModuliPerLaunch;
    DyadicModuli mods;
    for (u64 i = 0; i < count; ++i) mods.mult[i] = make_mult_op(moduli[first + i], 1);
    for (u64 i = count; i < kDyadicModuliPerLaunch; ++i) mods.mult[i] = mods.mult[0];
    const bool vec2 = (n & 1) == 0 && (n_proc & 1) == 0 &&
                      (((uintptr_t)result | (uintptr_t)op1 | (uintptr_t)op2) & 15) == 0;
    unsigned gx = grid_for(vec2 ? n_proc >> 1 : n_proc);
    if (gx > 65535u * 16) gx = 65535u * 16;
    if (vec2)
      hipLaunchKernelGGL(dyadic_multiply_kernel<true>, dim3(gx, (unsigned)count, (unsigned)pairs),
                         dim3(256), 0, st, result, op1, op2, n, n_proc, poly_size, first, mods);
    else
      hipLaunchKernelGGL(dyadic_multiply_kernel<false>, dim3(gx, (unsigned)count, (unsigned)pairs),
                         dim3(256), 0, st, result, op1, op2, n, n_proc, poly_size, first, mods);
  }
  return hipGetLastError();
}

// splitmix64 stream: coefficient i of polynomial b = mix(seed0 + b + (i+1)*gamma) mod bound
__global__ void __launch_bounds__(256)
fill_splitmix_kernel(u64* data, u64 n, u64 total, u64 seed0, u64 bound) {
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < total; k += stride) {
    const u64 b = k / n, i = k - b * n;
    u64 z = seed0 + b + (i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    data[k] = bound ? z % bound : z;
  }
}

// Debug contract (HEXL_CHECK_BOUNDS of the reference's debug builds, e.g.
// hexl/ntt/ntt-internal.cpp:198, :261, hexl/eltwise/eltwise-mult-mod.cpp:31-33): the number
// of words >= bound, added to *violations (one atomic per wave that found any).
__global__ void __launch_bounds__(256)
count_out_of_bounds_kernel(const u64* data, u64 n, u64 bound, unsigned long long* violations) {
  const u64 stride = (u64)gridDim.x * 256;
  unsigned long long bad = 0;
  for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) bad += data[k] >= bound;
  for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(violations, bad);
}

hipError_t count_out_of_bounds_launch(const u64* data, u64 n, u64 bound,
                                      unsigned long long* violations, hipStream_t st) {
  unsigned grid = grid_for(n);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(count_out_of_bounds_kernel, dim3(grid), dim3(256), 0, st, data, n, bound,
                     violations);
  return hipGetLastError();
}

// Behind everything enqueued on its stream so far: publishes `seq` in device-mapped host memory, where the
// calling host thread polls for it (capi.cpp: Staging::finish).
__global__ void completion_flag_kernel(u32* flag, u32 seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t completion_flag_launch(u32* flag, u32 seq, hipStream_t st) {
  hipLaunchKernelGGL(completion_flag_kernel, dim3(1), dim3(1), 0, st, flag, seq);
  return hipGetLastError();
}

hipError_t fill_splitmix_launch(u64* data, u64 n, u64 batch, u64 seed0, u64 bound,
                                hipStream_t st) {
  const u64 total = n * batch;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(fill_splitmix_kernel, dim3(grid_for(total)), dim3(256), 0, st, data, n,
                     total, seed0, bound);
  return hipGetLastError();
}

}  // namespace hexl_amd
