// ubench.hip -- gfx950 micro-benchmarks that size the NTT design:
//   * issue rate of the integer instructions a 64-bit Harvey butterfly is made
//     of (v_mad_u64_u32, v_mul_lo_u32, v_mul_hi_u32, v_mad_u32_u24, 64-bit add,
//     v_fma_f64), in wave-instructions per cycle per SIMD;
//   * streaming copy bandwidth with 8-byte and 16-byte accesses per lane (the
//     "achievable HBM" figure every roofline fraction in DESIGN.md is quoted
//     against).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                             \
    }                                                                      \
  } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;

// Each kernel runs ILP independent dependency chains of one instruction kind.
template <int KIND>
__global__ void __launch_bounds__(256) alu_kernel(uint64_t* out, uint64_t seed) {
  uint64_t a[ILP];
  uint32_t lo[ILP], hi[ILP];
  double d[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) {
    a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B97F4A7C15ULL;
    lo[i] = (uint32_t)a[i];
    hi[i] = (uint32_t)(a[i] >> 32) | 1;
    d[i] = (double)lo[i];
  }
  const uint32_t m = (uint32_t)seed | 1;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      if (KIND == 0) {  // v_mad_u64_u32
        a[i] = (uint64_t)(uint32_t)a[i] * m + a[i];
      } else if (KIND == 1) {  // v_mul_lo_u32
        lo[i] = lo[i] * hi[i];
      } else if (KIND == 2) {  // v_mul_hi_u32
        lo[i] = __umulhi(lo[i], hi[i]) + 3;
      } else if (KIND == 3) {  // v_mad_u32_u24
        lo[i] = __umul24(lo[i], hi[i]) + lo[i];
      } else if (KIND == 4) {  // 64-bit add
        a[i] = a[i] + (a[i] >> 1);
      } else if (KIND == 5) {  // v_fma_f64
        d[i] = __fma_rn(d[i], 1.0000001, 0.5);
      } else if (KIND == 6) {  // v_add_u32 (full-rate reference)
        lo[i] = lo[i] + hi[i];
        asm volatile("" : "+v"(lo[i]));
      } else if (KIND == 7) {  // full 64x64 -> hi 64
        a[i] = __umul64hi(a[i], a[i] | 1);
      } else if (KIND == 8) {  // 64-bit low product
        a[i] = a[i] * (a[i] | 1);
      }
    }
  }
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) acc += a[i] + lo[i] + (uint64_t)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename T>
__global__ void __launch_bounds__(256) copy_kernel(T* __restrict__ dst,
                                                   const T* __restrict__ src,
                                                   size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}

// Copy with a (4 MiB apart) 16-way strided tile per thread like the first NTT pass
__global__ void __launch_bounds__(256) copy_strided16(uint64_t* __restrict__ dst,
                                                      const uint64_t* __restrict__ src,
                                                      size_t npoly) {
  // poly = 65536 u64; thread handles column c of poly p: elements c + e*4096
  size_t wi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t p = wi >> 12, c = wi & 4095;
  if (p >= npoly) return;
  const uint64_t* s = src + p * 65536 + c;
  uint64_t* d = dst + p * 65536 + c;
  uint64_t x[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) x[e] = s[e * 4096];
#pragma unroll
  for (int e = 0; e < 16; ++e) d[e * 4096] = x[e] + 1;
}

template <int KIND>
static void run_alu(const char* name, double ops_per_iter) {
  const int blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
  uint64_t* out;
  CK(hipMalloc(&out, (size_t)blocks * threads * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  alu_kernel<KIND><<<blocks, threads>>>(out, 12345);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  alu_kernel<KIND><<<blocks, threads>>>(out, 12345);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  double wave_instr = (double)blocks * (threads / 64) * ITERS * ILP * ops_per_iter;
  double per_simd_per_s = wave_instr / (256.0 * 4) / (ms * 1e-3);
  printf("%-28s %8.3f ms  %7.3f G wave-instr/s/SIMD  => %6.2f cycles/wave-instr @2.4GHz\n",
         name, ms, per_simd_per_s * 1e-9, 2.4e9 / per_simd_per_s);
  CK(hipFree(out));
}

template <typename T>
static void run_copy(const char* name, size_t bytes) {
  T *a, *b;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes));
  size_t n = bytes / sizeof(T);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int grid : {2048, 8192, 65536}) {
    copy_kernel<T><<<grid, 256>>>(b, a, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) copy_kernel<T><<<grid, 256>>>(b, a, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-20s grid %6d  %8.3f ms/copy  %8.1f GB/s (read+write)\n", name, grid,
           ms / 5, 2.0 * bytes / (ms / 5 * 1e-3) * 1e-9);
  }
  CK(hipFree(a));
  CK(hipFree(b));
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs %d  clock %d kHz  L2 %d B\n", p.name, p.multiProcessorCount,
         p.clockRate, p.l2CacheSize);
  run_alu<6>("v_add_u32 (reference)", 1);
  run_alu<0>("v_mad_u64_u32", 1);
  run_alu<1>("v_mul_lo_u32", 1);
  run_alu<2>("v_mul_hi_u32 (+add)", 1);
  run_alu<3>("v_mad_u32_u24", 1);
  run_alu<4>("64-bit add (+shift)", 1);
  run_alu<5>("v_fma_f64", 1);
  run_alu<7>("__umul64hi (sequence)", 1);
  run_alu<8>("64-bit mul lo (sequence)", 1);
  const size_t GB = 1ull << 30;
  run_copy<uint64_t>("copy 8 B/lane", 2 * GB);
  run_copy<ulonglong2>("copy 16 B/lane", 2 * GB);
  {
    uint64_t *a, *b;
    size_t npoly = 4096;
    CK(hipMalloc(&a, npoly * 65536 * 8));
    CK(hipMalloc(&b, npoly * 65536 * 8));
    CK(hipMemset(a, 1, npoly * 65536 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    copy_strided16<<<npoly * 4096 / 256, 256>>>(b, a, npoly);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) copy_strided16<<<npoly * 4096 / 256, 256>>>(b, a, npoly);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("copy 16x strided 8 B/lane (NTT pass shape, 2 GiB)  %8.3f ms  %8.1f GB/s\n", ms / 5,
           2.0 * npoly * 65536 * 8 / (ms / 5 * 1e-3) * 1e-9);
  }
  return 0;
}
