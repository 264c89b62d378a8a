// ubench3.hip -- ALU-only cost of the register subtrees (no memory, no LDS):
// wall-clock nanoseconds of SIMD time per wave-butterfly for the forward /
// inverse subtree of depth R under the Strict and Lazy arithmetic policies at
// 1..8 waves per SIMD.  This is the instruction-issue roofline of the NTT
// kernels.  (s_memtime ticks are printed too, but the counter does not run at
// the shader clock: only the wall-clock figure is a rate.)
// Build: hipcc --offload-arch=gfx950 -O3 -Ihexl_amd/csrc tools/ubench3.hip -o tools/ubench3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "modarith.h"
using namespace hexl_amd;

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

constexpr int TRIPS = 256;

template <int R, class A, bool FWD>
__global__ void __launch_bounds__(256) k(u64* out, const ulonglong2* tw, ModConst m, u64* cyc) {
  constexpr int E = 1 << R;
  u64 x[E];
  for (int e = 0; e < E; ++e) x[e] = (threadIdx.x * 977 + e * 131 + 7) % m.q;
  ulonglong2 w[E];  // E-1 twiddles, held in registers
  for (int e = 0; e < E; ++e) w[e] = tw[(threadIdx.x & 63) * E + e];
  u64 t0 = __builtin_readcyclecounter();
  for (int it = 0; it < TRIPS; ++it) {
#pragma unroll
    for (int v = 0; v < R; ++v) {
      const int half = 1 << (R - 1 - v);
#pragma unroll
      for (int g = 0; g < (1 << v); ++g) {
        const ulonglong2 ww = w[(1 << v) + g];
#pragma unroll
        for (int j = 0; j < half; ++j) {
          if (FWD)
            fwd_butterfly<A>(x[g * 2 * half + j], x[g * 2 * half + j + half], ww.x, ww.y, m);
          else if constexpr (A::kLazy) {  // (one butterfly of lazy_inverse.h's network, offset 8q)
            u64& a = x[g * 2 * half + j];
            u64& b = x[g * 2 * half + j + half];
            const u64 d = a + (m.two_q << 2) - b;
            a = a + b;
            b = mul_add_lazy2<false>(0, d, ww.x, ww.y, m.neg_two_q);
          } else
            inv_butterfly<A>(x[g * 2 * half + j], x[g * 2 * half + j + half], ww.x, ww.y, m);
        }
      }
    }
    // keep magnitudes bounded across trips (cheap, same for every variant)
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] &= (1ull << 56) - 1;
  }
  u64 t1 = __builtin_readcyclecounter();
  u64 acc = 0;
  for (int e = 0; e < E; ++e) acc += x[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// FP64 candidate for q < 2^50 (the reference's IFMA/FP64-class moduli):
// values are exact integers held in doubles, balanced residues, one twiddle word.
//   h = y*W (rounded), l = fma(y, W, -h) (exact low part), qe = rint(h / q),
//   T = (h - qe*q) + l  exactly = y*W - qe*q, |T| <~ 0.5 q + small.
struct DblConst {
  double q, qinv;
};
#pragma clang fp contract(off)
__device__ __forceinline__ void fwd_butterfly_dbl(double& x, double& y, double W,
                                                  const DblConst& c) {
  const double h = y * W;
  const double l = __builtin_fma(y, W, -h);
  const double qe = __builtin_rint(h * c.qinv);
  const double t = __builtin_fma(-qe, c.q, h) + l;
  y = x - t;
  x = x + t;
}
__device__ __forceinline__ void inv_butterfly_dbl(double& x, double& y, double W,
                                                  const DblConst& c) {
  const double d = x - y;
  x = x + y;
  const double h = d * W;
  const double l = __builtin_fma(d, W, -h);
  const double qe = __builtin_rint(h * c.qinv);
  y = __builtin_fma(-qe, c.q, h) + l;
}

template <int R, bool FWD>
__global__ void __launch_bounds__(256) kd(double* out, const double* tw, DblConst c, u64* cyc) {
  constexpr int E = 1 << R;
  double x[E];
  for (int e = 0; e < E; ++e) x[e] = (double)((threadIdx.x * 977 + e * 131 + 7) % 1000003);
  double w[E];
  for (int e = 0; e < E; ++e) w[e] = tw[(threadIdx.x & 63) * E + e];
  u64 t0 = __builtin_readcyclecounter();
  for (int it = 0; it < TRIPS; ++it) {
#pragma unroll
    for (int v = 0; v < R; ++v) {
      const int half = 1 << (R - 1 - v);
#pragma unroll
      for (int g = 0; g < (1 << v); ++g) {
        const double ww = w[(1 << v) + g];
#pragma unroll
        for (int j = 0; j < half; ++j) {
          if (FWD)
            fwd_butterfly_dbl(x[g * 2 * half + j], x[g * 2 * half + j + half], ww, c);
          else
            inv_butterfly_dbl(x[g * 2 * half + j], x[g * 2 * half + j + half], ww, c);
        }
      }
    }
    // full balanced reduction of every element once per subtree (what a kernel
    // would do every ~8 stages): 3 instructions per element
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = __builtin_fma(-__builtin_rint(x[e] * c.qinv), c.q, x[e]);
  }
  u64 t1 = __builtin_readcyclecounter();
  double acc = 0;
  for (int e = 0; e < E; ++e) acc += x[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int R, bool FWD>
static void run_dbl(const char* name, const double* tw, DblConst c) {
  printf("%-34s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;
    double* out;
    u64* cyc;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    CK(hipMalloc(&cyc, (size_t)blocks * 8));
    kd<R, FWD><<<blocks, 256>>>(out, tw, c, cyc);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 4; ++rep) kd<R, FWD><<<blocks, 256>>>(out, tw, c, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 4;
    const double bfly = (double)TRIPS * R * (1 << (R - 1));
    printf("  w%d: %5.1f ns/bfly/SIMD", wps, ms * 1e6 / (bfly * wps));
    CK(hipFree(out));
    CK(hipFree(cyc));
  }
  printf("\n");
}

template <int R, class A, bool FWD>
static void run(const char* name, const ulonglong2* tw, ModConst m) {
  printf("%-34s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;
    u64 *out, *cyc;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    CK(hipMalloc(&cyc, (size_t)blocks * 8));
    k<R, A, FWD><<<blocks, 256>>>(out, tw, m, cyc);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 4; ++rep) k<R, A, FWD><<<blocks, 256>>>(out, tw, m, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 4;
    std::vector<u64> h(blocks);
    CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= blocks;
    const double bfly = (double)TRIPS * R * (1 << (R - 1));
    // waves per SIMD = wps, 1024 SIMDs: SIMD-ns per wave-butterfly
    const double ns = ms * 1e6 / (bfly * wps);
    printf("  w%d: %5.1f ns/bfly/SIMD (%5.1f ticks/wave)", wps, ns, avg / bfly);
    CK(hipFree(out));
    CK(hipFree(cyc));
  }
  printf("\n");
}

int main() {
  const u64 q = 18014398510661633ull;
  const ModConst m = make_mod_const(q);
  std::vector<ulonglong2> h(64 * 16);
  for (size_t i = 0; i < h.size(); ++i) {
    u64 W = (0x9E3779B97F4A7C15ull * (i + 1)) % q;
    h[i].x = W;
    h[i].y = (u64)((((unsigned __int128)W) << 63) / q);
  }
  ulonglong2* tw;
  CK(hipMalloc(&tw, h.size() * sizeof(ulonglong2)));
  CK(hipMemcpy(tw, h.data(), h.size() * sizeof(ulonglong2), hipMemcpyHostToDevice));
  run<3, Lazy, true>("fwd subtree R=3 Lazy", tw, m);
  run<4, Lazy, true>("fwd subtree R=4 Lazy", tw, m);
  run<3, Strict, true>("fwd subtree R=3 Strict", tw, m);
  run<4, Strict, true>("fwd subtree R=4 Strict", tw, m);
  run<3, Lazy, false>("inv subtree R=3 Lazy (no ladder)", tw, m);
  run<4, Lazy, false>("inv subtree R=4 Lazy (no ladder)", tw, m);
  run<3, Strict, false>("inv subtree R=3 Strict", tw, m);
  {
    const u64 q50 = 562949954093057ull;  // BASELINE configs[1]'s 50-bit prime
    std::vector<double> hd(64 * 16);
    for (size_t i = 0; i < hd.size(); ++i) hd[i] = (double)((0x9E3779B97F4A7C15ull * (i + 1)) % q50);
    double* twd;
    CK(hipMalloc(&twd, hd.size() * sizeof(double)));
    CK(hipMemcpy(twd, hd.data(), hd.size() * sizeof(double), hipMemcpyHostToDevice));
    const DblConst c{(double)q50, 1.0 / (double)q50};
    run_dbl<3, true>("fwd subtree R=3 FP64 (q < 2^50)", twd, c);
    run_dbl<4, true>("fwd subtree R=4 FP64 (q < 2^50)", twd, c);
    run_dbl<3, false>("inv subtree R=3 FP64 (q < 2^50)", twd, c);
    run_dbl<4, false>("inv subtree R=4 FP64 (q < 2^50)", twd, c);
  }
  return 0;
}
