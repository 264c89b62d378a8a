// ubench2.hip -- per-instruction issue cost on gfx950 in shader cycles
// (s_memtime), for the integer instructions the NTT butterflies are built from.
// Each measurement: 8 independent dependency chains x 32 instructions per loop
// trip, emitted with inline asm so the compiler cannot rewrite them; 1, 2, 4
// and 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o tools/ubench2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

constexpr int TRIPS = 256;

#define REP4(X) X X X X
#define REP8(X) REP4(X) REP4(X)

// one "round" = the instruction applied to each of 8 chains
#define ROUND32(a0, a1, a2, a3, a4, a5, a6, a7, T) \
  T(a0) T(a1) T(a2) T(a3) T(a4) T(a5) T(a6) T(a7)

template <int KIND>
__global__ void __launch_bounds__(256) k(uint64_t* out, uint32_t seed, uint64_t* cyc) {
  uint32_t a[8], b = seed | 1, c = seed * 3 + 1;
  uint64_t w[8];
  double d[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = seed * (threadIdx.x + 1) + i;
    w[i] = ((uint64_t)a[i] << 32) | (a[i] * 77u);
    d[i] = 1.0 + a[i];
  }
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < TRIPS; ++it) {
#define T_ADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define T_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
#define T_MADU64(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c) : "vcc");
#define T_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define T_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define T_MAD24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define T_MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
#define T_MULHI24(x) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
#define T_LSHLADD64(x) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(w[7]));
#define T_ADDCO(x) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define T_CMP64(x) asm volatile("v_cmp_le_u64 vcc, %1, %0" : "+v"(x) : "v"(w[7]) : "vcc");
#define T_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define T_ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define T_FMA64(x) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(d[7]));
#define T_SUBREV64(x) asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define T_MAD_I64(x) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c) : "vcc");
#define T_MINU(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define T_LSHR64(x) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(x));
#define T_PKADD(x) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(b));
    if (KIND == 0) { REP4(ROUND32(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], T_ADD)) }
    if (KIND == 1) { REP4(ROUND32(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], T_MOV)) }
    if (KIND == 2) { REP4(ROUND32(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], T_MADU64)) }
    if (KIND == 3) { REP4(ROUND32(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], T_MULLO)) }
    if (KIND == 4) { REP4(ROUND32(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], T_MULHI)) }
    if (KIND == 5) { REP4(ROUND32(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], T_MAD24)) }
    if (KIND == 6) { REP4(ROUND32(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], T_MUL24)) }
    if (KIND == 7) { REP4(ROUND32(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], T_MULHI24)) }
    if (KIND == 8) { REP4(ROUND32(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[6], T_LSHLADD64)) }
    if (KIND == 9) { REP4(ROUND32(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], T_ADDCO)) }
    if (KIND == 10) { REP4(ROUND32(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[6], T_CMP64)) }
    if (KIND == 11) { REP4(ROUND32(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], T_CNDMASK)) }
    if (KIND == 12) { REP4(ROUND32(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], T_ADD3)) }
    if (KIND == 13) { REP4(ROUND32(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[6], T_FMA64)) }
    if (KIND == 14) { REP4(ROUND32(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], T_MAD_I64)) }
    if (KIND == 15) { REP4(ROUND32(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], T_MINU)) }
    if (KIND == 16) { REP4(ROUND32(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], T_LSHR64)) }
    if (KIND == 17) { REP4(ROUND32(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], T_PKADD)) }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  uint64_t acc = 0;
  for (int i = 0; i < 8; ++i) acc += a[i] + w[i] + (uint64_t)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int instr_per_t) {
  printf("%-26s", name);
  for (int wps : {1, 2, 4, 8}) {           // waves per SIMD
    const int blocks = 256 * wps;           // 256-thread blocks = 4 waves = 1 per SIMD
    uint64_t *out, *cyc;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    CK(hipMalloc(&cyc, (size_t)blocks * 8));
    k<KIND><<<blocks, 256>>>(out, 12345, cyc);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    k<KIND><<<blocks, 256>>>(out, 12345, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(blocks);
    CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= blocks;
    const double instr = (double)TRIPS * 32 * instr_per_t;  // per wave
    // wave-level cycles per instruction, and SIMD-level (x waves sharing the SIMD)
    printf("  w%d: %5.2f cyc/instr/wave (%4.2f per SIMD slot, %5.3f ms)", wps, avg / instr,
           avg / instr / wps, ms);
    CK(hipFree(out));
    CK(hipFree(cyc));
  }
  printf("\n");
}

int main() {
  run<0>("v_add_u32", 1);
  run<1>("v_mov_b32", 1);
  run<15>("v_min_u32", 1);
  run<11>("v_cndmask_b32", 1);
  run<12>("v_add3_u32", 1);
  run<17>("v_pk_add_u16", 1);
  run<5>("v_mad_u32_u24", 1);
  run<6>("v_mul_u32_u24", 1);
  run<7>("v_mul_hi_u32_u24", 1);
  run<3>("v_mul_lo_u32", 1);
  run<4>("v_mul_hi_u32", 1);
  run<2>("v_mad_u64_u32", 1);
  run<14>("v_mad_i64_i32", 1);
  run<8>("v_lshl_add_u64", 1);
  run<16>("v_lshrrev_b64", 1);
  run<9>("v_add_co+v_addc_co", 2);
  run<10>("v_cmp_le_u64", 1);
  run<13>("v_fma_f64", 1);
  return 0;
}
