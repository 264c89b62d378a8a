// ntt_kernels.hip -- negacyclic NTT for gfx950 (MI355X), hand-written HIP.
//
// Replaces the reference's forward / inverse transform kernels
// (hexl/ntt/ntt-radix-2.cpp:17-261 and :330-519; hexl/ntt/fwd-ntt-avx512.cpp,
// hexl/ntt/inv-ntt-avx512.cpp) for batches of independent polynomials.
//
// Structure of the transform.  The forward Cooley-Tukey network is a binary
// heap of butterflies: heap node n = m + i (m = 2^s groups at stage s, group i)
// multiplies by W = R[n] and its children are 2n, 2n+1 -- exactly the
// reference's root_of_unity_powers[m + i] indexing (ntt-radix-2.cpp:128-129).
// The device tables are therefore heap-ordered arrays of (W, W') pairs, one
// 16-byte load per twiddle; the inverse table holds R[n]^-1 at the same heap
// index (the reference's stage-ordered inverse layout, ntt-internal.cpp:143-154,
// is kept on the host for the getters only).
//
// Two kernels, both built from a register-resident "subtree": a thread owns
// E = 2^r elements and runs r stages on them with no communication.
//   * strided_pass  -- the top stages of a large transform (butterfly gap
//     >= 4096): each thread owns one column, elements N/2^(a0+r) apart; lanes
//     map to consecutive columns, so every load/store is a fully coalesced
//     512-byte wave access and all twiddles are wave-uniform (scalar loads).
//   * block_pass    -- the bottom <= 12 stages on a contiguous 4096-element
//     (32 KiB) tile staged through LDS: 2-3 subtree rounds separated by LDS
//     transposes, padded (one 8-byte slot per 16) so that every ds_read_b64 /
//     ds_write_b64 pattern used is bank-conflict free or at worst 2-way on one
//     slot; the final [0,4q)->[0,q) reduction (or N^-1 scaling for the
//     inverse) is fused into the last store.
// N <= 4096 is a single block_pass launch; N = 2^16 is strided_pass(4 stages)
// + block_pass(12 stages), i.e. two HBM round trips per transform.
//
// Values stay lazy: [0,4q) forward, [0,2q) inverse, as in the reference's
// Harvey butterflies (hexl/ntt/ntt-default.hpp:28-42, :112-125).  Canonical
// outputs (output_mod_factor == 1) are therefore bit-identical to the
// reference; lazy outputs are congruent and in range.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "internal.h"
#include "modarith.h"

namespace hexl_amd {

thread_local ProfileSink* g_profile = nullptr;

// ---------------------------------------------------------------------------
// Register subtrees
// ---------------------------------------------------------------------------

// r forward stages on x[0 .. 2^r): stage v pairs elements 2^(r-1-v) apart and
// uses heap nodes (node << v) + g, g = 0 .. 2^v - 1.
template <int R, class A>
__device__ __forceinline__ void fwd_subtree(u64* x, const ulonglong2* __restrict__ tw,
                                            u32 node, const ModConst& m) {
#pragma unroll
  for (int v = 0; v < R; ++v) {
    const int half = 1 << (R - 1 - v);
#pragma unroll
    for (int g = 0; g < (1 << v); ++g) {
      const ulonglong2 w = tw[(node << v) + g];
#pragma unroll
      for (int j = 0; j < half; ++j)
        fwd_butterfly<A>(x[g * 2 * half + j], x[g * 2 * half + j + half], w.x, w.y, m);
    }
  }
}

// number of leading zero bits of e seen as an R-bit number
constexpr int leading_zeros(int e, int R) {
  int n = 0;
  for (int b = R - 1; b >= 0 && !((e >> b) & 1); --b) ++n;
  return n;
}

template <int R, int E0, class A>
struct InvLadder {
  static __device__ __forceinline__ void run(u64* x, const ModConst& m) {
    x[E0] = inv_ladder<leading_zeros(E0, R)>(x[E0], m);
    InvLadder<R, E0 + 1, A>::run(x, m);
  }
};
template <int R, class A>
struct InvLadder<R, (1 << R), A> {
  static __device__ __forceinline__ void run(u64*, const ModConst&) {}
};

// r inverse stages (deepest level first).  With LAST the v == 0 stage is the
// root of the whole transform and folds N^-1 in (ntt-radix-2.cpp:490-509).
// Lazy policy: no conditional subtraction inside the subtree, [0,2q) restored
// at exit (not needed after LAST, whose outputs are both lazy products).
template <int R, class A, bool LAST>
__device__ __forceinline__ void inv_subtree(u64* x, const ulonglong2* __restrict__ tw,
                                            u32 node, const ModConst& m, const InvLast& il) {
#pragma unroll
  for (int v = R - 1; v >= 0; --v) {
    const int half = 1 << (R - 1 - v);
    const int k = R - 1 - v;
#pragma unroll
    for (int g = 0; g < (1 << v); ++g) {
      if (LAST && v == 0) {
#pragma unroll
        for (int j = 0; j < half; ++j)
          inv_butterfly_last<A>(x[j], x[j + half], il.n1, il.n1p, il.n1w, il.n1wp, m, k);
      } else {
        const ulonglong2 w = tw[(node << v) + g];
#pragma unroll
        for (int j = 0; j < half; ++j)
          inv_butterfly<A>(x[g * 2 * half + j], x[g * 2 * half + j + half], w.x, w.y, m, k);
      }
    }
  }
  if (A::kLazy && !LAST) InvLadder<R, 0, A>::run(x, m);
}

// ---------------------------------------------------------------------------
// strided_pass: R stages whose subtree roots sit at heap level a0
// ---------------------------------------------------------------------------
// Work item -> (poly b, subtree h in [0, 2^a0), column c in [0, S)),
// S = N >> (a0 + R); element e of the item is at b*N + (h*2^R + e)*S + c.
template <bool FWD, int R, class A>
__global__ void __launch_bounds__(256)
strided_pass(u64* __restrict__ out, const u64* __restrict__ in,
             const ulonglong2* __restrict__ tw, ModConst m, u32 log_n, u32 a0, u32 finish,
             u64 items, InvLast il) {
  constexpr int E = 1 << R;
  const u64 wi = (u64)blockIdx.x * 256 + threadIdx.x;
  if (wi >= items) return;
  const u32 log_s = log_n - a0 - R;
  const u64 b = wi >> (log_n - R);
  const u32 rem = (u32)(wi & ((1ull << (log_n - R)) - 1));
  const u32 h = rem >> log_s;
  const u32 c = rem & ((1u << log_s) - 1);
  const u64 base = (b << log_n) + ((u64)h << (log_s + R)) + c;
  u32 node = (1u << a0) + h;
  // all lanes of a wave share h when a wave spans <= S columns
  if (log_s >= 6) node = __builtin_amdgcn_readfirstlane(node);

  u64 x[E];
#pragma unroll
  for (int e = 0; e < E; ++e) x[e] = in[base + ((u64)e << log_s)];

  // finish: 0 = more passes follow, 1 = end of the network (lazy output range),
  // 2 = end of the network, canonical output in [0,q)
  if (FWD) {
    fwd_subtree<R, A>(x, tw, node, m);
    if (finish) {
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = fwd_finish<A>(x[e], m, finish == 2);
    }
  } else {
    if (a0 == 0) {
      inv_subtree<R, A, true>(x, tw, node, m, il);
    } else {
      inv_subtree<R, A, false>(x, tw, node, m, il);
    }
    if (finish == 2) {
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = csub(x[e], m.q);
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) out[base + ((u64)e << log_s)] = x[e];
}

// ---------------------------------------------------------------------------
// block_pass: the bottom TB stages on contiguous 2^TB-element blocks
// ---------------------------------------------------------------------------
// A workgroup owns a 4096-element (32 KiB) tile = 4096 >> TB blocks.  RE = log2
// of the elements a thread holds: RE = 4 -> 256 threads, rounds of 4 stages;
// RE = 3 -> 512 threads, rounds of 3 stages, <= 64 VGPRs so that 4 workgroups
// = 32 waves fit a CU (the integer pipes of gfx950 need ~8 waves per SIMD to
// reach their issue rate -- tools/ubench2.hip).
constexpr int kTileLog = 12;
constexpr int kLdsWords = (1 << kTileLog) + (1 << (kTileLog - 4));

// one 8-byte pad slot per 16 elements
__device__ __forceinline__ u32 lds_slot(u32 p) { return p + (p >> 4); }

template <int TB, int RE>
struct Rounds {
  static constexpr int NR = (TB + RE - 1) / RE;
  static constexpr int R0 = TB - (NR - 1) * RE;  // stages of round 0 (1..RE)
  static constexpr int r(int j) { return j == 0 ? R0 : RE; }
  static constexpr int u(int j) { return j == 0 ? 0 : R0 + (j - 1) * RE; }
  static constexpr int w(int j) { return TB - u(j) - r(j); }
  static constexpr int kThreadsLog = kTileLog - RE;
  static constexpr int kThreads = 1 << kThreadsLog;
  static constexpr int kE = 1 << RE;
};

// Tile-local index of element e of virtual thread vt in a round with r stages
// whose finest butterfly gap is 2^w.
template <int r, int w>
__device__ __forceinline__ u32 tile_index(u32 vt, int e) {
  return ((vt >> w) << (w + r)) + ((u32)e << w) + (vt & ((1u << w) - 1));
}

// vt >> w for vt = s*threads + tid, written so that it is visibly uniform
// when 2^w >= threads.
template <int w, int TL>
__device__ __forceinline__ u32 vt_high(int s, u32 tid) {
  if (w >= TL) return (u32)s >> (w - TL);
  return (((u32)s << TL) + tid) >> w;
}

template <int TB, int RE, int j, class A, bool FWD, bool LAST>
__device__ __forceinline__ void run_round(u64* x, const ulonglong2* __restrict__ tw,
                                          u32 tid, u32 a0, u32 tile_blk0,
                                          const ModConst& m, const InvLast& il) {
  using RD = Rounds<TB, RE>;
  constexpr int r = RD::r(j);
  constexpr int w = RD::w(j);
  constexpr int u = RD::u(j);
  constexpr int SS = RD::kE >> r;
  const u32 level = 1u << (a0 + u);
#pragma unroll
  for (int s = 0; s < SS; ++s) {
    u32 node = level + (((tile_blk0 << u) + vt_high<w, RD::kThreadsLog>(s, tid)) & (level - 1));
    if (w >= 6) node = __builtin_amdgcn_readfirstlane(node);
    if (FWD)
      fwd_subtree<r, A>(x + (s << r), tw, node, m);
    else
      inv_subtree<r, A, LAST>(x + (s << r), tw, node, m, il);
  }
}

template <int TB, int RE, int j>
__device__ __forceinline__ void lds_load_round(u64* x, const u64* lds, u32 tid) {
  using RD = Rounds<TB, RE>;
  constexpr int r = RD::r(j);
  constexpr int w = RD::w(j);
  constexpr int SS = RD::kE >> r;
#pragma unroll
  for (int s = 0; s < SS; ++s)
#pragma unroll
    for (int e = 0; e < (1 << r); ++e)
      x[(s << r) + e] = lds[lds_slot(tile_index<r, w>(s * RD::kThreads + tid, e))];
}

template <int TB, int RE, int j>
__device__ __forceinline__ void lds_store_round(const u64* x, u64* lds, u32 tid) {
  using RD = Rounds<TB, RE>;
  constexpr int r = RD::r(j);
  constexpr int w = RD::w(j);
  constexpr int SS = RD::kE >> r;
#pragma unroll
  for (int s = 0; s < SS; ++s)
#pragma unroll
    for (int e = 0; e < (1 << r); ++e)
      lds[lds_slot(tile_index<r, w>(s * RD::kThreads + tid, e))] = x[(s << r) + e];
}

// forward rounds J .. NR-1: LDS -> registers -> subtree -> LDS
template <int TB, int RE, int J, class A>
__device__ __forceinline__ void fwd_mid_rounds(u64* x, u64* lds, const ulonglong2* tw, u32 tid,
                                               u32 a0, u32 tile_blk0, const ModConst& m,
                                               const InvLast& il) {
  if constexpr (J < Rounds<TB, RE>::NR) {
    lds_load_round<TB, RE, J>(x, lds, tid);
    run_round<TB, RE, J, A, true, false>(x, tw, tid, a0, tile_blk0, m, il);
    __syncthreads();
    lds_store_round<TB, RE, J>(x, lds, tid);
    __syncthreads();
    fwd_mid_rounds<TB, RE, J + 1, A>(x, lds, tw, tid, a0, tile_blk0, m, il);
  }
}

// inverse rounds J .. 1 (deepest first)
template <int TB, int RE, int J, class A>
__device__ __forceinline__ void inv_mid_rounds(u64* x, u64* lds, const ulonglong2* tw, u32 tid,
                                               u32 a0, u32 tile_blk0, const ModConst& m,
                                               const InvLast& il) {
  if constexpr (J >= 1) {
    lds_load_round<TB, RE, J>(x, lds, tid);
    run_round<TB, RE, J, A, false, false>(x, tw, tid, a0, tile_blk0, m, il);
    __syncthreads();
    lds_store_round<TB, RE, J>(x, lds, tid);
    __syncthreads();
    inv_mid_rounds<TB, RE, J - 1, A>(x, lds, tw, tid, a0, tile_blk0, m, il);
  }
}

// FWD:  global --(round 0)--> LDS --(rounds 1..)--> LDS --> coalesced store
// INV:  coalesced load --> LDS --(rounds NR-1..1)--> LDS --(round 0)--> global
template <bool FWD, int TB, int RE, class A>
__global__ void __launch_bounds__(1 << (kTileLog - RE), (RE == 3 ? 8 : 4))
block_pass(u64* __restrict__ out, const u64* __restrict__ in,
           const ulonglong2* __restrict__ tw, ModConst m, u32 log_n, u32 finish, u64 total,
           u32 vec16, InvLast il) {
  using RD = Rounds<TB, RE>;
  constexpr int NR = RD::NR;
  constexpr int kThreads = RD::kThreads;
  constexpr int kE = RD::kE;
  __shared__ u64 lds[kLdsWords];
  const u32 tid = threadIdx.x;
  const u64 tile_base = (u64)blockIdx.x << kTileLog;
  const u32 a0 = log_n - TB;  // heap level of the block roots
  // index (within its polynomial) of the first 2^TB-block of this tile
  const u32 tile_blk0 = (u32)((tile_base & ((1ull << log_n) - 1)) >> TB);
  u64 x[kE];

  if (FWD) {
    {  // round 0 straight from global memory
      constexpr int r = RD::r(0), w = RD::w(0), SS = kE >> r;
#pragma unroll
      for (int s = 0; s < SS; ++s)
#pragma unroll
        for (int e = 0; e < (1 << r); ++e) {
          const u64 k = tile_base + tile_index<r, w>(s * kThreads + tid, e);
          x[(s << r) + e] = (k < total) ? in[k] : 0;
        }
      run_round<TB, RE, 0, A, true, false>(x, tw, tid, a0, tile_blk0, m, il);
      lds_store_round<TB, RE, 0>(x, lds, tid);
      __syncthreads();
    }
    fwd_mid_rounds<TB, RE, 1, A>(x, lds, tw, tid, a0, tile_blk0, m, il);
    // coalesced copy-out, 16 bytes per lane, final reduction fused
#pragma unroll
    for (int i = 0; i < kE / 2; ++i) {
      const u32 p = 2 * (i * kThreads + tid);
      u64 v0 = lds[lds_slot(p)];
      u64 v1 = lds[lds_slot(p + 1)];
      if (finish) {
        v0 = fwd_finish<A>(v0, m, finish == 2);
        v1 = fwd_finish<A>(v1, m, finish == 2);
      }
      if (tile_base + p < total) {
        if (vec16) {
          *reinterpret_cast<ulonglong2*>(out + tile_base + p) = make_ulonglong2(v0, v1);
        } else {  // caller's buffer is only 8-byte aligned
          out[tile_base + p] = v0;
          out[tile_base + p + 1] = v1;
        }
      }
    }
  } else {
    // coalesced copy-in
#pragma unroll
    for (int i = 0; i < kE / 2; ++i) {
      const u32 p = 2 * (i * kThreads + tid);
      ulonglong2 v = make_ulonglong2(0, 0);
      if (tile_base + p < total) {
        if (vec16) {
          v = *reinterpret_cast<const ulonglong2*>(in + tile_base + p);
        } else {
          v.x = in[tile_base + p];
          v.y = in[tile_base + p + 1];
        }
      }
      lds[lds_slot(p)] = v.x;
      lds[lds_slot(p + 1)] = v.y;
    }
    __syncthreads();
    inv_mid_rounds<TB, RE, NR - 1, A>(x, lds, tw, tid, a0, tile_blk0, m, il);
    lds_load_round<TB, RE, 0>(x, lds, tid);
    if (a0 == 0)
      run_round<TB, RE, 0, A, false, true>(x, tw, tid, a0, tile_blk0, m, il);
    else
      run_round<TB, RE, 0, A, false, false>(x, tw, tid, a0, tile_blk0, m, il);
    {
      constexpr int r = RD::r(0), w = RD::w(0), SS = kE >> r;
#pragma unroll
      for (int s = 0; s < SS; ++s)
#pragma unroll
        for (int e = 0; e < (1 << r); ++e) {
          const u64 k = tile_base + tile_index<r, w>(s * kThreads + tid, e);
          u64 v = x[(s << r) + e];
          if (finish == 2) v = csub(v, m.q);
          if (k < total) out[k] = v;
        }
    }
  }
}

// ---------------------------------------------------------------------------
// Host-side planning and launch
// ---------------------------------------------------------------------------

template <bool FWD, class A>
static hipError_t launch_strided(int R, u64* out, const u64* in, const ulonglong2* tw,
                                 const ModConst& m, u32 log_n, u32 a0, u32 finish, u64 batch,
                                 const InvLast& il, hipStream_t st) {
  const u64 items = batch << (log_n - R);
  const unsigned grid = (unsigned)((items + 255) / 256);
  ScopedKernelTimer timer(FWD ? "ntt_fwd_strided_pass" : "ntt_inv_strided_pass", st);
#define HX_LAUNCH_S(RR)                                                                  \
  case RR:                                                                               \
    hipLaunchKernelGGL((strided_pass<FWD, RR, A>), dim3(grid), dim3(256), 0, st, out, in, \
                       tw, m, log_n, a0, finish, items, il);                             \
    break;
  switch (R) {
    HX_LAUNCH_S(1)
    HX_LAUNCH_S(2)
    HX_LAUNCH_S(3)
    HX_LAUNCH_S(4)
    HX_LAUNCH_S(5)
    default:
      return hipErrorInvalidValue;
  }
#undef HX_LAUNCH_S
  return hipGetLastError();
}

// Experiment knob: HEXL_AMD_BLOCK_RE=3|4 picks the block_pass geometry for
// 12-stage tiles (default 3: 512 threads x 8 elements).
static int block_re() {
  static const int v = [] {
    const char* e = getenv("HEXL_AMD_BLOCK_RE");
    return (e && e[0] == '4') ? 4 : 3;
  }();
  return v;
}

template <bool FWD, class A>
static hipError_t launch_block(int TB, u64* out, const u64* in, const ulonglong2* tw,
                               const ModConst& m, u32 log_n, u32 finish, u64 batch,
                               const InvLast& il, hipStream_t st) {
  const u64 total = batch << log_n;
  const unsigned grid = (unsigned)((total + (1u << kTileLog) - 1) >> kTileLog);
  const u32 vec16 = (((uintptr_t)out | (uintptr_t)in) & 15) == 0 ? 1u : 0u;
  ScopedKernelTimer timer(FWD ? "ntt_fwd_block_pass" : "ntt_inv_block_pass", st);
#define HX_LAUNCH_B(T, RE)                                                              \
  hipLaunchKernelGGL((block_pass<FWD, T, RE, A>), dim3(grid), dim3(1 << (kTileLog - RE)), 0, \
                     st, out, in, tw, m, log_n, finish, total, vec16, il)
  switch (TB) {
    case 1: HX_LAUNCH_B(1, 4); break;
    case 2: HX_LAUNCH_B(2, 4); break;
    case 3: HX_LAUNCH_B(3, 4); break;
    case 4: HX_LAUNCH_B(4, 4); break;
    case 5: HX_LAUNCH_B(5, 4); break;
    case 6: HX_LAUNCH_B(6, 4); break;
    case 7: HX_LAUNCH_B(7, 4); break;
    case 8: HX_LAUNCH_B(8, 4); break;
    case 9: HX_LAUNCH_B(9, 4); break;
    case 10: HX_LAUNCH_B(10, 4); break;
    case 11: HX_LAUNCH_B(11, 4); break;
    case 12:
      if (block_re() == 3) {
        HX_LAUNCH_B(12, 3);
      } else {
        HX_LAUNCH_B(12, 4);
      }
      break;
    default:
      return hipErrorInvalidValue;
  }
#undef HX_LAUNCH_B
  return hipGetLastError();
}

// Lazy range policy: every multiplicand must stay below 2^62, i.e.
// (4 + 2*20) q (forward) and 64 q (inverse subtrees of up to 5 stages) < 2^62.


// Split the `top` leading stages into strided passes of at most 5 stages.
static int split_top(int top, int* sizes) {
  int n = 0;
  if (top <= 0) return 0;
  if (top <= 5) {
    sizes[n++] = top;
    return n;
  }
  int passes = (top + 3) / 4;
  int base = top / passes, extra = top % passes;
  for (int i = 0; i < passes; ++i) sizes[n++] = base + (i < extra ? 1 : 0);
  return n;
}

template <class A>
static hipError_t forward_impl(const NttTables& t, u64* result, const u64* operand, u64 batch,
                               u64 out_mf, hipStream_t st) {
  const int L = (int)t.log_n;
  const int TB = L < kTileLog ? L : kTileLog;
  int sizes[8];
  const int np = split_top(L - TB, sizes);
  const u64* src = operand;
  u32 a0 = 0;
  InvLast il{};
  for (int i = 0; i < np; ++i) {
    hipError_t e = launch_strided<true, A>(sizes[i], result, src, t.fwd, t.mod, t.log_n, a0, 0,
                                           batch, il, st);
    if (e != hipSuccess) return e;
    a0 += sizes[i];
    src = result;
  }
  return launch_block<true, A>(TB, result, src, t.fwd, t.mod, t.log_n, out_mf == 1 ? 2 : 1,
                               batch, il, st);
}

template <class A>
static hipError_t inverse_impl(const NttTables& t, u64* result, const u64* operand, u64 batch,
                               u64 out_mf, hipStream_t st) {
  const int L = (int)t.log_n;
  const int TB = L < kTileLog ? L : kTileLog;
  int sizes[8];
  const int np = split_top(L - TB, sizes);
  const u32 fin = out_mf == 1 ? 2 : 1;
  hipError_t e = launch_block<false, A>(TB, result, operand, t.inv, t.mod, t.log_n,
                                        np == 0 ? fin : 0, batch, t.inv_last, st);
  if (e != hipSuccess) return e;
  u32 a0 = (u32)(L - TB);
  for (int i = np - 1; i >= 0; --i) {
    a0 -= sizes[i];
    e = launch_strided<false, A>(sizes[i], result, result, t.inv, t.mod, t.log_n, a0,
                                 i == 0 ? fin : 0, batch, t.inv_last, st);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t ntt_forward_launch(const NttTables& t, u64* result, const u64* operand, u64 batch,
                              u64 out_mf, hipStream_t st) {
  if (batch == 0) return hipSuccess;
  if (t.mod.q < kLazyModulusBound) return forward_impl<Lazy>(t, result, operand, batch, out_mf, st);
  return forward_impl<Strict>(t, result, operand, batch, out_mf, st);
}

hipError_t ntt_inverse_launch(const NttTables& t, u64* result, const u64* operand, u64 batch,
                              u64 out_mf, hipStream_t st) {
  if (batch == 0) return hipSuccess;
  if (t.mod.q < kLazyModulusBound) return inverse_impl<Lazy>(t, result, operand, batch, out_mf, st);
  return inverse_impl<Strict>(t, result, operand, batch, out_mf, st);
}

}  // namespace hexl_amd
