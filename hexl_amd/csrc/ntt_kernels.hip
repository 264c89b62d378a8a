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
// 16-byte load per twiddle (one double per twiddle under the Fp64 arithmetic
// policy); the inverse table holds R[n]^-1 at the same heap index (the
// reference's stage-ordered inverse layout, ntt-internal.cpp:143-154, is kept
// on the host for the getters only).
//
// Everything is built from a register-resident "subtree": a thread owns 2^r
// elements and runs r stages on them with no communication.
//
//   strided_pass<R>   register-only pass (no LDS): each thread owns one column,
//     2^R elements N >> (a0 + R) apart, lanes = adjacent columns (coalesced),
//     twiddles wave-uniform (scalar loads).  HBM-bound: 0.68 ms for 2 GiB in +
//     2 GiB out.  The top pass for N >= 2^15 (5 + 11 stages for N = 2^16).
//   tile_pass<S, CB, TL>  One workgroup (2^(TL-3) threads x 8 elements) owns a tile
//     of 2^TL elements in LDS (TL = 12: 32 KiB, 4 workgroups = 32 waves per CU) and
//     runs S stages on it as ceil(S/3) subtree rounds separated by LDS
//     transposes.  The tile-index bits are [ sub-block | S stage bits | CB column
//     bits ]:
//       CB = 0      the tile is 2^TL contiguous coefficients = whole sub-blocks of
//                   heap level log2(N) - S ("bottom" stages; N <= 2^14 is this
//                   kernel alone, one HBM round trip: a 64 KiB tile for N = 2^13,
//                   a 128 KiB tile with 16 elements per thread for N = 2^14);
//       CB = TL - S all 2^S rows x 2^CB adjacent columns of one polynomial ("top" stages:
//                   the geometry is general, but no plan instantiates it -- the two-tile-pass
//                   plan it served measured 4 % slower, experiments/).
//     LDS slots are XOR-swizzled so that every ds_read_b64 / ds_write_b64 pattern
//     of every round is bank-conflict free (SQ_LDS_BANK_CONFLICT = 0); the swizzle
//     is linear over XOR, so an element's address is its thread's base address
//     XOR a compile-time constant.  Rounds whose gap is <= 64 elements exchange
//     data inside one wave and need no workgroup barrier.  The doubling on entry
//     and the output reduction / N^-1 scaling are fused into the first load and
//     the last store of a transform.
//     Bound by VALU issue (integer multiplies), not by HBM: see DESIGN.md.
//
// Values stay lazy as in the reference's Harvey butterflies
// (hexl/ntt/ntt-default.hpp:28-42, :112-125); see modarith.h for the eight
// arithmetic policies (Small, Fp64 / Fp64L, Lazy / Lazy32 / Lazy16, Harvey60, Strict;
// choose_policy below picks one per modulus).  Below the kernels: the multi-plan variants
// (polynomials of several moduli in one launch: RNS limbs, KeySwitch), and the
// host-side planning.  Canonical outputs (output_mod_factor == 1) are bit-identical to the
// reference; lazy outputs are congruent and inside the reference's ranges.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>
#include <mutex>

#include "internal.h"
#include "lazy_inverse.h"
#include "modarith.h"
#include "tile_geometry.h"
#include <cstddef>

#include "workspace.h"

// Translation units.  Compiled as it is this file holds everything.  hexl_amd/build.py
// compiles it several times in parallel instead, with -DHEXL_AMD_TU=<p>: p = 0..4 builds the
// kernels and launch code of ONE arithmetic policy (ArithPolicy value) behind two entry
// points, p = -1 the dispatch over them plus the process-wide state (tuning, profiling
// sink).  The template instantiations are what takes the compile time, and they are
// disjoint between policies.
#ifndef HEXL_AMD_TU
#define HX_TU_POLICY(p) 1
#define HX_TU_DISPATCH 1
#else
#define HX_TU_POLICY(p) (HEXL_AMD_TU == (p))
#define HX_TU_DISPATCH (HEXL_AMD_TU < 0)
#if defined(HEXL_AMD_PHASE_PROFILE)
#error "developer builds with device-side diagnostics are single translation units"
#endif
#endif

namespace hexl_amd {

#if HX_TU_DISPATCH
thread_local ProfileSink* g_profile = nullptr;
#endif

// Developer diagnostic (tools/phase_profile.py): per-wave s_memtime stamps at the
// phase boundaries of tile_pass.  Compiled out of the product build.
#ifdef HEXL_AMD_PHASE_PROFILE
__device__ unsigned long long* g_phase_buf = nullptr;  // [block][wave][16]
#define HX_STAMP(i)                                                                      \
  do {                                                                                   \
    if ((threadIdx.x & 63) == 0 && g_phase_buf)                                          \
      g_phase_buf[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] =            \
          __builtin_readcyclecounter();                                                  \
  } while (0)
#define HX_PROFILE_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
extern "C" int hexl_amd_debug_set_phase_buf(void* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_buf), &buf, sizeof(buf));
}
#else
#define HX_STAMP(i)
#define HX_PROFILE_WAIT_VMEM()
#endif

// Kernel `flags` argument.  Bits 0-1: 0 = more passes follow, 1 = this pass ends
// the network (lazy output range), 2 = ends the network with canonical output in
// [0,q).  Bit 2: this pass starts the network (its input is the caller's data).
constexpr u32 kFinishMask = 3;
constexpr u32 kFirstPass = 4;
// Bit 3 (with bit 2): the input words are arbitrary 64-bit values and are reduced modulo q
// on load (multi-plan launches with a source map, see MultiMap).
constexpr u32 kReduceFirst = 8;
// Bit 4 (with bit 2; multi-plan launches whose map carries rounding constants): the input words
// are first rounded to the special modulus of a KeySwitch -- x' = (x + q_k / 2) mod q_k, the
// reference's AddUIntMod-free form of key-switch-internal.cpp:146-160 -- then reduced modulo q if
// bit 3 says so, and the correction q - (q_k / 2 mod q) is added (:162-175): what ks_round_kernel
// wrote to a buffer of its own before round 6.
constexpr u32 kRoundFirst = 16;
// Bits 8 and up (forward passes of the bounded members of the Lazy family only, modarith.h): bit
// 8 + s set = the x operands of stage s of this pass lose kLimit/2 * q by one sign-tested
// subtraction before the stage (the host walks the bound of the values through the whole network
// and marks the stages whose growth would otherwise pass kLimit * q: forward_stage_masks).
constexpr u32 kStageMaskShift = 8;

// Global access as wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte
// offset: `global_load/store v, v_off, s[base]`, no 64-bit vector address math and
// no address registers kept live between the load and the in-place store.
//
// Access kinds.  kStream = nontemporal (the `nt` bit): a pass streams its data
// exactly once, and marking the accesses so is worth 4-6 % on the HBM-bound strided
// pass and 3 % on the forward tile pass; the single-plan inverse tile pass streams too since round 6
// (0.7 % of the headline step; the multi-plan inverse keeps plain accesses).
enum : int { kPlain = 0, kStream = 1, kRaw = 2 };  // kRaw: raw_load_b64 / raw_store_b64 below (nontemporal)

template <int KIND>
__device__ __forceinline__ u64 ld_global(const u64* p) {
  if (KIND == kStream) return __builtin_nontemporal_load(p);
  return *p;
}
template <int KIND>
__device__ __forceinline__ void st_global(u64* p, u64 v) {
  if (KIND == kStream)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}
template <int KIND>
__device__ __forceinline__ u64 load_global(const u64* uniform_base, u32 byte_off) {
  return ld_global<KIND>(
      reinterpret_cast<const u64*>(reinterpret_cast<const char*>(uniform_base) + byte_off));
}
template <int KIND>
__device__ __forceinline__ void store_global(u64* uniform_base, u32 byte_off, u64 v) {
  st_global<KIND>(reinterpret_cast<u64*>(reinterpret_cast<char*>(uniform_base) + byte_off), v);
}

// The caller's words on entry of a first pass, as the flags say (kReduceFirst, kRoundFirst).
template <int E>
__device__ __forceinline__ void entry_words(u64* x, u32 flags, const ModConst& m) {
  if (flags & kRoundFirst) {
    // (ks_round_kernel's arithmetic, word for word)
    const u64 fix = m.q - reduce_any_straight(m.rnd_half, m.q, m.barrett);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const u64 t = x[e] + m.rnd_half;
      u64 v = csub(t - __umul64hi(t, m.rnd_barrett) * m.rnd_qk, m.rnd_qk);
      if (flags & kReduceFirst) v = reduce_any(v, m.q, m.barrett);
      x[e] = v + fix;
    }
  } else if (flags & kReduceFirst) {
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = reduce_any_straight(x[e], m.q, m.barrett);
  }
}

// Vector memory operations OUTSIDE the compiler's s_waitcnt bookkeeping (the persistent tile walk).
// hipcc counts the operations in flight per basic block and merges the counts where paths join by
// assuming the FEWEST newer operations: at the head of the walk's loop (entered from the prologue,
// with nothing behind the tile's loads, and from the back edge, with the previous tile's sixteen
// stores behind them) every wait for a load became a wait for those stores' acknowledgements too --
// vmcnt(13) ... vmcnt(0) where the back edge needs vmcnt(29) ... vmcnt(16) -- and under a branch
// the same happened to the prefetch.  These loads and stores are invisible to that pass; the walk
// waits for them itself (raw_wait: exact counts, every use of a loaded value behind the wait).
__device__ __forceinline__ u64 raw_load_b64(const u64* uniform_base, u32 byte_off) {
  u64 v;
  asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(v) : "v"(byte_off), "s"(uniform_base) : "memory");
  return v;
}
__device__ __forceinline__ void raw_store_b64(u64* uniform_base, u32 byte_off, u64 v) {
  asm volatile("global_store_dwordx2 %0, %1, %2 nt" : : "v"(byte_off), "v"(v), "s"(uniform_base) : "memory");
}
// wait until at most NEWER of the raw operations issued after the awaited ones are in flight, then
// pin the E words behind the wait (volatile statements keep their order)
template <int NEWER, int E>
__device__ __forceinline__ void raw_wait(u64* x) {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NEWER) : "memory");
#pragma unroll
  for (int i = 0; i < E; ++i) asm volatile("" : "+v"(x[i]));
}

// ---------------------------------------------------------------------------
// Register subtrees
// ---------------------------------------------------------------------------

// A twiddle as it sits in the device table: (W, Shoup factor) for the integer
// policies, ONE double (balanced W) for Fp64.
template <class A, bool FP = A::kFp>
struct TwOf {
  typedef ulonglong2 T;
};
template <class A>
struct TwOf<A, true> {
  typedef double T;
};
template <class A>
using TwT = typename TwOf<A>::T;

template <class A>
__device__ __forceinline__ void bf_fwd(u64& x, u64& y, const TwT<A>& w, const ModConst& m) {
  if constexpr (A::kFp)
    fwd_butterfly_fp(x, y, w, m);
  else
    fwd_butterfly<A>(x, y, w.x, w.y, m);
}
// (not the Lazy policy: its inverse network is lazy_inverse.h)
template <class A>
__device__ __forceinline__ void bf_inv(u64& x, u64& y, const TwT<A>& w, const ModConst& m) {
  if constexpr (A::kFp)
    inv_butterfly_fp(x, y, w, m);
  else
    inv_butterfly<A>(x, y, w.x, w.y, m);
}
// MONT: the sum branch without a modular product (scale_by_inverse_degree; Strict, Harvey60)
template <class A, bool MONT>
__device__ __forceinline__ void bf_inv_last(u64& x, u64& y, const InvLast& il, const ModConst& m) {
  if constexpr (A::kFp)
    inv_butterfly_last_fp(x, y, fp_bits_to_double(il.n1), fp_bits_to_double(il.n1w), m);
  else
    inv_butterfly_last<A, (MONT && !A::kSmall)>(x, y, il, m);
}
// Fp64: full reduction of all E elements (end of a run of stages); nothing otherwise.
template <class A, int E>
__device__ __forceinline__ void fp_bound_all(u64* x, const ModConst& m) {
  if constexpr (A::kFp) {
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = fp_bound<A>(x[e], m);
  }
}

// The 2^R - 1 twiddles of the subtree rooted at heap node `node`, fetched
// before the first butterfly so that their latencies overlap (left to itself
// the compiler issues one load + one full wait per stage).  wv[2^v + g] is the
// twiddle of group g at depth v.
// The tables are read through the CONSTANT address space (they are never written while a
// transform runs): a wave-uniform index is then a scalar load whatever the compiler can or
// cannot prove about aliasing stores (in the multi-plan kernels with 12- and 13-stage tiles
// it could not, and every twiddle came through the vector memory path into VGPRs).
typedef const __attribute__((address_space(4))) u64 ConstWord;
typedef const __attribute__((address_space(4))) double ConstDouble;
__device__ __forceinline__ ulonglong2 load_twiddle(const ulonglong2* tw, u32 i) {
  const ConstWord* c = (const ConstWord*)(unsigned long long)tw;
  ulonglong2 w;
  w.x = c[2 * (u64)i];
  w.y = c[2 * (u64)i + 1];
  return w;
}
__device__ __forceinline__ double load_twiddle(const double* tw, u32 i) {
  const ConstDouble* c = (const ConstDouble*)(unsigned long long)tw;
  return c[i];
}
// CTW: through the constant address space (the multi-plan kernels).  The single-plan kernels
// keep the plain form -- their table pointer is a __restrict__ kernel argument, which gets
// them scalar loads already, and the constant-space form costs the 11-stage forward tile
// pass a spilled register pair (0.93 -> 0.95 ms).
template <int R, bool CTW = false, class T>
__device__ __forceinline__ void load_twiddles(T* wv, const T* __restrict__ tw, u32 node) {
#pragma unroll
  for (int v = 0; v < R; ++v)
#pragma unroll
    for (int g = 0; g < (1 << v); ++g) {
      if constexpr (CTW)
        wv[(1 << v) + g] = load_twiddle(tw, (node << v) + g);
      else
        wv[(1 << v) + g] = tw[(node << v) + g];
    }
}

// Bounded members of the Lazy family: the x operands of a stage whose bit is set in `smask` are
// brought below kLimit/2 * q first (a uniform branch: the mask is a kernel argument).
template <class A>
constexpr bool bounded_lazy() { return A::kLazy && A::kLimit < kLazyLimit; }
template <class A>
constexpr int bounded_lazy_shift() {  // kLimit/2 * q = 2q << shift
  int s = 0;
  while ((4 << s) < A::kLimit) ++s;
  return s;
}
template <int R, int V, int G0, int G1, class A>
__device__ __forceinline__ void fwd_bound_level(u64* x, const ModConst& m, u32 smask) {
  if constexpr (bounded_lazy<A>()) {
    if (smask & (1u << V)) {
      constexpr int half = 1 << (R - 1 - V);
#pragma unroll
      for (int g = G0; g < G1; ++g)
#pragma unroll
        for (int j = 0; j < half; ++j) {
          x[g * 2 * half + j] = lazy_csub(x[g * 2 * half + j], m, bounded_lazy_shift<A>());
          // (32 elements per thread: keep the scheduler from issuing all the additions before
          // the first selection -- their sums would all be live at once and spill)
          if (R >= 5 && (j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
  }
}

// Hooks of the persistent tile walk (tile_walk): a callable run at one point inside a round --
// forward: in front of the LAST stage of the last round; inverse: in round 1 between its LDS
// load and its arithmetic -- where the next tile's loads are issued.
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};

// R forward stages on x[0 .. 2^R): stage v pairs elements 2^(R-1-v) apart.  smask: see
// fwd_bound_level (bit v = stage v of this subtree).
template <int R, class A, class Hook = NoHook>
__device__ __forceinline__ void fwd_subtree(u64* x, const TwT<A>* wv, const ModConst& m, u32 smask = 0,
                                            const Hook& hook = Hook()) {
  static_assert(!A::kFp || R <= A::kFwdRun, "Fp64 forward run too long");
#pragma unroll
  for (int v = 0; v < R; ++v) {
    const int half = 1 << (R - 1 - v);
    if (v == R - 1) hook();
    if constexpr (bounded_lazy<A>()) {
      if (smask & (1u << v)) {
#pragma unroll
        for (int g = 0; g < (1 << v); ++g)
#pragma unroll
          for (int j = 0; j < half; ++j)
            x[g * 2 * half + j] = lazy_csub(x[g * 2 * half + j], m, bounded_lazy_shift<A>());
      }
    }
#pragma unroll
    for (int g = 0; g < (1 << v); ++g) {
      const TwT<A> w = wv[(1 << v) + g];
#pragma unroll
      for (int j = 0; j < half; ++j)
        bf_fwd<A>(x[g * 2 * half + j], x[g * 2 * half + j + half], w, m);
      // Fp64, deep subtrees: keep the scheduler from interleaving every product of a
      // stage (their temporaries would all be live at once and spill).
      if (A::kFp && R >= 4) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// R inverse stages (deepest level first).  With LAST the v == 0 stage is the
// root of the whole transform and folds N^-1 in (ntt-radix-2.cpp:490-509).
// B, T (Lazy policy only): bound of the inputs and what the consumer of the outputs takes, in
// units of q -- the subtree is then lazy_inverse.h's, which subtracts only where its
// compile-time range bookkeeping says a value would pass 2^63.
// Fp64: the sums double, so every kFpInvRun stages all elements are fully reduced,
// and again at exit (not after LAST: the finish does it).
// MONT (with LAST): the transform has at least 64 coefficients, the N^-1 scaling of the sum branch
// is scale_by_inverse_degree (modarith.h).
// (B = T = 0: what a pass hands over under the policy's limit)
// FP_EXIT (Fp64 family): all elements are fully reduced at exit (the caller knows whether the
// run of stages since the last reduction may go on: Rounds::fp_inv_reduce_after).
template <int R, class A, bool LAST, int B = 0, int T = 0, bool MONT = true, bool FP_EXIT = !LAST>
__device__ __forceinline__ void inv_subtree(u64* x, const TwT<A>* wv, const ModConst& m,
                                            const InvLast& il) {
  static_assert(R <= 5, "register subtrees are at most 5 stages deep");
  if constexpr (A::kLazy) {
    constexpr int kHand = lazy_handover(A::kLimit);
    inv_subtree_lazy<R, (B ? B : kHand), (T ? T : kHand), LAST, MONT, A::kLimit>(x, wv, m, il);
    return;
  } else {
#pragma unroll
    for (int v = R - 1; v >= 0; --v) {
      const int half = 1 << (R - 1 - v);
      const int t = R - 1 - v;  // execution order
#pragma unroll
      for (int g = 0; g < (1 << v); ++g) {
        if (LAST && v == 0) {
#pragma unroll
          for (int j = 0; j < half; ++j) bf_inv_last<A, MONT>(x[j], x[j + half], il, m);
        } else {
          const TwT<A> w = wv[(1 << v) + g];
#pragma unroll
          for (int j = 0; j < half; ++j)
            bf_inv<A>(x[g * 2 * half + j], x[g * 2 * half + j + half], w, m);
        }
        // Deep subtrees: stop the scheduler from interleaving every butterfly of a
        // stage (it would keep all their temporaries live at once and spill).
        if (R >= 4) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (A::kFp) {
        if ((t + 1) % A::kInvRun == 0 && v > 0) fp_bound_all<A, (1 << R)>(x, m);
      }
    }
    if (FP_EXIT) fp_bound_all<A, (1 << R)>(x, m);
  }
}

// One level of a subtree on its own, groups [G0, G1): used by the 5-stage strided
// pass, which fetches its 31 wave-uniform twiddles in four batches of <= 8 (each
// requested one batch ahead of its use) because all of them at once do not fit
// the SGPR file.
template <int R, int V, int G0, int G1, class A>
__device__ __forceinline__ void fwd_level(u64* x, const TwT<A>* wl, const ModConst& m) {
  constexpr int half = 1 << (R - 1 - V);
#pragma unroll
  for (int g = G0; g < G1; ++g) {
#pragma unroll
    for (int j = 0; j < half; ++j) {
      bf_fwd<A>(x[g * 2 * half + j], x[g * 2 * half + j + half], wl[g - G0], m);
      if (A::kFp && (j & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // see fwd_subtree
    }
    if (A::kFp) __builtin_amdgcn_sched_barrier(0);
  }
}

// Inverse level V; LAST folds N^-1 in (then V == 0).  Lazy: the level of lazy_inverse.h's
// schedule SC (execution stage R-1-V).
// (only the 5-stage strided subtree comes here: N >= 2^13, the N^-1 scaling of the sum branch is
// scale_by_inverse_degree)
template <int R, int V, int G0, int G1, class A, bool LAST, class SC = void>
__device__ __forceinline__ void inv_level(u64* x, const TwT<A>* wl, const ModConst& m,
                                          const InvLast& il) {
  if constexpr (A::kLazy) {
    inv_level_lazy<SC, R, R - 1 - V, G0, G1, LAST, true>(x, wl, m, il);
  } else {
    constexpr int half = 1 << (R - 1 - V);
#pragma unroll
    for (int g = G0; g < G1; ++g)
#pragma unroll
      for (int j = 0; j < half; ++j) {
        u64& a = x[g * 2 * half + j];
        u64& b = x[g * 2 * half + j + half];
        if (LAST)
          bf_inv_last<A, true>(a, b, il, m);
        else
          bf_inv<A>(a, b, wl[g - G0], m);
      }
  }
}

// CTW: through the constant address space (see load_twiddles): in the multi-plan kernels the
// table pointer comes out of a kernel-argument array and is not `__restrict__`, and without it
// all 31 wave-uniform twiddles of a 5-stage subtree came through the vector memory path into
// VGPRs (121-126 VGPRs, ~30 SGPRs; the bounded members of the Lazy family spilled).
template <int COUNT, bool CTW = false, class T>
__device__ __forceinline__ void load_twiddle_run(T* w, const T* __restrict__ tw, u32 first) {
#pragma unroll
  for (int i = 0; i < COUNT; ++i) {
    if constexpr (CTW)
      w[i] = load_twiddle(tw, first + i);
    else
      w[i] = tw[first + i];
  }
}

// The 5-stage subtrees of the strided pass, twiddles streamed (see fwd_level).
template <class A, bool CTW = false>
__device__ __forceinline__ void fwd_subtree5_streamed(u64* x, const TwT<A>* __restrict__ tw,
                                                      u32 node, const ModConst& m, u32 smask) {
  TwT<A> w0, w1[2], w2[4], w3[8], w4a[8], w4b[8];
  load_twiddle_run<1, CTW>(&w0, tw, node);
  load_twiddle_run<2, CTW>(w1, tw, node << 1);
  load_twiddle_run<4, CTW>(w2, tw, node << 2);
  load_twiddle_run<8, CTW>(w3, tw, node << 3);
  fwd_bound_level<5, 0, 0, 1, A>(x, m, smask);
  fwd_level<5, 0, 0, 1, A>(x, &w0, m);
  fwd_bound_level<5, 1, 0, 2, A>(x, m, smask);
  fwd_level<5, 1, 0, 2, A>(x, w1, m);
  fwd_bound_level<5, 2, 0, 4, A>(x, m, smask);
  fwd_level<5, 2, 0, 4, A>(x, w2, m);
  load_twiddle_run<8, CTW>(w4a, tw, node << 4);
  fwd_bound_level<5, 3, 0, 8, A>(x, m, smask);
  fwd_level<5, 3, 0, 8, A>(x, w3, m);
  load_twiddle_run<8, CTW>(w4b, tw, (node << 4) + 8);
  fwd_bound_level<5, 4, 0, 16, A>(x, m, smask);
  fwd_level<5, 4, 0, 8, A>(x, w4a, m);
  fwd_level<5, 4, 8, 16, A>(x, w4b, m);
}

template <class A, bool LAST, bool CTW = false>
__device__ __forceinline__ void inv_subtree5_streamed(u64* x, const TwT<A>* __restrict__ tw,
                                                      u32 node, const ModConst& m,
                                                      const InvLast& il) {
  // Lazy: the schedule of a 5-stage subtree entered and (unless LAST) left below kLazyHandOver
  // -- three quotient estimates where the sums of the deepest chains would pass the limit.
  // Fp64: full reduction after the third stage and at exit.
  constexpr int kLim = A::kLazy ? A::kLimit : kLazyLimit;
  using SC = InvSchedOf<5, lazy_handover(kLim), lazy_handover(kLim), LAST, kLim>;
  TwT<A> w0, w1[2], w2[4], w3[8], w4a[8], w4b[8];
  load_twiddle_run<8, CTW>(w4a, tw, node << 4);
  load_twiddle_run<8, CTW>(w4b, tw, (node << 4) + 8);
  inv_level<5, 4, 0, 8, A, false, SC>(x, w4a, m, il);
  load_twiddle_run<8, CTW>(w3, tw, node << 3);
  inv_level<5, 4, 8, 16, A, false, SC>(x, w4b, m, il);
  load_twiddle_run<4, CTW>(w2, tw, node << 2);
  load_twiddle_run<2, CTW>(w1, tw, node << 1);
  load_twiddle_run<1, CTW>(&w0, tw, node);
  inv_level<5, 3, 0, 8, A, false, SC>(x, w3, m, il);
  inv_level<5, 2, 0, 4, A, false, SC>(x, w2, m, il);
  if constexpr (A::kFp && A::kInvRun < 5) fp_bound_all<A, 32>(x, m);  // (three stages done)
  inv_level<5, 1, 0, 2, A, false, SC>(x, w1, m, il);
  inv_level<5, 0, 0, 1, A, LAST, SC>(x, &w0, m, il);
  if constexpr (A::kLazy) inv_exit_lazy<SC, 5>(x, m);
  if (!LAST) fp_bound_all<A, 32>(x, m);
}

// ---------------------------------------------------------------------------
// strided_pass: R stages whose subtree roots sit at heap level a0 (registers only)
// ---------------------------------------------------------------------------
// Work item -> (poly b, subtree h in [0, 2^a0), column c in [0, S)),
// S = N >> (a0 + R); element e of the item is at b*N + (h*2^R + e)*S + c.
// Lanes map to consecutive columns: every access is a coalesced 512-byte wave
// access and the twiddles are wave-uniform (scalar loads).
// Occupancy floor (workgroups of 4 waves per CU = waves per SIMD): the data alone
// is 2 * 2^R VGPRs, so 4 waves/SIMD (<= 128 VGPRs) is the most a 5-stage subtree
// can have; shallower ones get 6.
template <int R>
constexpr int strided_min_waves() { return R >= 5 ? 4 : 6; }

// End of a strided pass: finish (FIN: 0 = more passes follow, 1 = lazy output range,
// 2 = canonical) and store the E rows of the column.
template <bool FWD, int FIN, int E, class A, int STK>
__device__ __forceinline__ void strided_store(u64* out, u64* x, u64 base, u32 log_s,
                                              const ModConst& m) {
  // (row offsets and base + lane are recomputed here from opaque copies: shared with the
  // loads' they stay live in 2 x 32 SGPRs and a VGPR pair across the whole kernel)
  asm volatile("" : "+s"(log_s));
  // (the lane index from v_mbcnt: not even threadIdx has to survive the arithmetic)
  const u32 lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const u64 vbase = base + lane;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    u64 v = x[e];
    if constexpr (A::kFp) {
      // every pass hands over fully reduced values; the last one canonical residues
      if (FWD) v = fp_pass_end(fp_bits_to_double(v), m, FIN != 0);
      else if (FIN) v = inv_finish<A>(v, m, FIN == 2);
    } else if (FIN) {
      v = FWD ? fwd_finish<A>(v, m, FIN == 2) : inv_finish<A>(v, m, FIN == 2);
    }
    st_global<STK>(&out[vbase + ((u64)e << log_s)], v);
    if (E > 8 && (e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
}

// The work of one 256-thread workgroup `bid` of the pass (4 waves x 64 columns).
// LAST (inverse only): the pass contains the root stage of the transform (a0 == 0).
// LDK / STK: access kinds of the loads and stores (see ld_global).
// The data pointers are not __restrict__: transforms run in place (out == in).
// DATA_FIRST (the multi-plan kernels): the data loads are issued before anything that
// depends on the plan (twiddle table pointer, modulus constants), which those kernels fetch
// from device memory through a chain of scalar loads -- the chain then resolves under the
// data loads' latency instead of in front of it.
template <bool FWD, int R, class A, bool LAST, int LDK, int STK, bool DATA_FIRST = false,
          bool CTW = false>
__device__ __forceinline__ void strided_body(u64* out, const u64* in,
                                             const ulonglong2* __restrict__ tw, const ModConst& m,
                                             u32 log_n, u32 a0, u32 flags, u32 bid,
                                             const InvLast& il) {
  constexpr int E = 1 << R;
  const u32 finish = flags & kFinishMask;
  // A wave covers 64 adjacent columns of ONE subtree, so everything but the lane's
  // column offset is wave-uniform, in particular the twiddles (scalar loads into
  // SGPRs).
  const u32 lane = threadIdx.x & 63;
  const u64 wi = __builtin_amdgcn_readfirstlane(bid * 4 + (threadIdx.x >> 6));  // wave index
  const u32 log_s = log_n - a0 - R;
  const u32 log_wpp = log_n - R - 6;  // log2(waves per polynomial)
  const u64 b = wi >> log_wpp;
  const u32 rem = ((u32)wi & ((1u << log_wpp) - 1)) << 6;  // first work item of the wave
  const u32 h = rem >> log_s;
  const u32 c0 = rem & ((1u << log_s) - 1);
  const u64 base = (b << log_n) + ((u64)h << (log_s + R)) + c0;
  const u64 vbase = base + lane;  // (SGPR base + 32-bit lane offset measured 3% slower here)
  const u32 node = (1u << a0) + h;
  const TwT<A>* __restrict__ twa = reinterpret_cast<const TwT<A>*>(tw);
  TwT<A> wv[E];
  if constexpr (R < 5 && !DATA_FIRST) load_twiddles<R, CTW>(wv, twa, node);

  u64 x[E];
  // loads go out at raised priority (forward: 0.70 -> 0.68 ms; the inverse pass
  // measured 2 % slower with it)
  if (FWD) __builtin_amdgcn_s_setprio(3);
#pragma unroll
  for (int e = 0; e < E; ++e) x[e] = ld_global<LDK>(&in[vbase + ((u64)e << log_s)]);
  if (FWD) __builtin_amdgcn_s_setprio(0);
  if constexpr (R < 5 && DATA_FIRST) load_twiddles<R, CTW>(wv, twa, node);
  if (flags & kFirstPass) {
    // (source maps exist in the multi-plan kernels only -- CTW -- : compiled into the single-plan
    // kernels too, the rounding path cost the Small inverse tile pass its 64-register budget,
    // 36 bytes of scratch and 18-34 % of its time)
    if constexpr (CTW && FWD) {  // (... and on forward transforms only: ntt_multi_launch)
      if (flags & (kReduceFirst | kRoundFirst)) entry_words<E>(x, flags, m);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = to_internal<A>(x[e], m);
  }

  if (FWD) {
    if constexpr (R < 5)
      fwd_subtree<R, A>(x, wv, m, flags >> kStageMaskShift);
    else
      fwd_subtree5_streamed<A, CTW>(x, twa, node, m, flags >> kStageMaskShift);
  } else {
    if constexpr (R < 5)
      inv_subtree<R, A, LAST>(x, wv, m, il);
    else
      inv_subtree5_streamed<A, LAST, CTW>(x, twa, node, m, il);
  }
  // The finish kind is uniform: three straight-line store loops behind scalar branches, each
  // element finished right in front of its store (a select per element costs 3 instructions
  // of the 13; finishing all 32 elements first keeps 64 VGPRs live beside the temporaries).
  if (finish == 2)
    strided_store<FWD, 2, E, A, STK>(out, x, base, log_s, m);
  else if (finish)
    strided_store<FWD, 1, E, A, STK>(out, x, base, log_s, m);
  else
    strided_store<FWD, 0, E, A, STK>(out, x, base, log_s, m);
}

template <bool FWD, int R, class A, bool LAST>
__global__ void __launch_bounds__(256, (strided_min_waves<R>()))
strided_pass(u64* out, const u64* in, const ulonglong2* __restrict__ tw, ModConst m, u32 log_n,
             u32 a0, u32 flags, u64 items, InvLast il) {
  // items is a multiple of 64 (>= 64 columns per subtree): waves are whole.
  // XCD-aware block order: the dispatcher deals consecutive workgroups round-robin
  // to the 8 XCDs; remapped, each XCD walks one contiguous eighth of the work, so
  // the rows it has open in HBM are few and long (measured: 0.79 -> 0.71 ms at full
  // occupancy; without it the pass only reaches 0.70 ms when occupancy is throttled
  // to 2 waves per SIMD).
  u32 bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  if ((u64)bid * 256 + threadIdx.x >= items) return;
  strided_body<FWD, R, A, LAST, kStream, kStream>(out, in, tw, m, log_n, a0, flags, bid, il);
}

// Polynomials of several moduli in one launch (RNS limbs, the per-modulus
// transforms of KeySwitch): the workgroup's polynomial selects the plan, whose
// parameters are read from its device-resident copy instead of kernel arguments.
struct MultiCtx {
  const PlanDev* p[kMaxMultiPlans];
  // The twiddle tables of each plan, once more: as kernel-argument pointers the compiler
  // knows them to be global memory and serves wave-uniform twiddles with scalar loads;
  // fetched from the plan's device copy they were generic pointers, every twiddle a
  // flat_load into VGPRs (the multi-plan tile pass 17-32 % slower than the single-plan one).
  const ulonglong2* tw_fwd[kMaxMultiPlans];
  const ulonglong2* tw_inv[kMaxMultiPlans];
  uint8_t policy[kMaxMultiPlans];  // ArithPolicy of each plan
  u32 one_policy;  // every plan is of one arithmetic policy (the usual case): no workgroup leaves
  MultiMap map;
};
// The plan of polynomial `poly`, or nullptr when it is not of policy `want` (a launch
// sequence serves one arithmetic policy; workgroups of the others leave at once).
// Entry `idx` of a byte table that lives in the kernel arguments, fetched as the aligned
// dword that holds it: a SCALAR load (there is no scalar byte load on gfx950; as a byte
// access it becomes a vector load + wait + readfirstlane, ~1 us in front of everything the
// workgroup does -- two of them in a row made the multi-plan tile pass 17-32 % slower than
// the single-plan one).  The tables are 4-byte aligned inside their structs.
__device__ __forceinline__ u32 kernarg_byte(const uint8_t* tab, u32 idx) {
  idx = __builtin_amdgcn_readfirstlane(idx);
  const u32 w = reinterpret_cast<const u32*>(tab)[idx >> 2];
  return (w >> ((idx & 3) * 8)) & 0xffu;
}
static_assert(offsetof(MultiMap, plan_tab) % 4 == 0 && offsetof(MultiMap, src_tab) % 4 == 0 &&
                  offsetof(MultiCtx, policy) % 4 == 0 && offsetof(MultiCtx, map) % 4 == 0,
              "byte tables must be dword aligned for kernarg_byte");
__device__ __forceinline__ const PlanDev* multi_plan(const MultiCtx& mc, u32 poly, int want,
                                                     u32& plan_index) {
  // (the plan index goes through readfirstlane so that the pointer is fetched with a
  // cleanly aligned scalar load: left to itself the compiler derived its address from
  // the byte address of policy[k], base + k plus an offset of 7 k, for an s_load whose
  // base must be dword aligned -- an aperture violation for k % 4 != 0)
  const u32 k = kernarg_byte(mc.map.plan_tab, (poly / mc.map.inner) % mc.map.period);
  plan_index = k;
  // (with one policy in the launch the check -- one more scalar load in front of the data
  // loads -- is skipped)
  if (mc.one_policy) return mc.p[k];
  const u32 pol = kernarg_byte(mc.policy, k);
  return pol == (u32)want ? mc.p[k] : nullptr;
}
// Source map of a multi-plan launch (MultiMap): where polynomial `poly` of the FIRST pass
// reads its input, as an adjusted `in` pointer (the bodies index by output position), and
// whether its words are reduced on load.
__device__ __forceinline__ const u64* multi_source(const MultiCtx& mc, u32 poly, u32 log_n,
                                                   const u64* in, u32& flags) {
  if (mc.map.src_stride == 0 || !(flags & kFirstPass)) return in;
  const u32 s = poly % mc.map.period, grp = poly / mc.map.period;
  const u32 e = kernarg_byte(mc.map.src_tab, s);
  const u32 src = grp * mc.map.src_stride + (e & 0x7f);
  if (e & 0x80) flags |= kReduceFirst;
  if (mc.map.rnd_qk) flags |= kRoundFirst;
  return in + (((long long)src - (long long)poly) << log_n);
}
template <class A>
constexpr int policy_id() {
  return A::kSmall ? kPolicySmall
         : A::kFp  ? (A::kInvRun > 3 ? kPolicyFp64L : kPolicyFp64)
         : (A::kLazy && A::kLimit == 32) ? kPolicyLazy32
         : (A::kLazy && A::kLimit == 16) ? kPolicyLazy16
         : A::kLazy ? kPolicyLazy
         : A::kH60  ? kPolicyHarvey60
                    : kPolicyStrict;
}

template <bool FWD, int R, class A, bool LAST>
__global__ void __launch_bounds__(256, (strided_min_waves<R>()))
strided_pass_multi(u64* out, const u64* in, MultiCtx mc, u32 log_n, u32 a0, u32 flags, u64 items) {
  u32 bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  if ((u64)bid * 256 + threadIdx.x >= items) return;
  // a workgroup's 4 waves lie in one polynomial (>= 256 columns per polynomial: N >= 2^13)
  const u32 poly = (bid * 4u) >> (log_n - R - 6);
  u32 k;
  const PlanDev* __restrict__ pd = multi_plan(mc, poly, policy_id<A>(), k);
  if (!pd) return;
  ModConst m = pd->mod;
  m.rnd_qk = mc.map.rnd_qk;  // (rounding on load: see kRoundFirst)
  m.rnd_barrett = mc.map.rnd_barrett;
  m.rnd_half = mc.map.rnd_half;
  const InvLast il = pd->il;
  in = multi_source(mc, poly, log_n, in, flags);
  // (data first -- see strided_body -- except where the 5-stage forward subtree of a bounded
  // member of the Lazy family has no registers to spare for it)
  strided_body<FWD, R, A, LAST, kStream, kStream, !(FWD && R >= 5 && bounded_lazy<A>()), true>(
      out, in, FWD ? mc.tw_fwd[k] : mc.tw_inv[k], m, log_n, a0, flags, bid, il);
}

// ---------------------------------------------------------------------------
// tile_pass: S stages on a 2^TL-element tile staged through LDS
// ---------------------------------------------------------------------------
// Tile size is a template parameter TL (log2 elements): 10 -> 1024 elements (8 KiB
// of LDS, 128 threads = 2 waves, 16 workgroups per CU) for N <= 2^10, 11 -> 2048
// elements for the 11-stage bottom pass of N = 2^13..2^16, 12 -> 4096 elements
// (32 KiB, 512 threads, 4 workgroups per CU) otherwise.  Either way a thread holds
// 8 elements, a round is 3 stages, and 32 waves fit a CU.
// (elements per thread, round structure, LDS swizzle and tile indices: tile_geometry.h)

// vt >> w for vt = s*threads + tid, visibly uniform when 2^w >= threads.
template <int w, int TL, int RE>
__device__ __forceinline__ u32 vt_high(int s, u32 tid) {
  constexpr int kThreadsLog = TL - RE;
  if (w >= kThreadsLog) return (u32)s >> (w - kThreadsLog);
  return (((u32)s << kThreadsLog) + tid) >> w;
}

// Where the tile lives: element p is at base + (p >> CB << log_row) + (p & cols).
struct TileGeom {
  u64 base;
  u32 log_row;    // 0 for CB == 0
  u32 a0;         // heap level of the subtree roots of this pass
  u32 tile_blk0;  // index within its polynomial of the tile's first sub-block
};

// Global address of tile element p = p0 + dp, split into a per-thread 32-bit byte
// offset (from p0) and a wave-uniform 64-bit part (tile base + dp, dp a
// compile-time multiple of 2^CB): the uniform part lives in SGPRs and the access
// is `global_load/store v, v_off, s[base]` with no per-element vector arithmetic.
template <int CB>
__device__ __forceinline__ u32 tile_byte_offset(const TileGeom& g, u32 p0) {
  if (CB == 0) return p0 << 3;
  return (((p0 >> CB) << g.log_row) + (p0 & ((1u << CB) - 1))) << 3;
}
template <int CB>
__device__ __forceinline__ u64 tile_uniform_offset(const TileGeom& g, u32 dp) {
  if (CB == 0) return g.base + dp;
  return g.base + ((u64)(dp >> CB) << g.log_row);
}

// Twiddles of round j for every sub-run this thread owns in that round, all
// requested before the round's first butterfly.  Wave-uniform twiddles (gap >= 64)
// come through scalar loads (no VGPR cost), the others through per-lane loads.
template <int S, int CB, int TL, int j, bool CTW = false, class T>
__device__ __forceinline__ void round_twiddles(T* wv, const T* __restrict__ tw, u32 tid,
                                               const TileGeom& g) {
  constexpr int kRE = re_of(S), kE = el_of(S);
  using RD = Rounds<S, CB>;
  constexpr int r = RD::r(j), w = RD::w(j), u = RD::u(j);
  constexpr int SS = kE >> r;
  const u32 level = 1u << (g.a0 + u);
#pragma unroll
  for (int s = 0; s < SS; ++s) {
    // sub-block-and-group index of the run: the bits of vt above the gap, minus
    // the column bits (which do not select a twiddle)
    u32 node = level + (((g.tile_blk0 << u) + vt_high<w, TL, RD::kRE>(s, tid)) & (level - 1));
    if (w >= 6) node = __builtin_amdgcn_readfirstlane(node);  // uniform across the wave
    load_twiddles<r, CTW>(wv + (s << r), tw, node);
  }
}

// fmask (forward, bounded members of the Lazy family): stage mask of the pass, bit s = stage s
template <int S, int CB, int j, class A, bool FWD, bool LAST, class Hook = NoHook>
__device__ __forceinline__ void round_compute(u64* x, const TwT<A>* wv, const ModConst& m,
                                              const InvLast& il, u32 fmask = 0, const Hook& hook = Hook()) {
  constexpr int kRE = re_of(S), kE = el_of(S);
  constexpr int r = Rounds<S, CB>::r(j);
  constexpr int SS = kE >> r;
#pragma unroll
  for (int s = 0; s < SS; ++s) {
    if (FWD)
      fwd_subtree<r, A, Hook>(x + (s << r), wv + (s << r), m, fmask >> Rounds<S, CB>::u(j), hook);
    else  // (Lazy: entry bound and exit threshold of round j of this pass's chain)
      inv_subtree<r, A, LAST,
                  lazy_chain_entry(j, Rounds<S, CB>::NR, Rounds<S, CB>::R0, kRE, A::kLazy ? A::kLimit : kLazyLimit),
                  lazy_chain_thresh(j, Rounds<S, CB>::R0, kRE, A::kLazy ? A::kLimit : kLazyLimit), (S >= 6),
                  // (Fp64 family: reduce at the round's exit only where the run may not go on)
                  (A::kFp ? Rounds<S, CB>::fp_inv_reduce_after(j, A::kInvRun > 0 ? A::kInvRun : 1, LAST) : !LAST)>(
          x + (s << r), wv + (s << r), m, il);
  }
  if constexpr (FWD && A::kFp) {
    if (Rounds<S, CB>::fp_reduce_after(j, A::kFwdRun)) fp_bound_all<A, kE>(x, m);
  }
}

// LDS byte address of a slot.  lds_slot is linear over XOR and the fields of a
// tile index are disjoint bit ranges, so the address of element e of a thread is
// (address of its element 0) ^ (compile-time constant): one v_xor per access.
__device__ __forceinline__ u64& lds_at(u64* lds, u32 byte_addr) {
  return *reinterpret_cast<u64*>(reinterpret_cast<char*>(lds) + byte_addr);
}

template <int S, int CB, int TL, int j>
__device__ __forceinline__ void lds_load_round(u64* x, u64* lds, u32 tid) {
  constexpr int kRE = re_of(S), kE = el_of(S);
  constexpr int r = Rounds<S, CB>::r(j), w = Rounds<S, CB>::w(j);
  constexpr int SS = kE >> r;
  constexpr int kThreads = 1 << (TL - kRE);
#pragma unroll
  for (int s = 0; s < SS; ++s) {
    const u32 a0 = lds_slot<kRE>(tile_index<r, w>(s * kThreads + tid, 0)) << 3;
#pragma unroll
    for (int e = 0; e < (1 << r); ++e) x[(s << r) + e] = lds_at(lds, a0 ^ (lds_slot<kRE>((u32)e << w) << 3));
  }
}

template <int S, int CB, int TL, int j>
__device__ __forceinline__ void lds_store_round(const u64* x, u64* lds, u32 tid) {
  constexpr int kRE = re_of(S), kE = el_of(S);
  constexpr int r = Rounds<S, CB>::r(j), w = Rounds<S, CB>::w(j);
  constexpr int SS = kE >> r;
  constexpr int kThreads = 1 << (TL - kRE);
#pragma unroll
  for (int s = 0; s < SS; ++s) {
    const u32 a0 = lds_slot<kRE>(tile_index<r, w>(s * kThreads + tid, 0)) << 3;
#pragma unroll
    for (int e = 0; e < (1 << r); ++e) lds_at(lds, a0 ^ (lds_slot<kRE>((u32)e << w) << 3)) = x[(s << r) + e];
  }
}

// Hand-over of the tile from a round with finest gap 2^w to its neighbour.
// With w <= 6 a wave (64 consecutive virtual threads) owns one contiguous,
// aligned run of 512 tile elements in this round AND in every deeper round, so
// the next reader of those slots is the same wave: the LDS operations of one
// wave are processed in order and only the compiler has to be kept from
// reordering.  Transposes across waves (w > 6) need a workgroup barrier, and so
// does any hand-over involving a short round 0 (r < 3), whose threads own
// several runs spread over the tile (FULL == false).
template <int w, bool FULL>
__device__ __forceinline__ void handover() {
  if (w > 6 || !FULL) {
    __syncthreads();
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// Conversion of a tile's words on entry for a pass known to start the network (the one-kernel plans' tile_walk); REDUCE: the
// launch may carry a source map whose words are reduced on load (multi-plan form only).
template <int S, class A, bool REDUCE>
__device__ __forceinline__ void fetch_convert_first(u64* x, u32 flags, const ModConst& m) {
  constexpr int kE = el_of(S);
  if constexpr (REDUCE) {
    if (flags & (kReduceFirst | kRoundFirst)) entry_words<kE>(x, flags, m);
  }
#pragma unroll
  for (int i = 0; i < kE; ++i) x[i] = to_internal<A>(x[i], m);
}
// forward rounds J .. NR-1: LDS -> registers -> subtree -> same LDS slots.
// `pre` holds the twiddles of round J when Rounds::pre_fwd(J).
template <int S, int CB, int TL, int J, class A, bool CTW = false, class Hook = NoHook>
__device__ __forceinline__ void fwd_mid_rounds(u64* x, u64* lds, const TwT<A>* tw, u32 tid,
                                               const TileGeom& g, const ModConst& m,
                                               const InvLast& il, const TwT<A>* pre, u32 fmask,
                                               const Hook& hook = Hook()) {
  constexpr int kRE = re_of(S), kE = el_of(S);
  using RD = Rounds<S, CB>;
  if constexpr (J < RD::NR) {
    TwT<A> wv[kE], wn[kE];
    const TwT<A>* w = pre;
    if constexpr (!RD::pre_fwd(J)) {
      round_twiddles<S, CB, TL, J, CTW>(wv, tw, tid, g);
      w = wv;
    }
    lds_load_round<S, CB, TL, J>(x, lds, tid);
    if constexpr (RD::pre_fwd(J + 1)) round_twiddles<S, CB, TL, J + 1, CTW>(wn, tw, tid, g);
    if constexpr (J == RD::NR - 1 && RD::r(J) == kRE)  // (one subtree per thread: the hook runs once)
      round_compute<S, CB, J, A, true, false, Hook>(x, w, m, il, fmask, hook);
    else
      round_compute<S, CB, J, A, true, false>(x, w, m, il, fmask);
    lds_store_round<S, CB, TL, J>(x, lds, tid);
    handover<RD::w(J), RD::r(J) == kRE>();
    HX_STAMP(3 + J);
    fwd_mid_rounds<S, CB, TL, J + 1, A, CTW, Hook>(x, lds, tw, tid, g, m, il, wn, fmask, hook);
  }
}

// inverse rounds J .. 1 (deepest first).  `pre` holds the twiddles of round J when
// J == NR-1 or Rounds::pre_inv(J); `pre0` receives those of round 0 when
// Rounds::pre_inv(0).
// RAW (tile_walk): the tile sits in LDS as it came from memory; the words of the first round
// executed are brought to the internal form (and reduced, REDUCE / flags: see fetch_convert_first)
// right after their LDS load.
template <int S, int CB, int TL, int J, class A, bool CTW = false, bool RAW = false, bool REDUCE = false,
          class Hook = NoHook>
__device__ __forceinline__ void inv_mid_rounds(u64* x, u64* lds, const TwT<A>* tw, u32 tid,
                                               const TileGeom& g, const ModConst& m,
                                               const InvLast& il, const TwT<A>* pre,
                                               TwT<A>* pre0, u32 flags = 0, const Hook& hook = Hook()) {
  constexpr int kRE = re_of(S), kE = el_of(S);
  using RD = Rounds<S, CB>;
  if constexpr (J >= 1) {
    TwT<A> wv[kE], wn[kE];
    const TwT<A>* w = pre;
    if constexpr (J != RD::NR - 1 && !RD::pre_inv(J)) {
      round_twiddles<S, CB, TL, J, CTW>(wv, tw, tid, g);
      w = wv;
    }
    lds_load_round<S, CB, TL, J>(x, lds, tid);
    if constexpr (RD::pre_inv(J - 1)) round_twiddles<S, CB, TL, J - 1, CTW>(J == 1 ? pre0 : wn, tw, tid, g);
    if constexpr (RAW) fetch_convert_first<S, A, REDUCE>(x, flags, m);
    if constexpr (J == 1) hook();
    round_compute<S, CB, J, A, false, false>(x, w, m, il);
    lds_store_round<S, CB, TL, J>(x, lds, tid);
    // the next (shallower) round J-1 regroups across waves iff its gap exceeds a wave
    handover<RD::w(J - 1), RD::r(J - 1) == kRE>();
    HX_STAMP(3 + (RD::NR - 1 - J));
    inv_mid_rounds<S, CB, TL, J - 1, A, CTW, false, false, Hook>(x, lds, tw, tid, g, m, il, wn, pre0, 0, hook);
  }
}

template <int S, int CB, int TL>
__device__ __forceinline__ TileGeom make_geom(u32 tile, u32 log_n) {
  TileGeom g;
  if (CB == 0) {
    g.base = (u64)tile << TL;
    g.log_row = 0;
    g.a0 = log_n - S;
    g.tile_blk0 = (u32)((g.base & ((1ull << log_n) - 1)) >> S);
  } else {  // all 2^S rows x 2^CB columns of one polynomial, N >= 2^TL
    const u32 tpl = log_n - TL;  // log2(tiles per polynomial)
    const u64 poly = (u64)tile >> tpl;
    const u32 t = tile & ((1u << tpl) - 1);
    g.base = (poly << log_n) + ((u64)t << CB);
    g.log_row = log_n - S;
    g.a0 = 0;
    g.tile_blk0 = 0;
  }
  return g;
}

// GUARD: the batch may end inside the tile (only possible for CB == 0 and a batch
// smaller than / not a multiple of the tile); otherwise every access is in range.
// CONVERT == false: only the loads (fetch_convert follows once the plan constants are there).
// ENTRY: the launch may carry a source map (the multi-plan kernels): words reduced / rounded on load.
template <bool ROUND0, int S, int CB, int TL, bool GUARD, class A, int LDK, bool CONVERT = true,
          bool ENTRY = false>
__device__ __forceinline__ void fetch_tile(u64* x, const u64* in, u32 tid, const TileGeom& g,
                                           u64 total, u32 flags, const ModConst& m) {
  constexpr int kRE = re_of(S), kE = el_of(S);
#pragma unroll
  for (int i = 0; i < kE; ++i) {
    const u32 p0 = xfer_p0<ROUND0, S, CB, TL>(tid, i);
    const u32 dp = xfer_dp<ROUND0, S, CB>(i);
    const u64* src = in + tile_uniform_offset<CB>(g, dp);
    if constexpr (LDK == kRaw) {  // (the walk: never ragged)
      x[i] = raw_load_b64(src, tile_byte_offset<CB>(g, p0));
      continue;
    }
#ifdef HEXL_AMD_EXP_NOLOAD  // developer experiment (timing only, wrong results): no global loads
    x[i] = (u64)(p0 + dp) * 0x9E3779B97F4A7C15ull >> 12;
    HX_OPAQUE(x[i]);
#else
    if (GUARD)
      x[i] = (g.base + p0 + dp < total) ? load_global<LDK>(src, tile_byte_offset<CB>(g, p0)) : 0;
    else
      x[i] = load_global<LDK>(src, tile_byte_offset<CB>(g, p0));
#endif
  }
  if constexpr (!CONVERT) return;
  if (flags & kFirstPass) {
    if constexpr (ENTRY) {
      if (flags & (kReduceFirst | kRoundFirst)) entry_words<kE>(x, flags, m);
    }
#pragma unroll
    for (int i = 0; i < kE; ++i) x[i] = to_internal<A>(x[i], m);
  }
}
template <int S, class A, bool ENTRY = false>
__device__ __forceinline__ void fetch_convert(u64* x, u32 flags, const ModConst& m) {
  constexpr int kE = el_of(S);
  if (flags & kFirstPass) {
    if constexpr (ENTRY) {
      if (flags & (kReduceFirst | kRoundFirst)) entry_words<kE>(x, flags, m);
    }
#pragma unroll
    for (int i = 0; i < kE; ++i) x[i] = to_internal<A>(x[i], m);
  }
}

template <bool ROUND0, int S, int CB, int TL, bool GUARD, int STK>
__device__ __forceinline__ void store_elem(u64* out, u32 tid, int i, u64 v,
                                           const TileGeom& g, u64 total) {
  const u32 p0 = xfer_p0<ROUND0, S, CB, TL>(tid, i);
  const u32 dp = xfer_dp<ROUND0, S, CB>(i);
#ifdef HEXL_AMD_EXP_NOSTORE  // developer experiment (timing only): one store in 2^20 happens
  if ((v & 0xfffff) != 0x12345) return;
#endif
  if constexpr (STK == kRaw) {
    raw_store_b64(out + tile_uniform_offset<CB>(g, dp), tile_byte_offset<CB>(g, p0), v);
    return;
  }
  if (!GUARD || g.base + p0 + dp < total)
    store_global<STK>(out + tile_uniform_offset<CB>(g, dp), tile_byte_offset<CB>(g, p0), v);
}

// End of a forward tile pass: the 512-element run this wave owns after the last round goes
// from LDS to global memory, 64 elements per access, with the final reduction fused.
// FIN: 0 = more passes follow, 1 = lazy output range, 2 = canonical.
template <int FIN, int S, int CB, int TL, bool GUARD, class A, int STK>
__device__ __forceinline__ void fwd_copy_out(u64* lds, u64* out, u32 tid, const TileGeom& g,
                                             u64 total, const ModConst& m) {
  constexpr int kE = el_of(S);
  u64 v[kE];
  u32 a0 = lds_slot<re_of(S)>(xfer_p0<false, S, CB, TL>(tid, 0)) << 3;
  HX_OPAQUE(a0);  // one v_xor per access (else: the swizzle redone and shifted per element)
#pragma unroll
  for (int i = 0; i < kE; ++i) v[i] = lds_at(lds, a0 ^ (lds_slot<re_of(S)>(xfer_dp<false, S, CB>(i)) << 3));
#pragma unroll
  for (int i = 0; i < kE; ++i) {
    if (FIN)
      v[i] = fwd_finish<A>(v[i], m, FIN == 2);
    else
      v[i] = fp_bound<A>(v[i], m);  // Fp64: every pass hands over fully reduced values
    store_elem<false, S, CB, TL, GUARD, STK>(out, tid, i, v[i], g, total);
  }
}

// End of the last forward transform of a KeySwitch (multi-plan launches with a KsEpilogue): the
// run's words t (lazy, < 4q) are not stored; result[p][l] += ((prod[i][tc][l] + 4q - t) mod q) * s
// mod q -- ks_finish_kernel's arithmetic, word for word (key-switch-internal.cpp:180-196).
struct NoEpilogue {};
template <int S, int CB, int TL, class A>
__device__ __forceinline__ void fwd_copy_out_finish(u64* lds, u32 tid, const TileGeom& g, u32 log_n,
                                                    const ModConst& m, const KsEpilogue& ep) {
  static_assert(CB == 0, "bottom tiles only");
  constexpr int kE = el_of(S);
  const u32 poly = __builtin_amdgcn_readfirstlane((u32)(g.base >> log_n));
  const u32 i = poly % ep.decomp, tc = poly / ep.decomp;
  const u64 within = g.base & ((1ull << log_n) - 1);  // the tile's first word inside its polynomial
  const u64* pp = ep.prod + ((((u64)i * ep.tc + tc) << log_n) + within);
  u64* dd = ep.result + (((u64)poly << log_n) + within);
  const u64 q = m.q, s = ep.s[i], sp = ep.sp[i];
  u64 v[kE], pv[kE], dv[kE];
  u32 a0 = lds_slot<re_of(S)>(xfer_p0<false, S, CB, TL>(tid, 0)) << 3;
  HX_OPAQUE(a0);
#pragma unroll
  for (int k = 0; k < kE; ++k) v[k] = lds_at(lds, a0 ^ (lds_slot<re_of(S)>(xfer_dp<false, S, CB>(k)) << 3));
  const u32 off0 = xfer_p0<false, S, CB, TL>(tid, 0) << 3;
#pragma unroll
  for (int k = 0; k < kE; ++k) {
    pv[k] = load_global<kStream>(pp + xfer_dp<false, S, CB>(k), off0);
    dv[k] = load_global<kStream>(dd + xfer_dp<false, S, CB>(k), off0);
  }
#pragma unroll
  for (int k = 0; k < kE; ++k) {
    const u64 t = fwd_finish<A>(v[k], m, false);
    u64 x = pv[k] + (q << 2) - t;  // < 8q
    x = csub(x, q << 2);
    x = csub(x, q << 1);
    x = csub(x, q);
    const u64 r = csub(mul_lazy(x, s, sp, q), q);
    store_global<kStream>(dd + xfer_dp<false, S, CB>(k), off0, csub(dv[k] + r, q));
  }
}

// One workgroup per tile; the hardware refills a CU's slots as workgroups retire.
// (Three persistent variants were built and measured in round 1 -- register
// prefetch at 8 and at 6 waves per SIMD, LDS-direct DMA prefetch -- all slower, see
// the note after the kernel and DESIGN.md.)
//
// FWD:  global --(round 0)--> LDS --(rounds 1..)--> LDS --> coalesced store
// INV:  coalesced load --> LDS --(rounds NR-1..1)--> LDS --(round 0)--> global
// (A one-off sleep de-phasing the first generation of workgroups was measured
// too: no effect -- the kernel is bound by VALU issue, not by phase alignment.)

// Occupancy target: 8 waves per SIMD (<= 64 VGPRs) for the shapes large transforms
// use; the short bottom passes of small N (several sub-runs per thread in round
// 0) would spill under that cap and get 6 (<= 80 VGPRs).
template <int S, int CB>
constexpr int min_waves() { return S >= 14 ? 4 : (S >= 10 || CB > 0) ? 8 : 6; }

// LAST (inverse only): the pass contains the root stage of the transform.
// tile_body: the work of one workgroup on tile `bid`; `lds` = 2^TL words of LDS.
// LDK / STK: access kinds of the global loads and stores (see ld_global).  The data
// pointers are not __restrict__: transforms run in place (out == in).
template <bool FWD, int S, int CB, int TL, bool GUARD, class A, bool LAST, int LDK, int STK,
          bool DATA_FIRST = false, bool CTW = false, class Epi = NoEpilogue>
__device__ __forceinline__ void tile_body(u64* lds, u64* out, const u64* in,
                                          const ulonglong2* __restrict__ tw_raw, const ModConst& m,
                                          u32 log_n, u32 flags, u64 total, const InvLast& il,
                                          u32 bid, const Epi& epi = Epi()) {
  constexpr int kRE = re_of(S), kE = el_of(S);
  using RD = Rounds<S, CB>;
  constexpr int NR = RD::NR;
  const TwT<A>* __restrict__ tw = reinterpret_cast<const TwT<A>*>(tw_raw);
  const u32 finish = flags & kFinishMask;
  const bool first = (flags & kFirstPass) != 0;
  const u32 tid = threadIdx.x;
  // A new wave first gets its tile's loads out, at raised priority, before it
  // competes with seven computing waves for VALU issue slots: its address
  // arithmetic would otherwise trickle through and delay the loads by ~1 us
  // (forward tile pass 1.03 -> 0.99 ms; neutral for the inverse).
  __builtin_amdgcn_s_setprio(3);
  const TileGeom g = make_geom<S, CB, TL>(bid, log_n);
  u64 x[kE];
  HX_STAMP(0);

  if (FWD) {
    TwT<A> wn[kE];  // twiddles of round 1 when requested ahead
    {  // round 0 straight from global memory; its twiddles are requested first
      TwT<A> wv[kE];
      if constexpr (DATA_FIRST) {  // see strided_body
        fetch_tile<true, S, CB, TL, GUARD, A, LDK, false>(x, in, tid, g, total, flags, m);
        round_twiddles<S, CB, TL, 0, CTW>(wv, tw, tid, g);
        fetch_convert<S, A, (CTW && FWD)>(x, flags, m);
      } else {
        round_twiddles<S, CB, TL, 0, CTW>(wv, tw, tid, g);
        fetch_tile<true, S, CB, TL, GUARD, A, LDK, true, (CTW && FWD)>(x, in, tid, g, total, flags, m);  // round-0 set
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (RD::pre_fwd(1)) round_twiddles<S, CB, TL, 1, CTW>(wn, tw, tid, g);
      HX_PROFILE_WAIT_VMEM();
      HX_STAMP(1);
      round_compute<S, CB, 0, A, true, false>(x, wv, m, il, flags >> kStageMaskShift);
      HX_STAMP(2);
      lds_store_round<S, CB, TL, 0>(x, lds, tid);
      handover<RD::w(0), RD::r(0) == kRE>();
      HX_STAMP(3);
    }
    fwd_mid_rounds<S, CB, TL, 1, A, CTW>(x, lds, tw, tid, g, m, il, wn, flags >> kStageMaskShift);
    // copy-out of the run this wave owns after the last round (w = CB <= 6):
    // 512 tile-contiguous elements, 64 per access; final reduction fused
    // (the finish kind is uniform: three straight-line copies of the copy-out behind scalar
    // branches instead of a branch and a select per element, with all LDS reads up front)
    if constexpr (!std::is_same<Epi, NoEpilogue>::value)
      fwd_copy_out_finish<S, CB, TL, A>(lds, tid, g, log_n, m, epi);
    else if (finish == 2)
      fwd_copy_out<2, S, CB, TL, GUARD, A, STK>(lds, out, tid, g, total, m);
    else if (finish)
      fwd_copy_out<1, S, CB, TL, GUARD, A, STK>(lds, out, tid, g, total, m);
    else
      fwd_copy_out<0, S, CB, TL, GUARD, A, STK>(lds, out, tid, g, total, m);
    HX_STAMP(8);
    HX_PROFILE_WAIT_VMEM();
    HX_STAMP(9);
  } else {
    // copy-in of the run this wave owns in the deepest round; the twiddles of that
    // round are requested first
    TwT<A> wtop[kE], w0[kE];
    if constexpr (DATA_FIRST) {
      fetch_tile<false, S, CB, TL, GUARD, A, LDK, false>(x, in, tid, g, total, flags, m);
      if constexpr (NR > 1) round_twiddles<S, CB, TL, NR - 1, CTW>(wtop, tw, tid, g);
      fetch_convert<S, A, (CTW && FWD)>(x, flags, m);
    } else {
      if constexpr (NR > 1) round_twiddles<S, CB, TL, NR - 1, CTW>(wtop, tw, tid, g);
      fetch_tile<false, S, CB, TL, GUARD, A, LDK, true, (CTW && FWD)>(x, in, tid, g, total, flags, m);
    }
    __builtin_amdgcn_s_setprio(0);
    HX_PROFILE_WAIT_VMEM();
    HX_STAMP(1);
    {
      const u32 a0 = lds_slot<re_of(S)>(xfer_p0<false, S, CB, TL>(tid, 0)) << 3;
#pragma unroll
      for (int i = 0; i < kE; ++i) lds_at(lds, a0 ^ (lds_slot<re_of(S)>(xfer_dp<false, S, CB>(i)) << 3)) = x[i];
    }
    handover<RD::w(NR - 1), RD::r(NR - 1) == kRE>();
    HX_STAMP(2);
    inv_mid_rounds<S, CB, TL, NR - 1, A, CTW>(x, lds, tw, tid, g, m, il, wtop, w0);
    {
      if constexpr (!RD::pre_inv(0)) round_twiddles<S, CB, TL, 0, CTW>(w0, tw, tid, g);
      lds_load_round<S, CB, TL, 0>(x, lds, tid);
      round_compute<S, CB, 0, A, false, LAST>(x, w0, m, il);
      HX_STAMP(7);
      if (finish == 2) {
#pragma unroll
        for (int i = 0; i < kE; ++i) x[i] = inv_finish<A>(x[i], m, true);
      } else if (finish) {
#pragma unroll
        for (int i = 0; i < kE; ++i) x[i] = inv_finish<A>(x[i], m, false);
      }
#pragma unroll
      for (int i = 0; i < kE; ++i) store_elem<true, S, CB, TL, GUARD, STK>(out, tid, i, x[i], g, total);
      HX_STAMP(8);
      HX_PROFILE_WAIT_VMEM();
      HX_STAMP(9);
    }
  }
}

template <bool FWD, int S, int CB, int TL, bool GUARD, class A, bool LAST>
__global__ void __launch_bounds__(1 << (TL - re_of(S)), (min_waves<S, CB>()))
tile_pass(u64* out, const u64* in, const ulonglong2* __restrict__ tw, ModConst m, u32 log_n,
          u32 flags, u64 total, InvLast il) {
#if defined(HEXL_AMD_EXP_PAD13)  // developer experiment: the 64 KiB tile at ONE workgroup per CU
  __shared__ u64 lds[TL == 13 ? (1 << 14) : (1 << TL)];
#elif defined(HEXL_AMD_EXP_PAD12)  // ... the 32 KiB tile padded to 2^PAD12 words (fewer workgroups per CU)
  __shared__ u64 lds[TL == 12 ? (1 << HEXL_AMD_EXP_PAD12) : (1 << TL)];
#else
  __shared__ u64 lds[1 << TL];
#endif
  // (The XCD-aware block order of strided_pass was measured here too: 2-7 % slower.  So was
  // a workgroup that walks 2 or 4 consecutive tiles instead of one -- a k-th of the
  // dispatches, no wait for the previous tile's store acknowledgements: forward +2 / +8 %,
  // inverse +65 / +53 % slower, round 3.  And a tile-index-major order -- consecutive workgroups
  // take the same tile of consecutive polynomials, everything in flight sharing one set of
  // per-lane twiddles (31.5 KiB: L1-resident): forward +1 %, inverse +8 % slower, round 3.)
  // streamed loads and stores
  // (an inverse pass that ends the transform -- the one-kernel plans, N <= 2^14 -- streams too:
  // 2-7 % at 1 GiB batches, round 3; its stores through an LDS copy-out, tile-contiguous per
  // wave like the forward's: slower; its data loads ahead of its twiddle loads: no effect)
  // (round 6, same-box A/B at N = 65536 x 4096, profiles/r6_inverse_access_kind_ab.txt: an inverse
  // pass followed by a strided pass streams too -- tile pass -2.4 % (Lazy) / -3 % (Fp64) / -4 %
  // (Small), the strided pass behind it +0 ... +2 %, the step -0.7 ... -0.8 %; round 3 had measured
  // it neutral)
  constexpr int kKind = kStream;
  tile_body<FWD, S, CB, TL, GUARD, A, LAST, kKind, kKind>(lds, out, in, tw, m, log_n, flags, total, il,
                                                          blockIdx.x);
}

// Bottom pass over polynomials of several moduli (see strided_pass_multi); N >= 2^TL, so
// a tile lies in one polynomial and no tile is ragged.
// EPI (forward): the pass ends a KeySwitch -- its output is folded into the result (KsEpilogue).
template <bool EPI>
struct EpilogueArg {
  typedef NoEpilogue T;
};
template <>
struct EpilogueArg<true> {
  typedef KsEpilogue T;
};
template <bool FWD, int S, int TL, class A, bool LAST, bool EPI = false>
__global__ void __launch_bounds__(1 << (TL - re_of(S)), (min_waves<S, 0>()))
tile_pass_multi(u64* out, const u64* in, MultiCtx mc, u32 log_n, u32 flags, u64 total,
                typename EpilogueArg<EPI>::T epi) {
  __shared__ u64 lds[1 << TL];
  const u32 poly = (u32)(((u64)blockIdx.x << TL) >> log_n);
  u32 k;
  const PlanDev* __restrict__ pd = multi_plan(mc, poly, policy_id<A>(), k);
  if (!pd) return;
  ModConst m = pd->mod;
  m.rnd_qk = mc.map.rnd_qk;  // (rounding on load: see kRoundFirst)
  m.rnd_barrett = mc.map.rnd_barrett;
  m.rnd_half = mc.map.rnd_half;
  const InvLast il = pd->il;
  in = multi_source(mc, poly, log_n, in, flags);
  // (data-first only for the small tiles: with 16 elements per thread it costs registers)
  tile_body<FWD, S, 0, TL, false, A, LAST, FWD ? kStream : kPlain, FWD ? kStream : kPlain, (S <= 12),
            true, typename EpilogueArg<EPI>::T>(
      lds, out, in, FWD ? mc.tw_fwd[k] : mc.tw_inv[k], m, log_n, flags, total, il, blockIdx.x, epi);
}

// ---------------------------------------------------------------------------
// tile_walk: the 128 KiB tile as a persistent workgroup (round 6)
// ---------------------------------------------------------------------------
// The 14-stage tile is ONE workgroup per CU (its LDS), so nothing on the CU overlaps a
// workgroup's load phase, its store drain and the dispatch of its successor: the counters show
// VALU busy 0.64 where the two-workgroups-per-CU 13-stage tile reaches 0.87, and the 13-stage
// tile forced to one workgroup per CU loses exactly that (profiles/r6_pmc_onekernel.md).  Two
// co-resident 14-stage tiles do not fit (2 x 128 KiB of data against 160 KiB of LDS; held in
// registers instead, 64 of a thread's 128 VGPRs at 2 x 8 waves -- half the waves, measured on the
// 12-stage tile as a third of the gain).  So the workgroup stays and walks the tiles
// blockIdx.x, blockIdx.x + gridDim.x, ... itself, with the NEXT tile's loads issued before THIS
// tile's stores: vector memory operations complete in order, so the wait for the loads does not
// include the stores' acknowledgements, which drain while the next tile computes.
//   forward: ... last round -> LDS | loads(next round-0 set) | copy-out: LDS -> finish -> stores
//   inverse: ... round 0: LDS -> registers | loads(next run) | round 0 arithmetic -> stores
// (the inverse hides its loads behind two stages of arithmetic, the forward behind the copy-out).
// Only the one-kernel plans: a tile is a whole polynomial (S == TL == log2 N, CB = 0, nothing
// ragged), the pass starts and ends the network.
__device__ __forceinline__ u32 walk_next(const MultiCtx* mc, u32 t, u32 ntiles, int want, u32& k) {
  if (mc == nullptr) return t;
  for (; t < ntiles; t += gridDim.x) {
    if (multi_plan(*mc, t, want, k)) break;
  }
  return t;
}

// (straight into the words of the twiddle array, two 8-byte loads per (W, W') pair: a 16-byte
// load into a raw 128-bit value and a conversion behind the wait cost a second array and 100 bytes
// of scratch)
// The run this wave owns in the deepest round, from registers to its LDS slots (inverse).
template <int S, int TL>
__device__ __forceinline__ void walk_park(const u64* x, u64* lds, u32 tid) {
  constexpr int kE = el_of(S);
  const u32 a0 = lds_slot<re_of(S)>(xfer_p0<false, S, 0, TL>(tid, 0)) << 3;
#pragma unroll
  for (int i = 0; i < kE; ++i) lds_at(lds, a0 ^ (lds_slot<re_of(S)>(xfer_dp<false, S, 0>(i)) << 3)) = x[i];
}

template <bool FWD, int S, class A, bool LAST, bool MULTI>
__device__ __forceinline__ void tile_walk(u64* lds, u64* out, const u64* in_arg,
                                          const ulonglong2* __restrict__ tw_arg, const ModConst& m_arg,
                                          u32 log_n, u32 flags_arg, u64 total, const InvLast& il_arg,
                                          const MultiCtx* mc) {
  constexpr int TL = S, CB = 0;
  constexpr int kRE = re_of(S), kE = el_of(S);
  using RD = Rounds<S, CB>;
  constexpr int NR = RD::NR;
  constexpr bool CTW = MULTI;
  // access kinds as in tile_pass / tile_pass_multi
  constexpr int kKind = MULTI ? (FWD ? kStream : kPlain) : ((FWD || LAST) ? kStream : kPlain);
  u32 tid = threadIdx.x;
  const u32 ntiles = (u32)(total >> TL);
  u32 k = 0;
  u32 tile = walk_next(MULTI ? mc : nullptr, blockIdx.x, ntiles, policy_id<A>(), k);
  if (tile >= ntiles) return;
  ModConst m = m_arg;
  InvLast il = il_arg;
  const TwT<A>* __restrict__ tw = reinterpret_cast<const TwT<A>*>(tw_arg);
  u32 flags = flags_arg;
  const u64* in = in_arg;
  if constexpr (MULTI) in = multi_source(*mc, tile, log_n, in_arg, flags);
  TileGeom g = make_geom<S, CB, TL>(tile, log_n);
  u64 x[kE];
  __builtin_amdgcn_s_setprio(3);
  if constexpr (FWD) {
    // (raw: nothing the compiler tracks is in flight at the head of the loop on either path)
    fetch_tile<true, S, CB, TL, false, A, kRaw, false>(x, in, tid, g, total, flags, m);
    raw_wait<0, kE>(x);
  } else {
    fetch_tile<false, S, CB, TL, false, A, kKind, false>(x, in, tid, g, total, flags, m);
  }
  __builtin_amdgcn_s_setprio(0);
  if constexpr (!FWD) walk_park<S, TL>(x, lds, tid);
  TileGeom gp = g;  // inverse: the tile whose results wait in x for their stores
  bool first_iteration = true;
  for (;;) {
    // (the thread index is made opaque per phase: left visible as a loop invariant, every LDS and
    // global address of every round is hoisted out of the loop and kept in registers across it --
    // the inverse spilled 340 bytes and parked the prefetched tile in scratch)
    asm volatile("" : "+v"(tid));
    if constexpr (MULTI) {
      const PlanDev* __restrict__ pd = mc->p[k];
      m = pd->mod;
      m.rnd_qk = mc->map.rnd_qk;  // (rounding on load: see kRoundFirst)
      m.rnd_barrett = mc->map.rnd_barrett;
      m.rnd_half = mc->map.rnd_half;
      il = pd->il;
      tw = reinterpret_cast<const TwT<A>*>(FWD ? mc->tw_fwd[k] : mc->tw_inv[k]);
    }
    // the tile after this one (looked up where its loads are issued: nothing of it is live before)
    u32 kn = 0, nt = 0, nflags = flags_arg;
    bool more = false;
    const u64* nin = in_arg;
    TileGeom ng = g;
    auto look_ahead = [&]() {
      nt = walk_next(MULTI ? mc : nullptr, tile + gridDim.x, ntiles, policy_id<A>(), kn);
      more = nt < ntiles;
      if (more) {
        ng = make_geom<S, CB, TL>(nt, log_n);
        if constexpr (MULTI) nin = multi_source(*mc, nt, log_n, in_arg, nflags);
      }
    };
    const u32 finish = flags & kFinishMask;
    if constexpr (FWD) {
      TwT<A> wn[kE];
      {
        TwT<A> wv[kE];
        round_twiddles<S, CB, TL, 0, CTW>(wv, tw, tid, g);
        fetch_convert_first<S, A, MULTI>(x, flags, m);
        if constexpr (RD::pre_fwd(1)) round_twiddles<S, CB, TL, 1, CTW>(wn, tw, tid, g);
        round_compute<S, CB, 0, A, true, false>(x, wv, m, il, flags >> kStageMaskShift);
        // (the previous tile's copy-out has read every slot: its LDS reads are behind this barrier)
        if (!first_iteration) __syncthreads();
        lds_store_round<S, CB, TL, 0>(x, lds, tid);
        handover<RD::w(0), RD::r(0) == kRE>();
      }
      // The next tile's round-0 set is requested in front of the LAST stage of the last round
      // (the twiddles of the round's first three stages are dead by then: 32 registers are
      // free), ahead of that stage, the round's LDS store, the copy-out and this tile's stores.
      // (Always defined: see the inverse.)
      u64 y[kE];
#pragma unroll
      for (int i = 0; i < kE; ++i) y[i] = 0;
      // (UNCONDITIONAL loads -- the last tile of a workgroup requests itself once more and drops
      // it: issued under `if (more)`, the join of the two paths made every later wait for the
      // round's own twiddle loads a wait for the prefetch as well -- vmcnt(5) ... vmcnt(0) where the
      // prefetching path needs vmcnt(21) ... vmcnt(16) -- and the walk gained nothing)
      auto prefetch = [&]() {
        asm volatile("" : "+v"(tid));
        look_ahead();
        const u64* pin = more ? nin : in;
        const TileGeom pg = more ? ng : g;
        // the round's own (tracked) twiddle loads are waited for here, where they have long
        // arrived: behind the raw loads a wait for them would be a wait for the prefetch
        __builtin_amdgcn_sched_barrier(0);  // (... here: the scheduler hoisted the wait to the loads' issue)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) expcnt(7) lgkmcnt(15)
        __builtin_amdgcn_s_setprio(3);
        fetch_tile<true, S, CB, TL, false, A, kRaw, false>(y, pin, tid, pg, total, nflags, m);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      };
      fwd_mid_rounds<S, CB, TL, 1, A, CTW>(x, lds, tw, tid, g, m, il, wn, flags >> kStageMaskShift,
                                           prefetch);
      asm volatile("" : "+v"(tid));
      // (raw stores, exactly kE of them behind the prefetch: the wait below counts on it)
      if (finish == 2)
        fwd_copy_out<2, S, CB, TL, false, A, kRaw>(lds, out, tid, g, total, m);
      else if (finish)
        fwd_copy_out<1, S, CB, TL, false, A, kRaw>(lds, out, tid, g, total, m);
      else
        fwd_copy_out<0, S, CB, TL, false, A, kRaw>(lds, out, tid, g, total, m);
      raw_wait<kE, kE>(y);  // the prefetch has landed; this tile's stores are still draining
#pragma unroll
      for (int i = 0; i < kE; ++i) x[i] = y[i];
    } else {
      // (the tile is in LDS already, as it came from memory: stored by the prologue / by the
      // previous iteration; this wave reads back its own run, so no barrier)
      // The PREVIOUS tile's results are still in x: they are stored here, behind the request for
      // this tile's per-lane twiddles -- vector memory operations complete in order, and a load
      // issued behind the stores would wait for their acknowledgements.
      // (Tracked loads and stores.  The compiler moves the conditional stores in front of the
      // twiddle loads and merges the two paths conservatively, so the first twiddle wait still
      // includes the stores' acknowledgements; the raw form of raw_load_b64 -- twiddles, stores and the
      // wait by hand -- was built for this spot too and cost the mid rounds their registers: 100 bytes
      // of scratch, the prefetch parked in scratch again.  Not adopted.)
      TwT<A> wtop[kE], w0[kE];
      if constexpr (NR > 1) round_twiddles<S, CB, TL, NR - 1, CTW>(wtop, tw, tid, g);
      if (!first_iteration) {
#pragma unroll
        for (int i = 0; i < kE; ++i) store_elem<true, S, CB, TL, false, kKind>(out, tid, i, x[i], gp, total);
      }
      handover<RD::w(NR - 1), RD::r(NR - 1) == kRE>();
      // The next tile's run is requested in round 1, between its LDS load and its arithmetic
      // (scalar twiddles there: 32 registers are free), ahead of rounds 1 and 0 -- six stages;
      // it is parked in LDS, as it comes, once every wave holds its round-0 set.  (Always
      // defined, loaded under one condition and parked under the same one: defined on one path
      // only, the register allocator kept it in scratch.)
      u64 y[kE];
#pragma unroll
      for (int i = 0; i < kE; ++i) y[i] = 0;
      auto prefetch = [&]() {
        asm volatile("" : "+v"(tid));
        look_ahead();
        if (more) {
          __builtin_amdgcn_s_setprio(3);
          fetch_tile<false, S, CB, TL, false, A, kKind, false>(y, nin, tid, ng, total, nflags, m);
          __builtin_amdgcn_s_setprio(0);
        }
      };
      // (reduce-on-load is compiled into the forward walk only: launch_bottom keeps inverse launches
      // with a source map on tile_pass_multi)
      inv_mid_rounds<S, CB, TL, NR - 1, A, CTW, true, false>(x, lds, tw, tid, g, m, il, wtop, w0, flags,
                                                             prefetch);
      asm volatile("" : "+v"(tid));
      if constexpr (!RD::pre_inv(0)) round_twiddles<S, CB, TL, 0, CTW>(w0, tw, tid, g);
      lds_load_round<S, CB, TL, 0>(x, lds, tid);
      if (more) {
        // every wave holds its round-0 set (other waves' runs among it): the tile's LDS is free
        __syncthreads();
        walk_park<S, TL>(y, lds, tid);
      }
      round_compute<S, CB, 0, A, false, LAST>(x, w0, m, il);
      if (finish == 2) {
#pragma unroll
        for (int i = 0; i < kE; ++i) x[i] = inv_finish<A>(x[i], m, true);
      } else if (finish) {
#pragma unroll
        for (int i = 0; i < kE; ++i) x[i] = inv_finish<A>(x[i], m, false);
      }
      gp = g;
      if (!more) {  // the last tile of this workgroup
#pragma unroll
        for (int i = 0; i < kE; ++i) store_elem<true, S, CB, TL, false, kKind>(out, tid, i, x[i], gp, total);
      }
    }
    if (!more) break;
    tile = nt;
    g = ng;
    flags = nflags;
    k = kn;
    first_iteration = false;
  }
}

template <bool FWD, int S, class A, bool LAST>
__global__ void __launch_bounds__(1 << (S - re_of(S)), (min_waves<S, 0>()))
tile_walk_pass(u64* out, const u64* in, const ulonglong2* __restrict__ tw, ModConst m, u32 log_n,
               u32 flags, u64 total, InvLast il) {
  __shared__ u64 lds[1 << S];
  tile_walk<FWD, S, A, LAST, false>(lds, out, in, tw, m, log_n, flags, total, il, nullptr);
}

template <bool FWD, int S, class A, bool LAST>
__global__ void __launch_bounds__(1 << (S - re_of(S)), (min_waves<S, 0>()))
tile_walk_pass_multi(u64* out, const u64* in, MultiCtx mc, u32 log_n, u32 flags, u64 total) {
  __shared__ u64 lds[1 << S];
  const ModConst m{};
  const InvLast il{};
  tile_walk<FWD, S, A, LAST, true>(lds, out, in, nullptr, m, log_n, flags, total, il, &mc);
}


// ---------------------------------------------------------------------------
// Host-side planning and launch
// ---------------------------------------------------------------------------

// Process-wide tuning state (hexl_amd_set_tuning; include/hexl_amd.h documents the keys).  The
// library reads no environment variable: every knob has a compiled-in default and changes only
// through that call.  Results never depend on it.
constexpr u64 kTile14MinBatch = 96;
struct Tuning {
  std::atomic<u32> fp64{1}, h60{1}, tile13{2}, bigtile{1}, lazy_family{1}, fp64_long{1}, walk14{1};
};
Tuning& tuning();  // one per process: defined by the dispatch unit
#if HX_TU_DISPATCH
Tuning& tuning() {
  static Tuning t;
  return t;
}
int set_tuning(const char* key, u64 value) {
  Tuning& t = tuning();
  if (strcmp(key, "fp64") == 0 && value <= 2) t.fp64 = (u32)value;
  else if (strcmp(key, "tile13") == 0 && value <= 2) t.tile13 = (u32)value;
  else if (strcmp(key, "h60") == 0 && value <= 1) t.h60 = (u32)value;
  else if (strcmp(key, "bigtile") == 0 && value <= 1) t.bigtile = (u32)value;
  else if (strcmp(key, "lazy_family") == 0 && value <= 1) t.lazy_family = (u32)value;
  else if (strcmp(key, "fp64_long") == 0 && value <= 1) t.fp64_long = (u32)value;
  else if (strcmp(key, "walk14") == 0 && value <= 2) t.walk14 = (u32)value;
  else return -1;
  return 0;
}
#endif  // HX_TU_DISPATCH

// Where the persistent 14-stage tile walk (tile_walk) replaces one workgroup per tile.  Measured
// at N = 16384 x 8192 (profiles/r6_walk14_ab.txt): the inverse gains with every arithmetic policy
// (Small -17 %, Fp64 / Fp64L -8 %, Lazy -11 %, Harvey60 -8 %); the forward only with the Fp64
// family (-7 %), within +-2 % under the other policies.  "walk14": 0 = never, 1 = this table,
// 2 = always (A/B).
// MULTI (several moduli in one launch): the inverse walk of the Lazy family carries the plan's
// constants through the loop beside everything else, spills 52-64 bytes and measured 3 % slower than
// one workgroup per polynomial (8 moduli x 512 polynomials, tools/rns_ab.py; Harvey60 -7 %, Fp64L
// -10 % with it): not used there.
template <bool FWD, class A, bool MULTI = false>
static bool walk14_wanted() {
  const u32 w = tuning().walk14.load();
  if (w == 0) return false;
  if (w == 2) return true;
  if (MULTI && !FWD && A::kLazy) return false;
  return !FWD || A::kFp;
}

// Compute units of the current device (the grid of the persistent tile walk), cached per device.
static unsigned cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cached[dev].load(std::memory_order_relaxed);
  if (n <= 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return (unsigned)n;
}

template <bool FWD, class A>
static hipError_t launch_strided(int R, u64* out, const u64* in, const ulonglong2* tw,
                                 const ModConst& m, u32 log_n, u32 a0, u32 finish, u64 batch,
                                 const InvLast& il, hipStream_t st, const MultiCtx* mc = nullptr) {
  const u64 items = batch << (log_n - R);
  const unsigned grid = (unsigned)((items + 255) / 256);
  if (log_n < a0 + (u32)R + 6) return hipErrorInvalidValue;  // needs >= 64 columns per subtree
  ScopedKernelTimer timer(FWD ? "ntt_fwd_strided_pass" : "ntt_inv_strided_pass", st);
#define HX_LAUNCH_S(RR)                                                                   \
  case RR:                                                                                \
    if (mc) {                                                                             \
      if (log_n < (u32)RR + 8) return hipErrorNotSupported;                               \
      if (!FWD && a0 == 0)                                                                \
        hipLaunchKernelGGL((strided_pass_multi<FWD, RR, A, !FWD>), dim3(grid), dim3(256), \
                           0, st, out, in, *mc, log_n, a0, finish, items);                \
      else                                                                                \
        hipLaunchKernelGGL((strided_pass_multi<FWD, RR, A, false>), dim3(grid), dim3(256), \
                           0, st, out, in, *mc, log_n, a0, finish, items);                \
    } else if (!FWD && a0 == 0)                                                           \
      hipLaunchKernelGGL((strided_pass<FWD, RR, A, !FWD>), dim3(grid), dim3(256), 0, st, \
                         out, in, tw, m, log_n, a0, finish, items, il);                   \
    else                                                                                  \
      hipLaunchKernelGGL((strided_pass<FWD, RR, A, false>), dim3(grid), dim3(256), 0, st, \
                         out, in, tw, m, log_n, a0, finish, items, il);                   \
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

// Bottom pass: S stages on contiguous sub-blocks (CB = 0), tile of 2^TL elements.
template <bool FWD, int TL, class A>
static hipError_t launch_bottom(int S, u64* out, const u64* in, const ulonglong2* tw,
                                const ModConst& m, u32 log_n, u32 finish, u64 batch,
                                const InvLast& il, hipStream_t st, const MultiCtx* mc = nullptr,
                                const KsEpilogue* epi = nullptr) {
  const u64 total = batch << log_n;
  const unsigned grid = (unsigned)((total + (1u << TL) - 1) >> TL);
  const bool guard = (total & ((1u << TL) - 1)) != 0;  // the batch ends inside the last tile
  ScopedKernelTimer timer(FWD ? "ntt_fwd_tile_pass_bottom" : "ntt_inv_tile_pass_bottom", st);
  if (mc) {  // several moduli: the shapes N >= 4096 use
    if constexpr (TL >= 11) {
      if (log_n < (u32)TL || S < 11 || S > 14 || S > TL) return hipErrorNotSupported;
      const bool last = !FWD && (u32)S == log_n;
      if (epi) {  // the pass ends a KeySwitch: its output is folded into the result (KsEpilogue)
        if constexpr (FWD) {
          if (S != TL) return hipErrorNotSupported;
          hipLaunchKernelGGL((tile_pass_multi<true, TL, TL, A, false, true>), dim3(grid),
                             dim3(1 << (TL - re_of(TL))), 0, st, out, in, *mc, log_n, finish, total, *epi);
          return hipGetLastError();
        } else {
          return hipErrorInvalidValue;
        }
      }
#define HX_LAUNCH_BM(T, LST)                                                                \
  hipLaunchKernelGGL((tile_pass_multi<FWD, T, TL, A, LST>), dim3(grid), dim3(1 << (TL - re_of(T))), \
                     0, st, out, in, *mc, log_n, finish, total, NoEpilogue{})
      if (S == 11) {
        if constexpr (TL == 11) {
          if (last) HX_LAUNCH_BM(11, !FWD); else HX_LAUNCH_BM(11, false);
        } else {
          return hipErrorNotSupported;
        }
      } else if (S == 12) {
        if constexpr (TL == 12) {
          if (last) HX_LAUNCH_BM(12, !FWD); else HX_LAUNCH_BM(12, false);
        } else {
          return hipErrorNotSupported;
        }
      } else if (S == 13) {  // N = 8192 as one kernel (64 KiB tile)
        if constexpr (TL == 13) {
          if (last) HX_LAUNCH_BM(13, !FWD); else HX_LAUNCH_BM(13, false);
        } else {
          return hipErrorNotSupported;
        }
      } else {  // N = 16384 as one kernel (128 KiB tile)
        if constexpr (TL == 14) {
          if (log_n == 14 && walk14_wanted<FWD, A, true>() && grid > cu_count()) {  // persistent: see tile_walk
            const unsigned pg = cu_count();
            if (last)
              hipLaunchKernelGGL((tile_walk_pass_multi<FWD, 14, A, !FWD>), dim3(pg), dim3(1024), 0, st, out,
                                 in, *mc, log_n, finish, total);
            else
              hipLaunchKernelGGL((tile_walk_pass_multi<FWD, 14, A, false>), dim3(pg), dim3(1024), 0, st, out,
                                 in, *mc, log_n, finish, total);
          } else if (last) HX_LAUNCH_BM(14, !FWD); else HX_LAUNCH_BM(14, false);
        } else {
          return hipErrorNotSupported;
        }
      }
#undef HX_LAUNCH_BM
      return hipGetLastError();
    } else {
      return hipErrorNotSupported;
    }
  }
#define HX_LAUNCH_B2(T, G, LST)                                                           \
  hipLaunchKernelGGL((tile_pass<FWD, T, 0, TL, G, A, LST>),                               \
                     dim3(grid),                                                          \
                     dim3(1 << (TL - re_of(T))), 0, st, out, in, tw, m, log_n, finish, total, il)
#define HX_LAUNCH_B(T)                                                                    \
  case T:                                                                                 \
    if constexpr (T <= TL && (TL <= 10 || T >= 9) && (TL != 11 || T == 11) &&               \
                  (TL != 13 || T == 13) && (TL != 14 || T == 14)) {             \
      if (!FWD && (u32)T == log_n) {                                                      \
        if (guard)                                                                        \
          HX_LAUNCH_B2(T, true, !FWD);                                                    \
        else                                                                              \
          HX_LAUNCH_B2(T, false, !FWD);                                                   \
      } else {                                                                            \
        if (guard)                                                                        \
          HX_LAUNCH_B2(T, true, false);                                                   \
        else                                                                              \
          HX_LAUNCH_B2(T, false, false);                                                  \
      }                                                                                   \
    } else                                                                                \
      return hipErrorInvalidValue;                                                        \
    break;
  if constexpr (TL == 14) {
    // N = 16384 as one kernel, more tiles than CUs: the persistent walk (tile_walk)
    if (S == 14 && log_n == 14 && !guard && walk14_wanted<FWD, A>() && grid > cu_count()) {
      const unsigned pg = cu_count();
      if (!FWD)
        hipLaunchKernelGGL((tile_walk_pass<FWD, 14, A, !FWD>), dim3(pg), dim3(1024), 0, st, out, in, tw, m,
                           log_n, finish, total, il);
      else
        hipLaunchKernelGGL((tile_walk_pass<FWD, 14, A, false>), dim3(pg), dim3(1024), 0, st, out, in, tw, m,
                           log_n, finish, total, il);
      return hipGetLastError();
    }
  }
  switch (S) {
    HX_LAUNCH_B(1)
    HX_LAUNCH_B(2)
    HX_LAUNCH_B(3)
    HX_LAUNCH_B(4)
    HX_LAUNCH_B(5)
    HX_LAUNCH_B(6)
    HX_LAUNCH_B(7)
    HX_LAUNCH_B(8)
    HX_LAUNCH_B(9)
    HX_LAUNCH_B(10)
    HX_LAUNCH_B(11)
    HX_LAUNCH_B(12)
    HX_LAUNCH_B(13)
    HX_LAUNCH_B(14)
    default:
      return hipErrorInvalidValue;
  }
#undef HX_LAUNCH_B
#undef HX_LAUNCH_B2
  return hipGetLastError();
}

// Plan of one transform: `n_strided` register-only passes, then a bottom tile_pass of `bottom`
// stages on tiles of 2^tl elements.  Degrees up to 2^12 (and 2^13, 2^14 on their big tiles) are
// the tile pass alone: one launch, one HBM round trip; above, one strided pass (two for 2^20)
// plus the tile pass.  (The plans that were built, measured and not adopted -- both passes in one
// persistent launch, two LDS-tiled launches, mixed launches over chunks -- are archived under
// experiments/.)
struct Plan {
  int tl;      // tile size (log2) of the tile pass
  int n_strided;
  int strided[8];
  int bottom;
};

static Plan make_plan(int L, bool allow_tile13 = true, u64 batch = ~0ull) {
  Plan p{};
  if (L <= 12) {  // one kernel, one HBM round trip
    p.tl = L <= 10 ? 10 : 12;
    p.bottom = L;
    return p;
  }
  if (L == 14 && allow_tile13 && tuning().tile13.load() >= 2 && batch >= kTile14MinBatch) {
    // N = 16384: 128 KiB tile, one workgroup of 1024 threads x 16 elements per CU (one
    // workgroup per polynomial: smaller batches keep the two-pass shape, whose tile pass spreads a
    // polynomial over 8 workgroups.  Measured as wall time per dependent call, round 5
    // (profiles/r5_small_batch_ab.txt): one polynomial 10.9 us against 17.0 for the one-kernel
    // plan -- a single workgroup's 14 stages are a long serial chain -- 64 polynomials 15.4
    // against 18.0, 100 polynomials 19.7 against 18.4, 150 23.8 against 19.6: the plans cross
    // between 64 and 100)
    p.tl = 14;
    p.bottom = 14;
    return p;
  }
  if (L == 13 && allow_tile13 && tuning().tile13.load()) {
    // N = 8192 also fits one workgroup's LDS: 64 KiB tile, 1024 threads x 8 elements, two
    // workgroups (32 waves) per CU -- one HBM round trip instead of two
    p.tl = 13;
    p.bottom = 13;
    return p;
  }
  {
    // 11 bottom stages on 2048-element tiles up to N = 2^16 (strided pass of <= 5
    // stages: both kernels then carry a comparable share of the arithmetic), 12 on
    // 4096-element tiles above.
    p.bottom = L <= 16 ? 11 : 12;
    // N = 2^18, 2^19: five strided stages + the 13- / 14-stage tile pass on the 64 / 128 KiB
    // tiles of the one-kernel plans -- two HBM round trips instead of three (3 + 3 + 12,
    // 4 + 3 + 12 stages); round 3.  set_tuning("bigtile", 0): the three-pass plans.
    if (tuning().bigtile.load() && (L == 18 || L == 19)) p.bottom = L - 5;
    p.tl = p.bottom;
    int top = L - p.bottom;
    if (top <= 5) {
      p.strided[p.n_strided++] = top;
    } else {
      int passes = (top + 3) / 4;
      int base = top / passes, extra = top % passes;
      for (int i = 0; i < passes; ++i) p.strided[p.n_strided++] = base + (i < extra ? 1 : 0);
    }
    return p;
  }
}

// Buffers across the link (`mid` given: the *_host paths): one, two or three polynomials of N = 8192 keep
// the two-pass shape -- the one-kernel plan is ONE workgroup per polynomial, whose 64 KiB come over the
// link as one CU's outstanding requests; two passes spread a polynomial over 16 + 4 workgroups
// (one polynomial per synchronous host call: 24.1 -> 20.4 us, two 27.2 -> 25.7, four 36.5 either way, round 6;
// on device memory the one-kernel plan wins at every batch: 9.2 against 12 us per dependent call)
constexpr u64 kLinkTile13MinBatch = 4;
[[maybe_unused]] static bool link_allows_tile13(u32 log_n, u64 batch, bool link) {
  return !(link && log_n == 13 && batch < kLinkTile13MinBatch);
}

template <bool FWD, class A>
static hipError_t launch_bottom_tl(int tl, int S, u64* out, const u64* in, const ulonglong2* tw,
                                   const ModConst& m, u32 log_n, u32 finish, u64 batch,
                                   const InvLast& il, hipStream_t st,
                                   const MultiCtx* mc = nullptr, const KsEpilogue* epi = nullptr) {
  if (tl == 10)
    return launch_bottom<FWD, 10, A>(S, out, in, tw, m, log_n, finish, batch, il, st, mc, epi);
  if (tl == 11)
    return launch_bottom<FWD, 11, A>(S, out, in, tw, m, log_n, finish, batch, il, st, mc, epi);
  if (tl == 13)
    return launch_bottom<FWD, 13, A>(S, out, in, tw, m, log_n, finish, batch, il, st, mc, epi);
  if (tl == 14)
    return launch_bottom<FWD, 14, A>(S, out, in, tw, m, log_n, finish, batch, il, st, mc, epi);
  return launch_bottom<FWD, 12, A>(S, out, in, tw, m, log_n, finish, batch, il, st, mc, epi);
}


// One transform of `batch` polynomials on stream `st`.
template <class A>
static hipError_t forward_seq(const NttTables& t, const Plan& p, u64* result, const u64* operand,
                              u64 batch, u64 out_mf, hipStream_t st,
                              const MultiCtx* mc = nullptr, const KsEpilogue* epi = nullptr,
                              u64* mid = nullptr) {
  // `mid` (two-pass plans only): where the first pass hands over to the second instead of `result`
  // -- a device buffer when result and operand are host memory across the link (capi.cpp: the
  // bounce buffer, mapped caller memory), so each polynomial crosses the link once per direction
  u64* const hand = (mid && p.n_strided == 1) ? mid : result;
  const u64* src = operand;
  InvLast il{};
  hipError_t e;
  u32 first = kFirstPass;  // consumed by whichever pass runs first
  u32 a0 = 0;
  // Bounded members of the Lazy family: the bound of the (doubled) values walked through the
  // network -- 8q from the caller (input_mod_factor <= 4), +6q per stage (+4q with the exact
  // product) -- and the stages marked whose x operands must lose kLimit/2 * q first.
  int bound = 8;
  auto stage_mask = [&bound](int stages) -> u32 {
    u32 mask = 0;
    if constexpr (bounded_lazy<A>()) {
      constexpr int grow = A::kExact ? 4 : 6;
      for (int s = 0; s < stages; ++s) {
        if (bound + grow > A::kLimit) {
          mask |= 1u << s;
          bound = A::kLimit / 2;
        }
        bound += grow;
      }
    }
    return mask << kStageMaskShift;
  };
  for (int i = 0; i < p.n_strided; ++i) {
    e = launch_strided<true, A>(p.strided[i], hand, src, t.fwd, t.mod, t.log_n, a0,
                                first | stage_mask(p.strided[i]), batch, il, st, mc);
    if (e != hipSuccess) return e;
    a0 += p.strided[i];
    src = hand;
    first = 0;
  }
  const u32 fin = out_mf == 1 ? 2 : 1;
  return launch_bottom_tl<true, A>(p.tl, p.bottom, result, src, t.fwd, t.mod, t.log_n,
                                   fin | first | stage_mask(p.bottom), batch, il, st, mc, epi);
}

template <class A>
static hipError_t inverse_seq(const NttTables& t, const Plan& p, u64* result, const u64* operand,
                              u64 batch, u64 out_mf, hipStream_t st,
                              const MultiCtx* mc = nullptr, u64* mid = nullptr) {
  const u32 fin = out_mf == 1 ? 2 : 1;
  const bool only = p.n_strided == 0;
  u64* const hand = (mid && p.n_strided == 1) ? mid : result;  // (see forward_seq)
  hipError_t e = launch_bottom_tl<false, A>(p.tl, p.bottom, hand, operand, t.inv, t.mod, t.log_n,
                                            kFirstPass | (only ? fin : 0), batch, t.inv_last, st,
                                            mc);
  if (e != hipSuccess) return e;
  u32 a0 = t.log_n - (u32)p.bottom;
  for (int i = p.n_strided - 1; i >= 0; --i) {
    a0 -= p.strided[i];
    e = launch_strided<false, A>(p.strided[i], result, hand, t.inv, t.mod, t.log_n, a0,
                                 i == 0 ? fin : 0, batch, t.inv_last, st, mc);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// (Round-1 experiment, removed: cutting the batch into chunks that alternate
// between the caller's stream and an internal one, so that the HBM-bound strided
// pass of chunk k+1 runs concurrently with the VALU-bound tile pass of chunk k.
// rocprofv3 timelines showed the kernels do overlap, but each slows the other by
// exactly the time it gains: 3.64 ms per step against 3.57 ms back to back.)
template <bool FWD, class A>
static hipError_t transform_impl(const NttTables& t, u64* result, const u64* operand, u64 batch,
                                 u64 out_mf, hipStream_t st, u64* mid) {
  const Plan p = make_plan((int)t.log_n, link_allows_tile13(t.log_n, batch, mid != nullptr), batch);
  return FWD ? forward_seq<A>(t, p, result, operand, batch, out_mf, st, nullptr, nullptr, mid)
             : inverse_seq<A>(t, p, result, operand, batch, out_mf, st, nullptr, mid);
}

template <class A>
static hipError_t multi_impl(bool forward, const NttTables& t0, const MultiCtx& mc, u64 polys,
                             u64* result, const u64* operand, u64 out_mf, hipStream_t st,
                             const KsEpilogue* epi) {
  // the multi-plan kernels: 11 / 12 bottom stages, or the whole of N = 8192 / 16384
  Plan p = make_plan((int)t0.log_n, /*allow_tile13=*/true, polys);
  return forward ? forward_seq<A>(t0, p, result, operand, polys, out_mf, st, &mc, epi)
                 : inverse_seq<A>(t0, p, result, operand, polys, out_mf, st, &mc);
}

// Entry points of one arithmetic policy (see "Translation units" at the top).
#define HX_POLICY_ENTRY_DECL(NAME)                                                              \
  hipError_t transform_entry_##NAME(bool forward, const NttTables& t, u64* result,              \
                                    const u64* operand, u64 batch, u64 out_mf, hipStream_t st,  \
                                    u64* mid);                                                  \
  hipError_t multi_entry_##NAME(bool forward, const NttTables& t0, const MultiCtx& mc, u64 polys, \
                                u64* result, const u64* operand, u64 out_mf, hipStream_t st,    \
                                const KsEpilogue* epi);
HX_POLICY_ENTRY_DECL(small)
HX_POLICY_ENTRY_DECL(fp64)
HX_POLICY_ENTRY_DECL(lazy)
HX_POLICY_ENTRY_DECL(strict)
HX_POLICY_ENTRY_DECL(harvey60)
HX_POLICY_ENTRY_DECL(lazy32)
HX_POLICY_ENTRY_DECL(lazy16)
HX_POLICY_ENTRY_DECL(fp64l)
#undef HX_POLICY_ENTRY_DECL

#define HX_POLICY_ENTRY_DEF(NAME, A)                                                            \
  hipError_t transform_entry_##NAME(bool forward, const NttTables& t, u64* result,              \
                                    const u64* operand, u64 batch, u64 out_mf, hipStream_t st,  \
                                    u64* mid) {                                                 \
    return forward ? transform_impl<true, A>(t, result, operand, batch, out_mf, st, mid)        \
                   : transform_impl<false, A>(t, result, operand, batch, out_mf, st, mid);      \
  }                                                                                             \
  hipError_t multi_entry_##NAME(bool forward, const NttTables& t0, const MultiCtx& mc, u64 polys, \
                                u64* result, const u64* operand, u64 out_mf, hipStream_t st,    \
                                const KsEpilogue* epi) {                                        \
    return multi_impl<A>(forward, t0, mc, polys, result, operand, out_mf, st, epi);             \
  }
#if HX_TU_POLICY(0)
HX_POLICY_ENTRY_DEF(small, Small)
#endif
#if HX_TU_POLICY(1)
HX_POLICY_ENTRY_DEF(fp64, Fp64)
#endif
#if HX_TU_POLICY(2)
HX_POLICY_ENTRY_DEF(lazy, Lazy)
#endif
#if HX_TU_POLICY(3)
HX_POLICY_ENTRY_DEF(strict, Strict)
#endif
#if HX_TU_POLICY(4)
HX_POLICY_ENTRY_DEF(harvey60, Harvey60)
#endif
#if HX_TU_POLICY(5)
HX_POLICY_ENTRY_DEF(lazy32, Lazy32)
#endif
#if HX_TU_POLICY(6)
HX_POLICY_ENTRY_DEF(lazy16, Lazy16)
#endif
#if HX_TU_POLICY(7)
HX_POLICY_ENTRY_DEF(fp64l, Fp64L)
#endif
#undef HX_POLICY_ENTRY_DEF
static_assert(kPolicySmall == 0 && kPolicyFp64 == 1 && kPolicyLazy == 2 && kPolicyStrict == 3 &&
                  kPolicyHarvey60 == 4 && kPolicyLazy32 == 5 && kPolicyLazy16 == 6 && kPolicyFp64L == 7,
              "the HX_TU_POLICY numbers above are the ArithPolicy values");

#if HX_TU_DISPATCH
static hipError_t transform_dispatch(bool forward, const NttTables& t, u64* result,
                                     const u64* operand, u64 batch, u64 out_mf, hipStream_t st,
                                     u64* mid) {
  if (batch == 0) return hipSuccess;
  switch (t.policy) {
    case kPolicySmall: return transform_entry_small(forward, t, result, operand, batch, out_mf, st, mid);
    case kPolicyFp64: return transform_entry_fp64(forward, t, result, operand, batch, out_mf, st, mid);
    case kPolicyLazy: return transform_entry_lazy(forward, t, result, operand, batch, out_mf, st, mid);
    case kPolicyHarvey60:
      return transform_entry_harvey60(forward, t, result, operand, batch, out_mf, st, mid);
    case kPolicyLazy32: return transform_entry_lazy32(forward, t, result, operand, batch, out_mf, st, mid);
    case kPolicyLazy16: return transform_entry_lazy16(forward, t, result, operand, batch, out_mf, st, mid);
    case kPolicyFp64L: return transform_entry_fp64l(forward, t, result, operand, batch, out_mf, st, mid);
    default: return transform_entry_strict(forward, t, result, operand, batch, out_mf, st, mid);
  }
}

hipError_t ntt_forward_launch(const NttTables& t, u64* result, const u64* operand, u64 batch,
                              u64 out_mf, hipStream_t st, u64* mid) {
  return transform_dispatch(true, t, result, operand, batch, out_mf, st, mid);
}

hipError_t ntt_inverse_launch(const NttTables& t, u64* result, const u64* operand, u64 batch,
                              u64 out_mf, hipStream_t st, u64* mid) {
  return transform_dispatch(false, t, result, operand, batch, out_mf, st, mid);
}

bool ntt_is_single_kernel(const NttTables& t, u64 batch, bool link) {
  const Plan p = make_plan((int)t.log_n, link_allows_tile13(t.log_n, batch, link), batch);
  return p.n_strided == 0;
}

bool ntt_is_two_pass(const NttTables& t, u64 batch, bool link) {
  const Plan p = make_plan((int)t.log_n, link_allows_tile13(t.log_n, batch, link), batch);
  return p.n_strided == 1;
}
#endif  // HX_TU_DISPATCH

#if HX_TU_DISPATCH
// Whether the multi-plan kernels cover every pass of the plan make_plan picks for (log_n,
// polys): checked BEFORE the first launch of a multi-plan sequence, so that "not supported"
// always means "nothing enqueued" and the caller may fall back to plan-by-plan calls (a
// sequence refused half way -- a later pass, or a later arithmetic policy of a mixed group --
// would leave half-transformed data behind).  Mirrors the shape checks of launch_strided /
// launch_bottom.
static bool multi_plan_supported(u32 log_n, u64 polys) {
  const Plan p = make_plan((int)log_n, /*allow_tile13=*/true, polys);
  if (p.bottom < 11 || p.bottom > 14 || p.tl != p.bottom || log_n < (u32)p.tl) return false;
  u32 a0 = 0;
  for (int i = 0; i < p.n_strided; ++i) {
    const int r = p.strided[i];
    if (r < 1 || r > 5 || log_n < (u32)r + 8 || log_n < a0 + (u32)r + 6) return false;
    a0 += (u32)r;
  }
  return a0 + (u32)p.bottom == log_n;
}
hipError_t ntt_multi_launch(bool forward, const NttTables* const* tabs, u32 num_plans,
                            const MultiMap& map, u64 polys, u64* result, const u64* operand,
                            u64 out_mf, hipStream_t st, const KsEpilogue* epi) {
  if (num_plans == 0 || polys == 0) return hipSuccess;
  if (num_plans > (u32)kMaxMultiPlans || polys >= (1ull << 31) || map.inner == 0 ||
      map.period == 0 || map.period > (u32)kMaxMultiPeriod || (map.src_stride && map.inner != 1) ||
      (map.src_stride && !forward))  // (source maps: forward transforms only)
    return hipErrorNotSupported;
  const NttTables& t0 = *tabs[0];
  if (t0.log_n < 12 || t0.log_n > 17) return hipErrorNotSupported;  // one strided + one bottom pass
  MultiCtx mc{};
  bool have[kNumPolicies] = {};
  for (u32 k = 0; k < num_plans; ++k) {
    if (tabs[k]->log_n != t0.log_n || !tabs[k]->dev) return hipErrorNotSupported;
    mc.p[k] = tabs[k]->dev;
    mc.tw_fwd[k] = tabs[k]->fwd;
    mc.tw_inv[k] = tabs[k]->inv;
    mc.policy[k] = (uint8_t)tabs[k]->policy;
  }
  mc.map = map;
  for (u32 i = 0; i < map.period; ++i) {
    if (map.plan_tab[i] >= num_plans) return hipErrorInvalidValue;
    have[tabs[map.plan_tab[i]]->policy] = true;
  }
  int policies = 0;
  for (int i = 0; i < kNumPolicies; ++i) policies += have[i] ? 1 : 0;
  mc.one_policy = policies == 1 ? 1u : 0u;
  // nothing is enqueued unless every pass of every policy's sequence can be
  if (!multi_plan_supported(t0.log_n, polys)) return hipErrorNotSupported;
  if (epi && (!forward || out_mf != 4 || epi->decomp == 0 || epi->decomp > (u32)kKsMaxDecomp))
    return hipErrorInvalidValue;
  hipError_t e = hipSuccess;
  if (have[kPolicySmall] && e == hipSuccess)
    e = multi_entry_small(forward, t0, mc, polys, result, operand, out_mf, st, epi);
  if (have[kPolicyFp64] && e == hipSuccess)
    e = multi_entry_fp64(forward, t0, mc, polys, result, operand, out_mf, st, epi);
  if (have[kPolicyLazy] && e == hipSuccess)
    e = multi_entry_lazy(forward, t0, mc, polys, result, operand, out_mf, st, epi);
  if (have[kPolicyHarvey60] && e == hipSuccess)
    e = multi_entry_harvey60(forward, t0, mc, polys, result, operand, out_mf, st, epi);
  if (have[kPolicyLazy32] && e == hipSuccess)
    e = multi_entry_lazy32(forward, t0, mc, polys, result, operand, out_mf, st, epi);
  if (have[kPolicyLazy16] && e == hipSuccess)
    e = multi_entry_lazy16(forward, t0, mc, polys, result, operand, out_mf, st, epi);
  if (have[kPolicyFp64L] && e == hipSuccess)
    e = multi_entry_fp64l(forward, t0, mc, polys, result, operand, out_mf, st, epi);
  if (have[kPolicyStrict] && e == hipSuccess)
    e = multi_entry_strict(forward, t0, mc, polys, result, operand, out_mf, st, epi);
  // past the check above a refusal can only come after launches were made: a hard error,
  // never the "fall back" signal
  return e == hipErrorNotSupported ? hipErrorLaunchFailure : e;
}

// The arithmetic policy a plan for modulus q is built for (its tables depend on it).
// The arithmetic policy of a modulus (the counterpart of the reference's dispatch by modulus size,
// hexl/ntt/ntt-internal.cpp:202-239), read when a plan is created.  The tuning keys (set_tuning;
// no environment variable) only exist for A/B runs: every policy gives the same canonical outputs.
int choose_policy(u64 q) {
  const bool fp = tuning().fp64.load() != 0;
  // ("fp64" = 2: Fp64 also below 2^30, against the 32-bit Small policy)
  if (q < kSmallModulusBound && !(fp && tuning().fp64.load() == 2)) return kPolicySmall;
  // (below 2^47 the long-run member of the Fp64 family; "fp64_long" = 0: Fp64 for all of them;
  // "fp64" = 0: 31..50-bit moduli on the integer policies)
  if (q < kFp64LongModulusBound && fp && tuning().fp64_long.load() != 0) return kPolicyFp64L;
  if (q < kFp64ModulusBound && fp) return kPolicyFp64;
  // (the Lazy policy's quotient estimates shift the HIGH word of a value: q >= 2^32; with the
  // Fp64 policy switched off the moduli between 2^30 and 2^32 take the Harvey60 policy)
  if (q >= (1ull << 32) && q < kLazyModulusBound) return kPolicyLazy;
  // the bounded members of the Lazy family ("lazy_family" = 0: Harvey60 from 2^56 on)
  if (tuning().lazy_family.load() != 0) {
    if (q >= kLazyModulusBound && q < kLazy32ModulusBound) return kPolicyLazy32;
    if (q >= kLazy32ModulusBound && q < kLazy16ModulusBound) return kPolicyLazy16;
  }
  // ("h60" = 0: 2^56 <= q < 2^60 + 2^28 on the Strict policy)
  if (q < kHarvey60ModulusBound && tuning().h60.load() != 0) return kPolicyHarvey60;
  return kPolicyStrict;
}

#endif  // HX_TU_DISPATCH

}  // namespace hexl_amd
