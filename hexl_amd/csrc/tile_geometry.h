// tile_geometry.h -- index arithmetic of the LDS-tiled pass (tile_pass, ntt_kernels.hip): elements
// per thread, the round structure of S stages, the tile index of every element a thread touches
// and the XOR swizzle of the LDS slots.  Host- and device-compilable: tests/cpp/lds_conflict_check.cpp
// walks every LDS access of every geometry the library instantiates through these very functions
// and counts bank conflicts on the CPU.
#pragma once
#include "modarith.h"

namespace hexl_amd {

// log2 elements per thread: 3 (8 elements, rounds of 3 stages) for every tile pass but the
// 14-stage one of N = 2^14, whose 128 KiB tile is one 1024-thread workgroup per CU with 16
// elements per thread (rounds of 4 stages, 4 waves per SIMD, <= 128 VGPRs).
constexpr int re_of(int S) { return S >= 14 ? 4 : 3; }
constexpr int el_of(int S) { return 1 << re_of(S); }
constexpr int kMaxTileLog = 14;

// XOR swizzle of the 8-byte slot index: every ds_read_b64 (32-lane groups, 64 banks) and
// ds_write_b64 (16-lane groups, 32 banks) access pattern of every round of every geometry
// the library instantiates for N >= 64 is bank-conflict free -- tests/cpp/lds_conflict_check.cpp
// walks them all through these functions; SQ_LDS_BANK_CONFLICT confirms.  No padding: a
// 4096-element tile is exactly 32 KiB, four workgroups = 32 waves per CU.  The swizzle is linear
// over XOR (the kernels form the address of element e as address(element 0) ^ constant).
// RE = log2 elements per thread.  The 16-element geometry (rounds of 4 stages: lane strides of
// 16 slots, runs of 16 lanes 256 slots apart, contiguous runs) has its own swizzle; with the
// 8-element one its two deepest rounds ran with two-way conflicts on every access.
template <int RE>
HX_HD constexpr u32 lds_slot(u32 p) {
  if (RE == 4) return p ^ ((p >> 4) & 63);
  return p ^ ((p >> 3) & 7) ^ (((p >> 6) & 7) << 3);
}

// Round structure of S stages: round 0 takes the remainder S - 3*(NR-1) stages
// (1..3), the others 3.  w(j) is log2 of the finest butterfly gap of round j in
// tile-index units (it includes the CB column bits).
template <int S, int CB>
struct Rounds {
  static constexpr int kRE = re_of(S);
  static constexpr int kE = 1 << kRE;
  static constexpr int NR = (S + kRE - 1) / kRE;
  static constexpr int R0 = S - (NR - 1) * kRE;
  static constexpr int r(int j) { return j == 0 ? R0 : kRE; }
  static constexpr int u(int j) { return j == 0 ? 0 : R0 + (j - 1) * kRE; }
  static constexpr int w(int j) { return CB + S - u(j) - r(j); }
  // Twiddles of round j are per-lane vector loads (gap < one wave) or wave-uniform
  // scalar loads.
  static constexpr bool vec(int j) { return w(j) < 6; }
  // Software pipelining of the twiddle fetch: wave-uniform twiddles (SGPRs, no
  // VGPR cost) of a round are requested before the arithmetic of the round
  // executed just before it.  (Doing the same for per-lane twiddles needs 28 more
  // VGPRs during a round and spills under the 64-VGPR cap of 8 waves per SIMD.)
  // Forward executes rounds 0..NR-1, inverse NR-1..0.
  // (round 6: the same for the per-lane twiddles of round 2 of the 16-element geometry -- requested during
  // round 1, whose twiddles are scalar: fits the 128 VGPRs, measured flat at N = 16384, not kept)
  static constexpr bool pre_fwd(int j) { return j >= 1 && j < NR && !vec(j); }
  static constexpr bool pre_inv(int j) { return j >= 0 && j <= NR - 2 && !vec(j); }
  // Fp64 family, forward: a pass starts from fully reduced values; all elements are reduced
  // again after round j when running round j + 1 too would make the run longer than `run`
  // stages (modarith.h: kFwdRun of the policy).  fwd_run(j) = stages since the last reduction
  // at the end of round j.
  static constexpr int fwd_run(int j, int run = kFpFwdRun) {
    int c = 0;
    for (int i = 0; i <= j; ++i) {
      c += r(i);
      if (i < j && c + r(i + 1) > run) c = 0;
    }
    return c;
  }
  static constexpr bool fp_reduce_after(int j, int run = kFpFwdRun) {
    return j + 1 < NR && fwd_run(j, run) + r(j + 1) > run;
  }
  // The same for the inverse, whose rounds execute NR-1 .. 0 (kInvRun of the policy): all
  // elements are reduced after round j when round j - 1 would make the run too long; after
  // round 0 unless the pass ends the transform (the hand-over between passes is reduced, the
  // finish reduces by itself).  inv_run(j) = stages since the last reduction at the end of
  // round j.
  static constexpr int inv_run(int j, int run) {
    int c = 0;
    for (int i = NR - 1; i >= j; --i) {
      c += r(i);
      if (i > j && c + r(i - 1) > run) c = 0;
    }
    return c;
  }
  static constexpr bool fp_inv_reduce_after(int j, int run, bool last) {
    return j == 0 ? !last : inv_run(j, run) + r(j - 1) > run;
  }
};

// Tile index of element e of virtual thread vt in a round with r stages whose
// finest gap is 2^w.
template <int r, int w>
HX_HD u32 tile_index(u32 vt, int e) {
  return ((vt >> w) << (w + r)) + ((u32)e << w) + (vt & ((1u << w) - 1));
}

// Tile index of the i-th element a thread moves between global memory and its
// registers, as (per-thread p0, uniform dp).  ROUND0: the thread's round-0 set
// (forward fetch, inverse store); otherwise its slice of the 512-element run its
// wave owns in the deepest round (inverse fetch, forward store), 64 per access.
template <bool ROUND0, int S, int CB, int TL>
HX_HD u32 xfer_p0(u32 tid, int i) {
  constexpr int kRE = re_of(S), kE = el_of(S);
  using RD = Rounds<S, CB>;
  if (ROUND0) {
    constexpr int r = RD::r(0), w = RD::w(0);
    return tile_index<r, w>(((u32)(i >> r) << (TL - kRE)) + tid, 0);
  }
  return ((tid >> 6) << (kRE + 6)) + (tid & 63);
}
template <bool ROUND0, int S, int CB>
HX_HD constexpr u32 xfer_dp(int i) {
  using RD = Rounds<S, CB>;
  return ROUND0 ? (u32)(i & ((1 << RD::r(0)) - 1)) << RD::w(0) : (u32)i << 6;
}

}  // namespace hexl_amd
