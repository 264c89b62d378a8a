// lds_conflict_check.cpp -- every LDS access of every tile geometry the library instantiates,
// replayed on the CPU through the kernels' own index functions (hexl_amd/csrc/tile_geometry.h),
// must be free of bank conflicts.  Bank model of gfx950's LDS (64 banks of 4 bytes): a
// ds_read_b64 is served 32 lanes at a time over all 64 banks, a ds_write_b64 16 lanes at a time
// over 32 banks; two lanes of a group conflict when their 8-byte slots are different and fall
// into the same bank pair.  (The model was fitted to SQ_LDS_BANK_CONFLICT: zero for the
// 8-element geometries it predicts conflict-free, and it predicted the two-way conflicts the
// 16-element geometry had under the 8-element swizzle.)
#include <cstdio>
#include <set>
#include <vector>

#include "tile_geometry.h"

using namespace hexl_amd;

static int g_bad = 0;

// extra cycles of one 64-lane access: per group of `group` lanes, (largest number of distinct
// slots that share a bank pair) - 1
static int extra_cycles(const u32* slot, int group, int slots_per_pass) {
  int extra = 0;
  for (int g0 = 0; g0 < 64; g0 += group) {
    std::vector<std::set<u32>> bank(slots_per_pass);
    for (int l = g0; l < g0 + group; ++l) bank[slot[l] % slots_per_pass].insert(slot[l]);
    size_t worst = 1;
    for (auto& b : bank) worst = b.size() > worst ? b.size() : worst;
    extra += (int)worst - 1;
  }
  return extra;
}

static void check_access(const char* what, int S, int TL, int j, int e, const u32* slot) {
  const int rd = extra_cycles(slot, 32, 32), wr = extra_cycles(slot, 16, 16);
  if (rd || wr) {
    if (g_bad < 20)
      std::printf("conflict: S=%d TL=%d %s round %d element %d: read +%d write +%d cycles\n", S, TL,
                  what, j, e, rd, wr);
    ++g_bad;
  }
}

template <int S, int TL, int J>
static void check_rounds() {
  using RD = Rounds<S, 0>;
  constexpr int kRE = re_of(S), kE = el_of(S);
  if constexpr (J < RD::NR) {
    constexpr int r = RD::r(J), w = RD::w(J);
    constexpr int kThreads = 1 << (TL - kRE);
    for (int s = 0; s < (kE >> r); ++s)
      for (int e = 0; e < (1 << r); ++e)
        for (int wave = 0; wave < kThreads / 64; ++wave) {
          u32 slot[64];
          for (int l = 0; l < 64; ++l) {
            const u32 tid = (u32)(wave * 64 + l);
            // the address the kernels form: slot(element 0 of the thread) ^ slot(e << w)
            slot[l] = lds_slot<kRE>(tile_index<r, w>((u32)s * kThreads + tid, 0)) ^ lds_slot<kRE>((u32)e << w);
            if (slot[l] != lds_slot<kRE>(tile_index<r, w>((u32)s * kThreads + tid, e))) {
              std::printf("swizzle not linear: S=%d TL=%d round %d\n", S, TL, J);
              ++g_bad;
            }
          }
          check_access("round", S, TL, J, e, slot);
        }
    check_rounds<S, TL, J + 1>();
  }
}

template <int S, int TL>
static void check_geometry() {
  constexpr int kRE = re_of(S), kE = el_of(S);
  constexpr int kThreads = 1 << (TL - kRE);
  static_assert(kThreads >= 64, "a workgroup is at least one wave");
  // the swizzle permutes the tile's slots
  std::set<u32> seen;
  for (u32 p = 0; p < (1u << TL); ++p) seen.insert(lds_slot<kRE>(p));
  if (seen.size() != (1u << TL) || *seen.rbegin() != (1u << TL) - 1) {
    std::printf("swizzle is not a permutation of the tile: S=%d TL=%d\n", S, TL);
    ++g_bad;
  }
  check_rounds<S, TL, 0>();
  // copy-in (inverse) / copy-out (forward): the run a wave owns in the deepest round
  for (int i = 0; i < kE; ++i)
    for (int wave = 0; wave < kThreads / 64; ++wave) {
      u32 slot[64];
      for (int l = 0; l < 64; ++l) {
        const u32 tid = (u32)(wave * 64 + l);
        slot[l] = lds_slot<kRE>(xfer_p0<false, S, 0, TL>(tid, 0)) ^ lds_slot<kRE>(xfer_dp<false, S, 0>(i));
      }
      check_access("copy", S, TL, -1, i, slot);
    }
}

template <int TL, int S0, int S1>
static void check_range() {
  if constexpr (S0 <= S1) {
    check_geometry<S0, TL>();
    check_range<TL, S0 + 1, S1>();
  }
}

int main() {
  // the (stages, tile) pairs launch_bottom instantiates (ntt_kernels.hip): 2^10-element tiles
  // with 6..10 stages (N = 64 .. 1024; N = 2, 4, 16 and 32 -- a short round 0 whose gap is
  // below a swizzle field -- do run with two-way conflicts in that round, at sizes where nothing
  // is bandwidth-relevant), 2^12 with 9..12, and the whole-polynomial tiles 2^11 / 2^13 / 2^14
  check_range<10, 6, 10>();
  check_range<12, 9, 12>();
  check_geometry<11, 11>();
  check_geometry<13, 13>();
  check_geometry<14, 14>();
  if (g_bad) {
    std::printf("lds_conflict_check FAILED: %d access(es)\n", g_bad);
    return 1;
  }
  std::printf("lds_conflict_check OK\n");
  return 0;
}
