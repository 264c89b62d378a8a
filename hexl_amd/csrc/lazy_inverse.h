// lazy_inverse.h -- the inverse (Gentleman-Sande) network under the Lazy arithmetic policy with
// its range bookkeeping done at COMPILE TIME.
//
// Under the Lazy policy (modarith.h; 2^32 <= q < 2^56) values are held doubled and an inverse
// butterfly never subtracts conditionally:
//     x' = x + y                      grows: bound(x') = bound(x) + bound(y)
//     y' = (x + OFF - y) * W  lazily  resets: y' < 6q   (OFF a multiple of 2q, >= y)
// so an element's size depends on how many consecutive stages it has been the SUM of, and every
// value has to stay below 2^63 (sign-test conditional subtraction, carry-free product chain).
// Rounds 1-3 restored [0, 8q) on every element that could exceed it at every subtree exit (72 of
// the 825 VALU instructions per thread of the 11-stage inverse tile pass) and
// bounded the sums of the first stage of a 5-stage subtree (64 of 1711).  Here the bound of every
// element at every stage -- exclusive, in units of q, doubled domain -- is tracked by a constexpr
// scheduler, and a reduction is emitted only where the NEXT use of a value would overflow
//     kLazyLimit = 128 <= floor(2^63 / q)   for every q < 2^56:
//   * inside a subtree: an input of a butterfly whose difference x + OFF - y could exceed the
//     limit is brought to [0, 4q) first (`pre`; the quotient estimate of lazy_estimate_reduce);
//   * at the exit of a subtree: elements above the threshold T the consumer of the values can
//     take (`post`: one conditional subtraction where that is enough, else the estimate).
// In a tile pass the consumer is the next round, whose threads each hold 2^r elements that sat at
// ONE position of this round's subtrees (different lanes: different positions), so the round is
// scheduled for the largest bound any position leaves; T = limit >> r keeps its sums in
// range without reductions of its own.  Between passes (in HBM) and from the caller every value
// is below kLazyHandOver = 12 (doubled; the caller's input_mod_factor <= 2 gives < 4).
// 11-stage tile pass: 7 estimates per thread instead of 72 instructions of ladders; 5-stage
// strided subtree: 3 estimates instead of 16 conditional subtractions.
//
// Host- and device-compilable: tests/cpp/host_arith_check.cpp replays whole networks through the
// very schedules the kernels are instantiated with and checks every intermediate against the
// bound the scheduler claims for it.
#pragma once
#include <utility>

#include "modarith.h"

namespace hexl_amd {

constexpr int kLazyLimit = 128;    // the Lazy policy's (q < 2^56)
constexpr int kLazyHandOver = 12;
// The other members of the family (modarith.h: LazyT<LIMIT>) run the same network scheduled for
// their smaller limit; what a pass hands over shrinks with it.
constexpr int lazy_handover(int limit) { return limit >= 128 ? kLazyHandOver : limit >= 32 ? 8 : 4; }
constexpr int kLazyMaxR = 5;  // deepest register subtree (the 5-stage strided pass)

// Schedule of an R-stage inverse subtree over E = 2^R elements; stages in execution order
// t = 0 .. R-1 (heap depth v = R-1-t inside the subtree, pairs 2^t apart).
struct InvSched {
  bool pre[kLazyMaxR][1 << kLazyMaxR];  // estimate-reduce element i before stage t uses it
  int off[kLazyMaxR][1 << kLazyMaxR];   // butterfly with LOWER element i at stage t: OFF = 2q << off
  int post[1 << kLazyMaxR];             // at exit: 0 nothing, -1 estimate, s > 0: x >= (2q << s) ? x - (2q << s) : x
  int out[1 << kLazyMaxR];              // bound of element i at exit (after post)
  int peak;                             // largest bound any intermediate reaches (<= the limit)
  int max_out;                          // max of out[]
  int estimates, csubs;                 // reductions emitted (per subtree)
};

// OFF = 2q << s is the smallest such multiple that covers a subtrahend below `by` q.
constexpr int lazy_off_shift(int by) {
  int s = 0;
  while ((2 << s) < by) ++s;
  return s;
}

// entry: every element < B q.  limit: bounds must not exceed it.  T: exit threshold (ignored
// with `last`, whose final stage multiplies both outputs by N^-1-scaled constants with the exact
// quotient: outputs < 4q).
constexpr InvSched make_inv_sched(int R, int B, int limit, int T, bool last) {
  InvSched s{};
  const int E = 1 << R;
  int b[1 << kLazyMaxR] = {};
  for (int i = 0; i < E; ++i) b[i] = B;
  s.peak = B;
  for (int t = 0; t < R; ++t) {
    const int half = 1 << t;
    for (int base = 0; base < E; base += 2 * half)
      for (int j = 0; j < half; ++j) {
        const int i = base + j, k = i + half;
        // the difference x + OFF - y is the largest intermediate of the butterfly
        if (b[i] + (2 << lazy_off_shift(b[k])) > limit) {
          if (b[k] >= b[i]) {
            s.pre[t][k] = true;
            b[k] = 4;
          } else {
            s.pre[t][i] = true;
            b[i] = 4;
          }
          ++s.estimates;
          if (b[i] + (2 << lazy_off_shift(b[k])) > limit) {
            if (s.pre[t][k]) {
              s.pre[t][i] = true;
              b[i] = 4;
            } else {
              s.pre[t][k] = true;
              b[k] = 4;
            }
            ++s.estimates;
          }
        }
        s.off[t][i] = lazy_off_shift(b[k]);
        const int d = b[i] + (2 << s.off[t][i]);
        if (d > s.peak) s.peak = d;
        if (last && t == R - 1) {
          b[i] = 4;
          b[k] = 4;
        } else {
          b[i] = b[i] + b[k];
          b[k] = 6;
        }
      }
  }
  for (int i = 0; i < E; ++i) {
    if (!last && b[i] > T) {
      // one conditional subtraction of c q (c = 2 << sh a power of two <= T) lands below c when
      // the value is below 2 c
      int sh = 0;
      while ((2 << (sh + 1)) <= T) ++sh;  // largest c = 2 << sh with c <= T
      if (T >= 4 && b[i] <= 2 * (2 << sh)) {
        s.post[i] = sh;
        b[i] = 2 << sh;
        ++s.csubs;
      } else {
        s.post[i] = -1;
        b[i] = 4;
        ++s.estimates;
      }
    }
    s.out[i] = b[i];
    if (b[i] > s.max_out) s.max_out = b[i];
  }
  return s;
}

// The schedule as a type: static data evaluated once, read only in constant expressions.
template <int R, int B, int T, bool LAST, int LIMIT = kLazyLimit>
struct InvSchedOf {
  static_assert(R >= 1 && R <= kLazyMaxR, "register subtrees are at most 5 stages deep");
  static constexpr InvSched value = make_inv_sched(R, B, LIMIT, T, LAST);
  static_assert(value.peak <= LIMIT, "lazy inverse schedule exceeds the 2^63 budget");
  static_assert(LAST || value.max_out <= T, "lazy inverse schedule misses its exit threshold");
};

// `for (i = 0; i < N; ++i) f(integral_constant<int, i>)` with i a constant expression
template <class F, int... I>
HX_HD void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
HX_HD void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// x >= (2q << s) ? x - (2q << s) : x   for x < 2^63
HX_HD u64 lazy_csub(u64 x, const ModConst& m, int s) { return csub_neg(x, m.neg_two_q << s); }

// One level (execution stage t of R) of the subtree, butterflies of groups [G0, G1) of the
// 2^(R-1-t) groups; wl[g - G0] is the twiddle of group g.  LAST (then t == R-1): the root stage
// of the whole transform, N^-1 folded in (ntt-radix-2.cpp:490-509).
// MONT (with LAST): the sum branch through scale_by_inverse_degree (N >= 64: its result is
// below 4q like the exact product's).
template <class SC, int R, int t, int G0, int G1, bool LAST, bool MONT = false, class TW>
HX_HD void inv_level_lazy(u64* x, const TW* wl, const ModConst& m, const InvLast& il) {
  constexpr int half = 1 << t;
  static_for<G1 - G0>([&](auto gc) __attribute__((always_inline)) {
    constexpr int g = G0 + decltype(gc)::value;
    static_for<half>([&](auto jc) __attribute__((always_inline)) {
      constexpr int i = g * 2 * half + decltype(jc)::value, k = i + half;
      if constexpr (SC::value.pre[t][i]) x[i] = lazy_estimate_reduce(x[i], m);
      if constexpr (SC::value.pre[t][k]) x[k] = lazy_estimate_reduce(x[k], m);
      const u64 s = x[i] + x[k];
      const u64 d = x[i] + (m.two_q << SC::value.off[t][i]) - x[k];
      if constexpr (LAST && t == R - 1) {
        if constexpr (MONT)
          x[i] = scale_by_inverse_degree(s, il);
        else
          x[i] = mul_add_lazy2<true>(0, s, il.n1, il.n1p, m.neg_two_q);
        x[k] = mul_add_lazy2<true>(0, d, il.n1w, il.n1wp, m.neg_two_q);
      } else {
        x[i] = s;
        x[k] = mul_add_lazy2<false>(0, d, wl[g - G0].x, wl[g - G0].y, m.neg_two_q);
      }
    });
  });
}

// exit of a subtree: the reductions the schedule asks for
template <class SC, int R>
HX_HD void inv_exit_lazy(u64* x, const ModConst& m) {
  static_for<(1 << R)>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (SC::value.post[i] < 0)
      x[i] = lazy_estimate_reduce(x[i], m);
    else if constexpr (SC::value.post[i] > 0)
      x[i] = lazy_csub(x[i], m, SC::value.post[i]);
  });
}

// R inverse stages on x[0 .. 2^R), deepest level first; wv[2^v + g] = twiddle of group g at
// depth v (the layout load_twiddles fills).  B: bound of the inputs; T: what the consumer takes.
template <int R, int B, int T, bool LAST, bool MONT = false, int LIMIT = kLazyLimit, class TW>
HX_HD void inv_subtree_lazy(u64* x, const TW* wv, const ModConst& m, const InvLast& il) {
  using SC = InvSchedOf<R, B, T, LAST, LIMIT>;
  static_for<R>([&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value, v = R - 1 - t;
    inv_level_lazy<SC, R, t, 0, (1 << v), LAST, MONT>(x, wv + (1 << v), m, il);
#if defined(__HIP_DEVICE_COMPILE__)
    // deep subtrees: keep the scheduler from interleaving every butterfly of a stage (all their
    // temporaries would be live at once and spill)
    if (R >= 4) __builtin_amdgcn_sched_barrier(0);
#endif
  });
  inv_exit_lazy<SC, R>(x, m);
}

// Chain of the inverse rounds of a tile pass (tile_geometry.h: round j has r(j) stages; the
// inverse executes NR-1 .. 0): entry bound and exit threshold of round j.  `rounds` = NR,
// `r0` = stages of round 0, `re` = stages of the others; the pass is entered with values below
// kLazyHandOver and, unless it ends the transform, left with values below it.
constexpr int lazy_chain_thresh(int j, int r0, int re, int limit = kLazyLimit) {
  if (j == 0) return lazy_handover(limit);
  const int t = limit >> (j == 1 ? r0 : re);
  return t < 8 ? 8 : t;  // (below 8 the next round reduces on entry where it has to instead)
}
constexpr int lazy_chain_entry(int j, int rounds, int r0, int re, int limit = kLazyLimit) {
  if (j == rounds - 1) return lazy_handover(limit);
  // what round j + 1 (>= 1: `re` stages) leaves
  return make_inv_sched(re, lazy_chain_entry(j + 1, rounds, r0, re, limit), limit,
                        lazy_chain_thresh(j + 1, r0, re, limit), false)
      .max_out;
}

}  // namespace hexl_amd
