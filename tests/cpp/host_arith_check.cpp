// host_arith_check.cpp -- CPU check of hexl_amd/csrc/modarith.h (the host path of
// the very functions the HIP kernels inline) against the oracle.
//
// The transform networks are replayed stage by stage with the device butterflies:
// forward = plain radix-2 Cooley-Tukey order; inverse = Gentleman-Sande order.  The Lazy
// policy's inverse is cut into passes and rounds exactly like the kernels cut it (a tile pass
// of S stages = rounds of tile_geometry.h, strided passes of <= 5 stages) and every register
// subtree is run through the SAME compile-time schedule the kernels are instantiated with
// (lazy_inverse.h: make_inv_sched, lazy_chain_entry / _thresh), by an interpreter that checks
// every intermediate against the bound the scheduler claims for it (and all of them against
// 2^63), and through the templated device functions themselves, whose outputs must be
// identical.  Outputs are checked against the oracle, bit for bit.
//
// Build/run: see tests/test_host_arith.py.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lazy_inverse.h"
#include "modarith.h"
#include "tile_geometry.h"

extern "C" {
#include "hexl_oracle.h"
}

using namespace hexl_amd;

static int g_fail = 0;
static int g_cases = 0;
#define EXPECT(c, ...)                 \
  do {                                 \
    if (!(c)) {                        \
      if (g_fail < 400) {               \
        fprintf(stderr, __VA_ARGS__);  \
        fprintf(stderr, "\n");         \
      }                                \
      ++g_fail;                        \
    }                                  \
  } while (0)

static u64 rng_state = 0x1234567ull;
static u64 rnd() {
  u64 z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <class A>
static void check(u64 n, u64 q, const std::vector<int>& inv_runs, u64 in_mf_f, u64 in_mf_i) {
  int L = 0;
  while ((1ull << L) < n) ++L;
  const u64 shoup = A::kSmall ? 32 : (A::kLazy || A::kH60) ? 63 : 64;
  std::vector<u64> R(n), Rp(n), Ri_stage(n), Rip_stage(n);
  ho_ntt_tables(n, q, ho_minimal_primitive_root(2 * n, q), R.data(), Rp.data(), Ri_stage.data(),
                Rip_stage.data());
  std::vector<u64> W(n), Wp(n), V(n), Vp(n);  // heap order, forward and inverse
  for (u64 i = 1; i < n; ++i) {
    W[i] = R[i];
    Wp[i] = ho_multiply_factor(W[i], shoup, q);
    V[i] = ho_inverse_mod(R[i], q);
    Vp[i] = ho_multiply_factor(V[i], shoup, q);
  }
  const ModConst m = make_mod_const(q);
  ++g_cases;
  // doubled values: below 2^63 (Lazy), high word <= 2^31 (Harvey60)
  const u64 lim = A::kLazy ? (1ull << 63) : A::kH60 ? (1ull << 63) + (1ull << 32) : ~0ull;

  for (int canonical = 0; canonical < 2; ++canonical) {
    // ---------------- forward
    std::vector<u64> in(n), ref(n), x(n);
    for (auto& v : in) v = rnd() % (in_mf_f * q);
    // adversarial values at the range ends
    in[0] = in_mf_f * q - 1;
    in[1] = 0;
    in[n - 1] = in_mf_f * q - 1;
    ho_ntt_forward_radix2(ref.data(), in.data(), n, q, R.data(), Rp.data(), in_mf_f, 1);
    for (u64 i = 0; i < n; ++i) x[i] = to_internal<A>(in[i], m);
    // Lazy family: the bound of the doubled values walked like the library's host code walks it
    // (ntt_kernels.hip: forward_seq) -- 8q from the caller, + 6q per stage (+ 4q with the exact
    // product); a bounded member subtracts kLimit/2 * q from the x operands of a stage whose
    // growth would pass kLimit * q
    u64 fbound = 8;
    const u64 grow = A::kExact ? 4 : 6;
    for (int s = 0; s < L; ++s) {
      const u64 mgroups = 1ull << s, t = n >> (s + 1);
      bool subtract = false;
      if (A::kLazy) {
        if (A::kLimit < kLazyLimit && fbound + grow > (u64)A::kLimit) {
          subtract = true;
          fbound = A::kLimit / 2;
        }
        fbound += grow;
        EXPECT(fbound <= (u64)(A::kLazy ? A::kLimit : 0), "fwd lazy: bound %llu passes the limit", (unsigned long long)fbound);
      }
      int csub_shift = 0;
      while (A::kLazy && (4 << csub_shift) < A::kLimit) ++csub_shift;
      for (u64 i = 0; i < mgroups; ++i)
        for (u64 j = 0; j < t; ++j) {
          u64& a = x[2 * i * t + j];
          u64& b = x[2 * i * t + j + t];
          if (subtract) {
            EXPECT(a < (u64)A::kLimit * q, "fwd lazy: conditional subtraction out of its range");
            a = lazy_csub(a, m, csub_shift);
            EXPECT(a < (u64)(A::kLimit / 2) * q, "fwd lazy: conditional subtraction result");
          }
          fwd_butterfly<A>(a, b, W[mgroups + i], Wp[mgroups + i], m);
          EXPECT(a < lim && b < lim, "fwd range: stage %d", s);
          if (A::kLazy)
            EXPECT(a < fbound * q && b < fbound * q, "fwd lazy bound: stage %d", s);
          else if (A::kH60)  // Harvey's [0,4q) on doubled values
            EXPECT(a < 8 * q && b < 8 * q, "fwd harvey60 bound: stage %d", s);
          else
            EXPECT(a < 4 * q && b < 4 * q, "fwd strict bound: stage %d", s);
        }
    }
    for (u64 i = 0; i < n; ++i) {
      const u64 r = fwd_finish<A>(x[i], m, canonical);
      if (canonical)
        EXPECT(r == ref[i], "fwd canonical mismatch at %llu: %llu vs %llu",
               (unsigned long long)i, (unsigned long long)r, (unsigned long long)ref[i]);
      else
        EXPECT(r < 4 * q && r % q == ref[i], "fwd lazy mismatch at %llu", (unsigned long long)i);
    }

    // ---------------- inverse (the Lazy policy's: check_lazy_inverse below)
    if constexpr (A::kLazy) continue;
    else {
    for (auto& v : in) v = rnd() % (in_mf_i * q);
    in[0] = in_mf_i * q - 1;
    in[1] = in_mf_i * q - 1;
    in[2] = 0;
    ho_ntt_inverse_radix2(ref.data(), in.data(), n, q, Ri_stage.data(), Rip_stage.data(), in_mf_i,
                          1);
    for (u64 i = 0; i < n; ++i) x[i] = to_internal<A>(in[i], m);
    InvLast il{};
    il.n1 = ho_inverse_mod(n, q);
    il.n1w = ho_multiply_mod(il.n1, V[1], q);
    il.n1p = ho_multiply_factor(il.n1, shoup, q);
    il.n1wp = ho_multiply_factor(il.n1w, shoup, q);
    il.c2 = (q - 1) / n;
    il.log_n = (u32)L;
    il.mont_mask = (u32)((A::kH60 ? 2 * n : n) - 1);
    int stage = L - 1;  // heap level of the stage to run; deepest first
    for (size_t ri = 0; ri < inv_runs.size() && stage >= 0; ++ri) {
      int len = inv_runs[ri];
      if (len > stage + 1) len = stage + 1;
      for (int tt = 0; tt < len; ++tt, --stage) {
        const u64 mgroups = 1ull << stage, t = n >> (stage + 1);
        for (u64 i = 0; i < mgroups; ++i)
          for (u64 j = 0; j < t; ++j) {
            const u64 ia = 2 * i * t + j, ib = ia + t;
            if (stage == 0) {
              // (the multiply-free N^-1 scaling of the sum branch from N = 64 on, as in the kernels)
              if (n >= 64 && !A::kSmall)
                inv_butterfly_last<A, true>(x[ia], x[ib], il, m);
              else
                inv_butterfly_last<A, false>(x[ia], x[ib], il, m);
            } else {
              inv_butterfly<A>(x[ia], x[ib], V[mgroups + i], Vp[mgroups + i], m);
            }
            EXPECT(x[ia] < lim && x[ib] < lim, "inv range: stage %d", stage);
          }
      }
      if (stage >= 0)
        for (u64 i = 0; i < n; ++i) EXPECT(x[i] < (A::kH60 ? 4 : 2) * q, "inv strict bound");
    }
    EXPECT(stage == -1, "runs do not cover the network");
    for (u64 i = 0; i < n; ++i) {
      const u64 r = inv_finish<A>(x[i], m, canonical);
      if (canonical)
        EXPECT(r == ref[i], "inv canonical mismatch at %llu: %llu vs %llu",
               (unsigned long long)i, (unsigned long long)r, (unsigned long long)ref[i]);
      else
        EXPECT(r < 2 * q && r % q == ref[i], "inv lazy mismatch at %llu", (unsigned long long)i);
    }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Lazy policy, inverse network: passes and rounds as the kernels cut them, every subtree through
// the schedule the kernels use.
struct HostTw {
  u64 x, y;
};

// One subtree through schedule `sc` (entered below B q), the interpreter: the same operations as
// inv_level_lazy / inv_exit_lazy, each intermediate checked against the scheduler's bookkeeping.
static void run_sched(const InvSched& sc, int limit, int R, int B, bool last, bool mont, u64* x,
                      const HostTw* wv, const ModConst& m, const InvLast& il) {
  const int E = 1 << R;
  const u64 q = m.q;
  std::vector<u64> b(E, (u64)B);
  for (int i = 0; i < E; ++i) EXPECT(x[i] < b[i] * q, "lazy inverse: entry bound %d", B);
  for (int t = 0; t < R; ++t) {
    const int half = 1 << t, v = R - 1 - t;
    for (int g = 0; g < (1 << v); ++g)
      for (int j = 0; j < half; ++j) {
        const int i = g * 2 * half + j, k = i + half;
        if (sc.pre[t][i]) {
          x[i] = lazy_estimate_reduce(x[i], m);
          b[i] = 4;
          EXPECT(x[i] < 4 * q, "lazy inverse: estimate lands below 4q");
        }
        if (sc.pre[t][k]) {
          x[k] = lazy_estimate_reduce(x[k], m);
          b[k] = 4;
          EXPECT(x[k] < 4 * q, "lazy inverse: estimate lands below 4q");
        }
        const u64 off = m.two_q << sc.off[t][i];
        EXPECT(off >= b[k] * q || off >= x[k], "lazy inverse: offset covers the subtrahend");
        EXPECT(x[k] <= off, "lazy inverse: difference would wrap");
        const u64 sum = x[i] + x[k], d = x[i] + off - x[k];
        EXPECT((b[i] + ((u64)2 << sc.off[t][i])) <= (u64)limit, "lazy inverse: schedule passes the limit");
        EXPECT(sum < (1ull << 63) && d < (1ull << 63), "lazy inverse: 2^63");
        if (last && t == R - 1) {
          x[i] = mont ? scale_by_inverse_degree(sum, il) : mul_add_lazy2<true>(0, sum, il.n1, il.n1p, m.neg_two_q);
          x[k] = mul_add_lazy2<true>(0, d, il.n1w, il.n1wp, m.neg_two_q);
          b[i] = b[k] = 4;
          EXPECT((x[i] & 1) == 0 && (x[k] & 1) == 0, "lazy inverse: doubled values are even");
        } else {
          x[i] = sum;
          x[k] = mul_add_lazy2<false>(0, d, wv[(1 << v) + g].x, wv[(1 << v) + g].y, m.neg_two_q);
          b[i] = b[i] + b[k];
          b[k] = 6;
        }
        EXPECT(x[i] < b[i] * q && x[k] < b[k] * q, "lazy inverse: stage bound (R %d t %d)", R, t);
      }
  }
  for (int i = 0; i < E; ++i) {
    if (sc.post[i] < 0) x[i] = lazy_estimate_reduce(x[i], m);
    else if (sc.post[i] > 0) x[i] = lazy_csub(x[i], m, sc.post[i]);
    EXPECT(x[i] < (u64)sc.out[i] * q, "lazy inverse: exit bound %d of element %d", sc.out[i], i);
  }
}

// The templated device functions on the same subtree: (R, B, T, LAST, MONT) must be one of the
// instantiations below -- the set the chains of every tile geometry and the strided passes use.
// (limit, R, B, T)
#define LAZY_INSTANCES(X)                                                                          \
  X(128, 2, 32, 12) X(128, 1, 64, 12) X(128, 1, 12, 12) X(128, 2, 12, 12) X(128, 3, 12, 12)         \
  X(128, 4, 12, 12) X(128, 5, 12, 12) X(128, 3, 12, 16) X(128, 3, 16, 16) X(128, 3, 16, 32)         \
  X(128, 3, 16, 64) X(128, 2, 24, 12) X(128, 1, 24, 12) X(128, 1, 16, 12) X(128, 3, 16, 12)         \
  X(128, 4, 12, 8) X(128, 4, 8, 8) X(128, 4, 8, 32) X(128, 2, 16, 12) X(128, 3, 12, 32)             \
  X(128, 3, 12, 64) X(128, 4, 12, 32) X(128, 4, 8, 12) X(128, 4, 12, 64) X(128, 4, 8, 64)           \
  LAZY_INSTANCES_BOUNDED(X)
#define LAZY_INSTANCES_BOUNDED(X)                                                                  \
  X(32, 1, 8, 8) X(32, 2, 8, 8) X(32, 3, 8, 8) X(32, 4, 8, 8) X(32, 5, 8, 8) X(32, 1, 16, 8)        \
  X(32, 3, 8, 16) X(32, 4, 8, 16) X(32, 2, 16, 8) X(32, 4, 16, 8)                                   \
  X(16, 1, 4, 4) X(16, 2, 4, 4) X(16, 3, 4, 4) X(16, 4, 4, 4) X(16, 5, 4, 4) X(16, 3, 4, 8)         \
  X(16, 3, 8, 8) X(16, 2, 8, 4) X(16, 1, 8, 4) X(16, 3, 8, 4) X(16, 4, 4, 8) X(16, 4, 8, 8) X(16, 4, 8, 4)
static bool run_template(int limit, int R, int B, int T, bool last, bool mont, u64* x, const HostTw* wv,
                         const ModConst& m, const InvLast& il) {
#define X(LL, RR, BB, TT)                                                                    \
  if (limit == LL && R == RR && B == BB && T == TT) {                                        \
    if (last && mont) inv_subtree_lazy<RR, BB, TT, true, true, LL>(x, wv, m, il);            \
    else if (last) inv_subtree_lazy<RR, BB, TT, true, false, LL>(x, wv, m, il);              \
    else inv_subtree_lazy<RR, BB, TT, false, false, LL>(x, wv, m, il);                       \
    return true;                                                                             \
  }
  LAZY_INSTANCES(X)
#undef X
  return false;
}

struct LazyPass {
  int stages;   // S
  bool tile;    // tile pass (rounds of tile_geometry.h) or one strided subtree
};

static void round_shape(int S, int& rounds, int& r0, int& re) {
  re = re_of(S);
  rounds = (S + re - 1) / re;
  r0 = S - (rounds - 1) * re;
}

static void check_lazy_inverse(u64 n, u64 q, const std::vector<LazyPass>& passes, u64 in_mf,
                               int limit = kLazyLimit) {
  int L = 0;
  while ((1ull << L) < n) ++L;
  std::vector<u64> R(n), Rp(n), Ri_stage(n), Rip_stage(n);
  ho_ntt_tables(n, q, ho_minimal_primitive_root(2 * n, q), R.data(), Rp.data(), Ri_stage.data(),
                Rip_stage.data());
  std::vector<HostTw> V(n);
  for (u64 i = 1; i < n; ++i) {
    V[i].x = ho_inverse_mod(R[i], q);
    V[i].y = ho_multiply_factor(V[i].x, 63, q);
  }
  const ModConst m = make_mod_const(q);
  InvLast il{};
  il.n1 = ho_inverse_mod(n, q);
  il.n1w = ho_multiply_mod(il.n1, V[1].x, q);
  il.n1p = ho_multiply_factor(il.n1, 63, q);
  il.n1wp = ho_multiply_factor(il.n1w, 63, q);
  il.c2 = (q - 1) / n;
  il.log_n = (u32)L;
  il.mont_mask = (u32)(2 * n - 1);
  ++g_cases;
  std::vector<u64> in(n), ref(n), x(n);
  for (auto& v : in) v = rnd() % (in_mf * q);
  in[0] = in_mf * q - 1;
  in[1] = in_mf * q - 1;
  in[2] = 0;
  for (u64 i = 3; i < n && i < 64; i += 3) in[i] = in_mf * q - 1 - (rnd() & 3);  // large everywhere
  ho_ntt_inverse_radix2(ref.data(), in.data(), n, q, Ri_stage.data(), Rip_stage.data(), in_mf, 1);
  for (u64 i = 0; i < n; ++i) x[i] = to_internal<Lazy>(in[i], m);
  int done = 0;  // stages run so far (deepest first)
  for (size_t pi = 0; pi < passes.size(); ++pi) {
    const LazyPass& p = passes[pi];
    const bool pass_last = done + p.stages == L;
    EXPECT(done + p.stages <= L, "passes longer than the network");
    int rounds = 1, r0 = p.stages, re = p.stages;
    if (p.tile) round_shape(p.stages, rounds, r0, re);
    int in_pass = 0;
    for (int j = rounds - 1; j >= 0; --j) {
      const int r = (j == 0) ? r0 : re;
      const int B = p.tile ? lazy_chain_entry(j, rounds, r0, re, limit) : lazy_handover(limit);
      const int T = p.tile ? lazy_chain_thresh(j, r0, re, limit) : lazy_handover(limit);
      const bool last = pass_last && j == 0;
      const bool mont = last && (p.tile ? p.stages >= 6 : true);
      const InvSched sc = make_inv_sched(r, B, limit, T, last);
      EXPECT(sc.peak <= limit && (last || sc.max_out <= T), "schedule (limit %d r %d B %d T %d)", limit, r, B, T);
      // the subtrees of this round: heap levels [top, top + r), top = L - done - in_pass - r
      const int top = L - done - in_pass - r;
      const u64 cols = n >> (top + r);
      const int E = 1 << r;
      std::vector<u64> a(E), c(E);
      std::vector<HostTw> wv(2 * E);
      bool have_template = true;
      for (u64 h = 0; h < (1ull << top); ++h)
        for (u64 col = 0; col < cols; ++col) {
          for (int e = 0; e < E; ++e) a[e] = c[e] = x[(h * E + e) * cols + col];
          for (int v = 0; v < r; ++v)
            for (int g = 0; g < (1 << v); ++g) wv[(1 << v) + g] = V[(((1ull << top) + h) << v) + g];
          run_sched(sc, limit, r, B, last, mont, a.data(), wv.data(), m, il);
          if (col < 4 || col + 2 > cols) {  // the device templates on a sample of the subtrees
            have_template = run_template(limit, r, B, T, last, mont, c.data(), wv.data(), m, il);
            if (have_template)
              for (int e = 0; e < E; ++e) EXPECT(a[e] == c[e], "template != interpreter (r %d B %d T %d)", r, B, T);
          }
          for (int e = 0; e < E; ++e) x[(h * E + e) * cols + col] = a[e];
        }
      EXPECT(have_template, "no inv_subtree_lazy<%d, %d, %d, ., ., %d> instance in the test: add X(%d, %d, %d, %d) to LAZY_INSTANCES",
             r, B, T, limit, limit, r, B, T);
      in_pass += r;
    }
    done += p.stages;
    if (!pass_last)
      for (u64 i = 0; i < n; ++i) EXPECT(x[i] < (u64)lazy_handover(limit) * q, "lazy inverse: hand-over bound between passes");
  }
  EXPECT(done == L, "passes do not cover the network");
  for (int canonical = 0; canonical < 2; ++canonical)
    for (u64 i = 0; i < n; ++i) {
      const u64 r = inv_finish<Lazy>(x[i], m, canonical);
      if (canonical)
        EXPECT(r == ref[i], "lazy inv canonical mismatch at %llu", (unsigned long long)i);
      else
        EXPECT(r < 2 * q && r % q == ref[i], "lazy inv lazy-output mismatch at %llu", (unsigned long long)i);
    }
}

// the plan the library picks for degree 2^L (ntt_kernels.hip: make_plan), as passes in inverse order
static std::vector<LazyPass> library_passes(int L, bool one_kernel_14 = true) {
  if (L <= 12 || L == 13 || (L == 14 && one_kernel_14)) return {{L, true}};
  int bottom = L <= 16 ? 11 : 12;
  if (L == 18 || L == 19) bottom = L - 5;
  std::vector<LazyPass> p = {{bottom, true}};
  int top = L - bottom;
  if (top <= 5) {
    p.push_back({top, false});
  } else {
    const int passes = (top + 3) / 4, base = top / passes, extra = top % passes;
    std::vector<int> sizes;
    for (int i = 0; i < passes; ++i) sizes.push_back(base + (i < extra ? 1 : 0));
    for (int i = passes - 1; i >= 0; --i) p.push_back({sizes[i], false});  // inverse order
  }
  return p;
}

// Fp64 policy (2^30 <= q < 2^50): exact integers in doubles, balanced twiddles.  The
// forward network is cut into runs of <= kFpFwdRun stages and the inverse into runs of
// <= kFpInvRun, every element fully reduced at each run end -- as the kernels do.  Checks
// every intermediate against 2^53 and against the bound the analysis in modarith.h
// promises, and the outputs against the oracle bit for bit.
#include <cmath>
static void check_fp(u64 n, u64 q, const std::vector<int>& fwd_runs,
                     const std::vector<int>& inv_runs, u64 in_mf_f, u64 in_mf_i,
                     int max_fwd_run = kFpFwdRun, int max_inv_run = kFpInvRun) {
  int L = 0;
  while ((1ull << L) < n) ++L;
  std::vector<u64> R(n), Rp(n), Ri_stage(n), Rip_stage(n);
  ho_ntt_tables(n, q, ho_minimal_primitive_root(2 * n, q), R.data(), Rp.data(), Ri_stage.data(),
                Rip_stage.data());
  auto balanced = [q](u64 w) { return w > q / 2 ? -(double)(q - w) : (double)w; };
  std::vector<double> W(n), V(n);
  for (u64 i = 1; i < n; ++i) {
    W[i] = balanced(R[i]);
    V[i] = balanced(ho_inverse_mod(R[i], q));
  }
  const ModConst m = make_mod_const(q);
  ++g_cases;
  const double two53 = 9007199254740992.0, qd = (double)q;
  // per unit of |y| / q the quotient estimate is off by at most 1.5 q 2^-53 (modarith.h):
  // 0.1875 for q < 2^50, 0.0235 for q < 2^47
  const double unit_err = 1.5 * qd / two53;
  std::vector<u64> in(n), ref(n), x(n);
  // ---------------- forward
  for (auto& v : in) v = rnd() % (in_mf_f * q);
  in[0] = in_mf_f * q - 1;
  in[1] = 0;
  in[n - 1] = in_mf_f * q - 1;
  ho_ntt_forward_radix2(ref.data(), in.data(), n, q, R.data(), Rp.data(), in_mf_f, 1);
  for (u64 i = 0; i < n; ++i) {
    x[i] = to_internal<Fp64>(in[i], m);
    EXPECT(std::fabs(fp_bits_to_double(x[i])) <= 0.5000001 * qd, "fp entry bound");
  }
  int s = 0;
  for (size_t ri = 0; ri < fwd_runs.size() && s < L; ++ri) {
    double bound = 0.5000001;  // in units of q
    for (int tt = 0; tt < fwd_runs[ri] && s < L; ++tt, ++s) {
      EXPECT(tt < max_fwd_run, "forward run longer than the policy's kFwdRun");
      bound = (1.0 + unit_err) * bound + 0.5;
      EXPECT(bound * qd < two53, "fp fwd: the bound itself passes 2^53 (stage %d)", s);
      const u64 mgroups = 1ull << s, t = n >> (s + 1);
      for (u64 i = 0; i < mgroups; ++i)
        for (u64 j = 0; j < t; ++j) {
          u64& a = x[2 * i * t + j];
          u64& b = x[2 * i * t + j + t];
          fwd_butterfly_fp(a, b, W[mgroups + i], m);
          const double av = std::fabs(fp_bits_to_double(a)), bv = std::fabs(fp_bits_to_double(b));
          EXPECT(av < two53 && bv < two53, "fp fwd 2^53: stage %d", s);
          EXPECT(av <= bound * qd && bv <= bound * qd, "fp fwd bound: stage %d run pos %d", s, tt);
        }
    }
    if (s < L)
      for (u64 i = 0; i < n; ++i) x[i] = fp_bound<Fp64>(x[i], m);
  }
  EXPECT(s == L, "forward runs do not cover the network");
  for (u64 i = 0; i < n; ++i) {
    EXPECT(fwd_finish<Fp64>(x[i], m, true) == ref[i], "fp fwd mismatch at %llu",
           (unsigned long long)i);
    EXPECT(fp_pass_end(fp_bits_to_double(x[i]), m, true) == ref[i], "fp pass end (canonical)");
    const double r = fp_bits_to_double(fp_pass_end(fp_bits_to_double(x[i]), m, false));
    EXPECT(std::fabs(r) <= 0.5000001 * qd, "fp pass end (reduced)");
  }
  // ---------------- inverse
  for (auto& v : in) v = rnd() % (in_mf_i * q);
  in[0] = in_mf_i * q - 1;
  in[1] = in_mf_i * q - 1;
  in[2] = 0;
  ho_ntt_inverse_radix2(ref.data(), in.data(), n, q, Ri_stage.data(), Rip_stage.data(), in_mf_i, 1);
  for (u64 i = 0; i < n; ++i) x[i] = to_internal<Fp64>(in[i], m);
  const u64 n1 = ho_inverse_mod(n, q);
  const double n1d = balanced(n1), n1wd = balanced(ho_multiply_mod(n1, ho_inverse_mod(R[1], q), q));
  int stage = L - 1;
  for (size_t ri = 0; ri < inv_runs.size() && stage >= 0; ++ri) {
    double bound = 0.5000001;  // every element at the start of a run
    for (int tt = 0; tt < inv_runs[ri] && stage >= 0; ++tt, --stage) {
      EXPECT(tt < max_inv_run, "inverse run longer than the policy's kInvRun");
      // sums double; a product of a difference below 2 B q comes out below (0.5 + unit_err 2 B) q
      const double diff = 2.0 * bound;
      EXPECT(diff * qd < two53, "fp inv: differences pass 2^53 (stage %d)", stage);
      bound = std::max(2.0 * bound, 0.5 + unit_err * diff);
      const u64 mgroups = 1ull << stage, t = n >> (stage + 1);
      for (u64 i = 0; i < mgroups; ++i)
        for (u64 j = 0; j < t; ++j) {
          const u64 ia = 2 * i * t + j, ib = ia + t;
          if (stage == 0)
            inv_butterfly_last_fp(x[ia], x[ib], n1d, n1wd, m);
          else
            inv_butterfly_fp(x[ia], x[ib], V[mgroups + i], m);
          const double av = std::fabs(fp_bits_to_double(x[ia])),
                       bv = std::fabs(fp_bits_to_double(x[ib]));
          EXPECT(av < two53 && bv < two53, "fp inv 2^53: stage %d", stage);
          EXPECT(av <= bound * 1.0001 * qd && bv <= bound * 1.0001 * qd, "fp inv bound: stage %d", stage);
        }
    }
    if (stage >= 0)
      for (u64 i = 0; i < n; ++i) x[i] = fp_bound<Fp64>(x[i], m);
  }
  EXPECT(stage == -1, "inverse runs do not cover the network");
  for (u64 i = 0; i < n; ++i)
    EXPECT(inv_finish<Fp64>(x[i], m, true) == ref[i], "fp inv mismatch at %llu",
           (unsigned long long)i);
}

// The Fp64 product and reductions on their own at the edges of their domain.
static void check_fp_product(u64 q) {
  const ModConst m = make_mod_const(q);
  const double qd = (double)q;
  for (int it = 0; it < 400000; ++it) {
    // |y| up to just below 8q (the largest input a kFpFwdRun-stage run feeds a product)
    u64 mag;
    switch (it & 7) {
      case 0: mag = 8 * q - 1 - rnd() % 8; break;
      case 1: mag = rnd() % 8; break;
      case 2: mag = (rnd() % 8) * q + (rnd() % 4); break;
      case 3: mag = (1 + rnd() % 8) * q - 1 - (rnd() % 4); break;
      default: mag = rnd() % (8 * q); break;
    }
    const bool neg = (it >> 3) & 1;
    u64 W;
    switch ((it >> 4) & 3) {
      case 0: W = q - 1 - rnd() % 4; break;
      case 1: W = 1 + rnd() % 4; break;
      case 2: W = q / 2 + (rnd() % 5) - 2; break;
      default: W = rnd() % q; break;
    }
    const double Wb = W > q / 2 ? -(double)(q - W) : (double)W;
    const double y = neg ? -(double)mag : (double)mag;
    const double t = fp_mul(y, Wb, m);
    // exact residue: (+-mag) * W mod q
    u64 want = (u64)(((unsigned __int128)(mag % q) * W) % q);
    if (neg && want) want = q - want;
    EXPECT(t == std::floor(t) && std::fabs(t) <= (0.5 + 0.1875 * 8) * qd, "fp product range");
    EXPECT(fp_canonical(t, m) == want, "fp product residue");
    EXPECT(fp_canonical(y, m) == (neg && mag % q ? q - mag % q : mag % q), "fp canonical");
    if (mag < (1ull << 52)) EXPECT(fp_from_u64(mag) == (double)mag, "fp_from_u64");
  }
}

// The lazy product on its own, at the edges of its domain: D even, D < 2^63.
static void check_product(u64 q) {
  const ModConst m = make_mod_const(q);
  u64 worst = 0;
  for (int it = 0; it < 400000; ++it) {
    u64 D, W;
    switch (it & 7) {
      case 0: D = (1ull << 63) - 2; break;
      case 1: D = ((1ull << 63) - 2) - 2 * (rnd() % 1024); break;
      case 2: D = 2 * (rnd() % 1024); break;
      case 3: D = (rnd() | 0xFFFFFFFFull) & ((1ull << 63) - 2); break;   // low word all ones
      case 4: D = (rnd() & ~0xFFFFFFFFull) & ((1ull << 63) - 2); break;  // low word zero
      default: D = rnd() & ((1ull << 63) - 2); break;
    }
    switch ((it >> 3) & 3) {
      case 0: W = q - 1 - rnd() % 4; break;
      case 1: W = 1 + rnd() % 4; break;
      default: W = rnd() % q; break;
    }
    const u64 W63 = ho_multiply_factor(W, 63, q);
    const u64 x = D >> 1;
    const u64 want = (u64)(((unsigned __int128)x * W) % q);
    const u64 acc = rnd();
    const u64 t_apx = mul_add_lazy2<false>(acc, D, W, W63, m.neg_two_q) - acc;
    const u64 t_ex = mul_add_lazy2<true>(0, D, W, W63, m.neg_two_q);
    EXPECT(t_apx < 6 * q && (t_apx & 1) == 0 && (t_apx >> 1) % q == want, "approximate product");
    EXPECT(t_ex < 4 * q && (t_ex & 1) == 0 && (t_ex >> 1) % q == want, "exact product");
    if (t_apx > worst) worst = t_apx;
  }
  // forward finish over its whole input range: doubled values below 128q (the quotient
  // estimate shifts the high word: the Lazy policy serves q >= 2^32 only, choose_policy)
  for (int it = 0; it < 400000 && q >= (1ull << 32); ++it) {
    u64 x = rnd() % (64 * q);
    if ((it & 15) == 0) x = 64 * q - 1 - (rnd() & 7);
    if ((it & 15) == 1) x = (rnd() & 63) * q + ((it & 16) ? 0 : q - 1);
    EXPECT(fwd_finish<Lazy>(x << 1, m, true) == x % q, "fwd_finish canonical");
    const u64 r = fwd_finish<Lazy>(x << 1, m, false);
    EXPECT(r < 2 * q && r % q == x % q, "fwd_finish lazy");
  }
  (void)worst;
}

int main() {
  u64 primes[8];
  for (int bits : {3, 9, 16, 31, 32, 33, 48, 54, 55}) {
    const size_t got = ho_generate_primes(primes, 2, bits, 1, 2);
    for (size_t pi = 0; pi < got; ++pi) check_product(primes[pi]);
    const size_t got2 = ho_generate_primes(primes, 1, bits, 0, 2);
    for (size_t pi = 0; pi < got2; ++pi) check_product(primes[pi]);
  }
  struct Case {
    u64 n;
    int bits;
  };
  // moduli at both ends of the lazy range and beyond it (strict only)
  const Case lazy_cases[] = {{16, 10}, {64, 20}, {1024, 30}, {4096, 31}, {4096, 32}, {4096, 33},
                             {4096, 48}, {8192, 54}, {65536, 54}, {65536, 55}, {131072, 55}};
  const std::vector<std::vector<int>> run_sets = {
      {3, 3, 3, 3, 4, 4},     // tile rounds + strided 4 (+4)
      {3, 3, 3, 3, 5},        // 17 = 12 + 5: five-stage strided subtree
      {1, 3, 3, 3, 3, 3, 3},  // short first round
      {2, 3, 3, 3, 2, 4, 4},
      {3, 3, 3, 3, 3, 3}};
  for (const Case& c : lazy_cases) {
    const size_t got = ho_generate_primes(primes, 2, c.bits, 1, c.n);
    for (size_t pi = 0; pi < got; ++pi)
      for (const auto& runs : run_sets) {
        if (primes[pi] >= (1ull << 32)) {  // the Lazy policy's range
          check<Lazy>(c.n, primes[pi], runs, 4, 2);
          check<Lazy>(c.n, primes[pi], runs, 1, 1);
        } else {  // with the Fp64 policy switched off these moduli take Harvey60
          check<Harvey60>(c.n, primes[pi], runs, 4, 2);
        }
        check<Strict>(c.n, primes[pi], runs, 4, 2);
      }
  }
  // Lazy inverse: every degree's plan, moduli at both ends of the range, both input factors
  for (int L = 1; L <= 17; ++L) {
    for (int bits : {32, 44, 54, 55}) {
      size_t got = ho_generate_primes(primes, 1, bits, 1, 1ull << L);
      got += ho_generate_primes(primes + got, 1, bits, 0, 1ull << L);  // walking down from 2^(bits+1)
      for (size_t pi = 0; pi < got; ++pi) {
        if (primes[pi] < (1ull << 32) || primes[pi] >= (1ull << 56)) continue;
        if (L >= 15 && bits != 55 && pi) continue;  // (keep the run time down)
        check_lazy_inverse(1ull << L, primes[pi], library_passes(L), 2);
        check_lazy_inverse(1ull << L, primes[pi], library_passes(L), 1);
        if (L == 14) check_lazy_inverse(1ull << L, primes[pi], library_passes(L, false), 2);
      }
    }
  }
  // The bounded members of the Lazy family: 2^56 <= q < 2^58 (limit 32), 2^58 <= q < 2^59
  // (limit 16), primes at both ends of each range; forward through check<>, inverse through the
  // library's plans
  for (int L : {1, 3, 6, 10, 11, 12, 13, 14, 15, 16, 17}) {
    for (int bits : {56, 57, 58}) {
      size_t got = ho_generate_primes(primes, 1, bits, 1, 1ull << L);       // just above 2^bits
      got += ho_generate_primes(primes + got, 1, bits, 0, 1ull << L);      // just below 2^(bits+1)
      for (size_t pi = 0; pi < got; ++pi) {
        const u64 q = primes[pi];
        if (q < (1ull << 56) || q >= (1ull << 59)) continue;
        const int limit = q < (1ull << 58) ? 32 : 16;
        if (L >= 15 && pi) continue;
        check_lazy_inverse(1ull << L, q, library_passes(L), 2, limit);
        check_lazy_inverse(1ull << L, q, library_passes(L), 1, limit);
        if (L <= 13 || (L == 16 && pi == 0)) {
          if (limit == 32) {
            check<Lazy32>(1ull << L, q, run_sets[0], 4, 2);
            check<Lazy32>(1ull << L, q, run_sets[0], 1, 1);
          } else {
            check<Lazy16>(1ull << L, q, run_sets[0], 4, 2);
            check<Lazy16>(1ull << L, q, run_sets[0], 1, 1);
          }
        }
      }
    }
  }
  {  // the three-pass and big-tile plans in small: same pass shapes on a 2^18 / 2^20 network are
     // too slow here; their round and pass shapes are covered by these
    const size_t got = ho_generate_primes(primes, 1, 55, 0, 1ull << 16);
    if (got) {
      check_lazy_inverse(1ull << 16, primes[0], {{12, true}, {4, false}}, 2);
      check_lazy_inverse(1ull << 16, primes[0], {{13, true}, {3, false}}, 2);
      check_lazy_inverse(1ull << 15, primes[0], {{14, true}, {1, false}}, 2);
      check_lazy_inverse(1ull << 16, primes[0], {{10, true}, {2, false}, {4, false}}, 2);
      check_lazy_inverse(1ull << 14, primes[0], {{9, true}, {5, false}}, 2);
    }
  }
  // Small policy: q < 2^30, up to the bound (GeneratePrimes(., 29, false, .) walks down from 2^30)
  {
    const Case small_cases[] = {{16, 10}, {1024, 20}, {4096, 28}, {65536, 29}};
    for (const Case& c : small_cases) {
      size_t got = ho_generate_primes(primes, 2, c.bits, 1, c.n);
      got += ho_generate_primes(primes + got, 2, c.bits, 0, c.n);
      for (size_t pi = 0; pi < got; ++pi) {
        check<Small>(c.n, primes[pi], run_sets[0], 4, 2);
        check<Small>(c.n, primes[pi], run_sets[1], 1, 1);
      }
    }
  }
  // the largest primes below 2^56 (GeneratePrimes(., 55, false, .) walks downwards)
  {
    const size_t got = ho_generate_primes(primes, 2, 55, 0, 65536);
    for (size_t pi = 0; pi < got; ++pi) {
      check<Lazy>(65536, primes[pi], run_sets[0], 4, 2);
      check<Lazy>(65536, primes[pi], run_sets[1], 4, 2);
    }
  }
  {
    const size_t got = ho_generate_primes(primes, 2, 60, 0, 4096);
    for (size_t pi = 0; pi < got; ++pi) check<Strict>(4096, primes[pi], run_sets[0], 4, 2);
    const size_t got2 = ho_generate_primes(primes, 1, 61, 1, 4096);
    for (size_t pi = 0; pi < got2; ++pi) check<Strict>(4096, primes[pi], run_sets[0], 4, 2);
  }
  // Harvey60 policy: 2^56 <= q < 2^60 + 2^28, both ends of the range, SEAL-style 60-bit primes
  {
    // ({., 60}: only the primes just above 2^60 -- upwards from 2^60 -- are inside the range)
    const Case h_cases[] = {{16, 56}, {4096, 56}, {4096, 57}, {8192, 58}, {65536, 59}, {131072, 59},
                            {4096, 60}, {131072, 60}};
    for (const Case& c : h_cases) {
      size_t got = ho_generate_primes(primes, 2, c.bits, 1, c.n);
      if (c.bits < 60) got += ho_generate_primes(primes + got, 2, c.bits, 0, c.n);  // down from 2^(bits+1)
      for (size_t pi = 0; pi < got; ++pi) {
        check<Harvey60>(c.n, primes[pi], run_sets[0], 4, 2);
        check<Harvey60>(c.n, primes[pi], run_sets[1], 1, 1);
        check<Harvey60>(c.n, primes[pi], run_sets[3], 2, 2);
      }
    }
  }
  // Fp64 policy: moduli from 2^30 up to just below 2^50, including the survey's 50-bit
  // prime of BASELINE configs[1]; run layouts of the kernels (tile rounds 2|3|3|3 with one
  // mid-pass reduction, strided 4 / 5, the longest legal runs)
  {
    // (runs past the end of a network are ignored)
    const std::vector<std::vector<int>> fwd_sets = {{5, 5, 6, 7}, {4, 5, 6, 5}, {7, 7, 6}, {3, 3, 3, 3, 3, 3, 3},
                                                    {5, 7, 5, 3}, {1, 7, 7, 5}};
    const std::vector<std::vector<int>> inv_sets = {{3, 3, 3, 3, 3, 3, 2}, {2, 3, 3, 3, 3, 3, 3},
                                                    {3, 3, 3, 3, 3, 2, 3}, {1, 3, 3, 3, 3, 3, 3, 1}};
    const Case fp_cases[] = {{16, 30}, {1024, 35}, {4096, 49}, {4096, 45}, {65536, 49}, {131072, 49}};
    for (const Case& c : fp_cases) {
      size_t got = ho_generate_primes(primes, 2, c.bits, 1, c.n);
      got += ho_generate_primes(primes + got, 2, c.bits, 0, c.n);  // walking down from 2^(bits+1)
      for (size_t pi = 0; pi < got; ++pi) {
        check_fp_product(primes[pi]);
        for (size_t k = 0; k < fwd_sets.size(); ++k) {
          check_fp(c.n, primes[pi], fwd_sets[k], inv_sets[k % inv_sets.size()], 4, 2);
          check_fp(c.n, primes[pi], fwd_sets[k], inv_sets[(k + 1) % inv_sets.size()], 1, 1);
        }
      }
    }
    check_fp_product(562949954093057ull);
    check_fp(4096, 562949954093057ull, {3, 3, 6}, {3, 3, 3, 3}, 1, 1);
    // Fp64L (q < 2^47: forward passes without a reduction between first load and last store,
    // inverse runs of up to 6 stages), the run layouts of the kernels: one tile pass of the whole
    // network (N <= 2^14: rounds NR-1 .. 0, reduced where two more rounds would not fit), strided
    // + tile passes above
    const Case fpl_cases[] = {{16, 30}, {1024, 35}, {4096, 36}, {4096, 43}, {8192, 43}, {8192, 46},
                              {16384, 46}, {65536, 44}, {131072, 46}};
    for (const Case& c : fpl_cases) {
      int L = 0;
      while ((1ull << L) < c.n) ++L;
      size_t got = ho_generate_primes(primes, 1, c.bits, 1, c.n);
      got += ho_generate_primes(primes + got, 1, c.bits, 0, c.n);  // just below 2^(bits+1)
      for (size_t pi = 0; pi < got; ++pi) {
        if (primes[pi] >= (1ull << 47)) continue;
        check_fp_product(primes[pi]);
        std::vector<int> fwd, inv;
        if (L <= 14) {
          fwd = {L};
          // rounds of the tile pass in inverse order, merged into runs of <= 6 stages
          const int re = re_of(L), rounds = (L + re - 1) / re, r0 = L - (rounds - 1) * re;
          int c_run = 0;
          for (int j = rounds - 1; j >= 0; --j) {
            const int r = j == 0 ? r0 : re, rn = j >= 1 ? (j - 1 == 0 ? r0 : re) : 0;
            c_run += r;
            if (j == 0 || c_run + rn > 6) {
              inv.push_back(c_run);
              c_run = 0;
            }
          }
        } else {
          const int bottom = L <= 16 ? 11 : 12;
          fwd = {L - bottom, bottom};
          inv = bottom == 11 ? std::vector<int>{6, 5, L - bottom} : std::vector<int>{6, 6, L - bottom};
        }
        check_fp(c.n, primes[pi], fwd, inv, 4, 2, Fp64L::kFwdRun, Fp64L::kInvRun);
        check_fp(c.n, primes[pi], fwd, inv, 1, 1, Fp64L::kFwdRun, Fp64L::kInvRun);
      }
    }
  }
  if (g_fail) {
    fprintf(stderr, "host_arith_check: %d failures\n", g_fail);
    return 1;
  }
  printf("host_arith_check OK (%d network replays)\n", g_cases);
  return 0;
}
