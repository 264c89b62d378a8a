// internal.h -- declarations shared by the kernel translation units and the
// C-ABI layer (capi.cpp).  Not installed.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>

#include "modarith.h"

namespace hexl_amd {

// Moduli below this bound use the Lazy arithmetic policy (modarith.h); their
// device tables carry 63-bit Shoup factors.
constexpr u64 kLazyModulusBound = 1ull << 56;
// The bounded members of the Lazy family (modarith.h: LazyT): floor(2^63 / q) >= 32 below 2^58,
// >= 16 below 2^59.  From 2^59 on -- where SEAL's and OpenFHE's 60-bit primes sit -- no stage
// can go without a conditional subtraction and Harvey60 is the cheapest.
constexpr u64 kLazy32ModulusBound = 1ull << 58;
constexpr u64 kLazy16ModulusBound = 1ull << 59;
// Moduli below this bound use the Small policy (32-bit arithmetic); their device
// tables carry 32-bit Shoup factors.
constexpr u64 kSmallModulusBound = 1ull << 30;

// Moduli in [kSmallModulusBound, kFp64ModulusBound) use the Fp64 policy (exact integers in
// doubles, one-word balanced twiddles).
constexpr u64 kFp64ModulusBound = 1ull << 50;
// Below this bound the Fp64 arithmetic has 2^53 / q >= 64 of headroom and runs of stages between
// full reductions can be twice as long (modarith.h: Fp64L).
constexpr u64 kFp64LongModulusBound = 1ull << 47;

// Moduli in [kLazy16ModulusBound, kHarvey60ModulusBound) -- from kLazyModulusBound with the
// bounded Lazy members switched off -- use the Harvey60 policy (Harvey ranges on doubled values;
// 63-bit Shoup factors like Lazy).  Its products need the high word of
// a doubled value (< 8q) to stay <= 2^31, i.e. 8q <= 2^63 + 2^31: the bound sits 2^28 above
// 2^60, which takes in the smallest primes above 2^60 -- what GeneratePrimes(., 60, true, .)
// returns, the reference's "61-bit" test and benchmark moduli (BASELINE configs[4]).
constexpr u64 kHarvey60ModulusBound = (1ull << 60) + (1ull << 28);

enum ArithPolicy : int {
  kPolicySmall = 0,
  kPolicyFp64 = 1,
  kPolicyLazy = 2,
  kPolicyStrict = 3,
  kPolicyHarvey60 = 4,
  kPolicyLazy32 = 5,
  kPolicyLazy16 = 6,
  kPolicyFp64L = 7,
  kNumPolicies = 8
};
int choose_policy(u64 q);  // ntt_kernels.hip

// Device-resident state of one NTT plan.
struct NttTables {
  // heap-ordered twiddles.  Integer policies: (R[n], floor(R[n] 2^s / q)) pairs, s = 32
  // (Small), 63 (Lazy, Harvey60) or 64 (Strict).  Fp64: an array of doubles, R[n] balanced into
  // (-q/2, q/2] (the pointer type is nominal).
  const ulonglong2* fwd;
  const ulonglong2* inv;  // the same for R[n]^-1
  ModConst mod;
  u32 log_n;
  int policy;  // ArithPolicy the tables were built for
  InvLast inv_last;  // Fp64: n1 / n1w hold the bit patterns of the balanced doubles
  const struct PlanDev* dev;  // device copy of (fwd, inv, mod, inv_last)
};

// Optional per-kernel timing (bench support): when a sink is active on the
// calling thread every kernel launch is bracketed by a pair of hipEvents
// recorded on the launch stream.
struct ProfileRecord {
  const char* name;
  hipEvent_t start, stop;
};
struct ProfileSink {
  ProfileRecord* records;
  int capacity;
  int count;
};
extern thread_local ProfileSink* g_profile;

struct ScopedKernelTimer {
  ProfileRecord* r = nullptr;
  hipStream_t st;
  ScopedKernelTimer(const char* name, hipStream_t stream) : st(stream) {
    ProfileSink* s = g_profile;
    if (s && s->count < s->capacity) {
      r = &s->records[s->count++];
      r->name = name;
      (void)hipEventRecord(r->start, st);
    }
  }
  ~ScopedKernelTimer() {
    if (r) (void)hipEventRecord(r->stop, st);
  }
};

// Device-resident copy of a plan's kernel parameters: what the multi-plan launches
// (one launch over polynomials of several moduli) read per workgroup instead of
// kernel arguments.
struct PlanDev {
  const ulonglong2* fwd;
  const ulonglong2* inv;
  ModConst mod;
  InvLast il;
};
constexpr int kMaxMultiPlans = 40;
constexpr int kMaxMultiPeriod = 1024;

// Which plan a polynomial of a multi-plan launch uses:
//   plan = plan_tab[(polynomial / inner) % period]
// (RNS limbs: inner = polynomials per modulus, period = number of moduli, identity table;
// KeySwitch: inner = 1, period = polynomials per target, table = modulus of each).
// Optionally the INPUT of a polynomial lives elsewhere (KeySwitch: the D^2 product operands
// of a target are its D coefficient-form polynomials, each reduced to the key modulus where
// that is smaller -- key-switch-internal.cpp:77-89 -- and the first pass of the transform
// reads them straight from there instead of from a gathered copy): with src_stride != 0
// (inner must be 1) polynomial p reads
//   operand[(p / period) * src_stride + (src_tab[p % period] & 0x7f)]
// and reduces every word modulo its own modulus first when bit 7 of the entry is set.
// With rnd_qk != 0 (and a source map) the words are rounded on load first (ntt_kernels.hip:
// kRoundFirst): x' = (x + rnd_half) mod rnd_qk, reduced modulo the polynomial's own modulus where
// bit 7 of its entry says so, plus the correction q - (rnd_half mod q) -- the rounding stage of
// KeySwitch riding on the load of the forward transform that consumes it.
struct MultiMap {
  u32 inner, period;
  uint8_t plan_tab[kMaxMultiPeriod];
  u32 src_stride;
  uint8_t src_tab[kMaxMultiPeriod];
  u64 rnd_qk, rnd_barrett, rnd_half;
};

// One transform over `polys` polynomials of several moduli (all plans: same degree in
// [2^12, 2^17]; tabs[k]->dev set; num_plans <= kMaxMultiPlans).  Plans of different
// arithmetic policies are served by one launch sequence per policy.  Returns
// hipErrorNotSupported when the shapes do not fit; the caller then loops over plans.
// epi (forward only): the last pass folds its output into a KeySwitch result instead of storing
// it (KsEpilogue below); `result` then only holds what earlier passes hand over.
struct KsEpilogue;
hipError_t ntt_multi_launch(bool forward, const struct NttTables* const* tabs, u32 num_plans,
                            const MultiMap& map, u64 polys, u64* result, const u64* operand,
                            u64 out_mf, hipStream_t st, const KsEpilogue* epi = nullptr);

// `mid` (optional, batch polynomials of device memory): where the first pass of a two-pass plan hands
// over to the second instead of `result` -- for result / operand in host memory across the link.
hipError_t ntt_forward_launch(const NttTables& t, u64* result, const u64* operand, u64 batch,
                              u64 out_mf, hipStream_t st, u64* mid = nullptr);
hipError_t ntt_inverse_launch(const NttTables& t, u64* result, const u64* operand, u64 batch,
                              u64 out_mf, hipStream_t st, u64* mid = nullptr);

// Whether a transform of `batch` polynomials under plan `t` is ONE kernel launch (degrees up
// to 2^12, 2^13, and 2^14 from 96 polynomials): what the zero-copy host path can run straight
// on caller memory.
// (`link`: the buffers are host memory across the link and a `mid` buffer will be passed if the plan has
// two passes -- the plan may differ: see link_allows_tile13)
bool ntt_is_single_kernel(const NttTables& t, u64 batch, bool link = false);
// ... or two (one strided pass + the tile pass: N = 2^15 ... 2^19, N = 2^14 below 96 polynomials): the
// shape for which ntt_forward_launch / ntt_inverse_launch use `mid`.
bool ntt_is_two_pass(const NttTables& t, u64 batch, bool link = false);

// Tuning / diagnostic knobs of the NTT launch logic (ntt_kernels.hip); 0 on success.
int set_tuning(const char* key, u64 value);

// Element-wise launchers (eltwise_kernels.hip)
enum EltOp {
  ELT_ADD = 0,
  ELT_ADD_SCALAR = 1,
  ELT_SUB = 2,
  ELT_SUB_SCALAR = 3,
  ELT_MULT = 4,
  ELT_FMA = 5,
  ELT_REDUCE = 6,
  ELT_REDUCE_FMA = 7,
  ELT_CMP_ADD = 8,
  ELT_CMP_SUB_MOD = 9
};

struct EltArgs {
  u64* result;
  const u64* a;
  const u64* b;  // second vector (add/sub/mult) or addend (fma); may be null
  u64 scalar;    // scalar operand (add/sub scalar forms, fma multiplier)
  u64 n;
  u64 q;
  u64 in_mf;
  u64 out_mf;
  int cmp = 0;    // CMPINT of the cmp_* ops (their `diff` travels in `scalar`)
  u64 bound = 0;
};

hipError_t eltwise_launch(EltOp op, const EltArgs& args, hipStream_t st);
// `pairs` ciphertext pairs back to back (operands 2, results 3 polynomials of n * num_moduli)
hipError_t dyadic_multiply_launch(u64* result, const u64* op1, const u64* op2, u64 n,
                                  const u64* moduli_host, u64 num_moduli, u64 pairs,
                                  hipStream_t st);
// ---- KeySwitch stages (keyswitch_kernels.hip), batched over `targets` ciphertexts that
// share keys and moduli.  Argument blocks travel as kernel arguments, so the number of
// decomposition moduli per call is bounded.  Buffer layouts (polynomials of n words):
//   t_target [target][j]            coefficient form of the target per decomposition modulus
//   ntt_buf  [target][s], s < D^2   operand of product (i, j != i): s = i (D-1) + (j < i ? j : j-1)
//                                   for RNS index i < D, s = D (D-1) + j for the extra index D
//   prod     [i][target][k]         accumulated products per RNS index i <= D, key component k
//   tbuf     [target][k][i]         the rounded last component brought to modulus i < D
//   result   [target][k][i]         (the caller's: `targets` results of the reference's layout)
constexpr int kKsMaxDecomp = 32;
struct KsDims {
  u64 n;
  u32 decomp, targets, components, key_moduli;  // D, T, C, K
};
struct KsGatherAll {  // per RNS index i <= D: its key modulus
  u64 q[kKsMaxDecomp + 1], barrett[kKsMaxDecomp + 1];
  u32 reduce_mask[kKsMaxDecomp + 1];  // bit j: moduli[j] > q[i], reduce operand j
};
struct KsMacAll {
  const u64* keys[kKsMaxDecomp];
  u64 q[kKsMaxDecomp + 1], barrett[kKsMaxDecomp + 1], two64_mod_q[kKsMaxDecomp + 1],
      mu[kKsMaxDecomp + 1];  // mu, shift: generalised Barrett of MultOp
  u32 shift[kKsMaxDecomp + 1], key_index[kKsMaxDecomp + 1];
  // a 128-bit sum whose high word is below this takes ONE generalised Barrett step (ks_mac_kernel); 0: never
  u64 hi_limit[kKsMaxDecomp + 1];
};
struct KsRoundMod {
  u64 q, barrett, fix;
  u32 reduce;
};
struct KsRound {
  u64 qk, barrett_k, qk_half;
  KsRoundMod mod[kKsMaxDecomp];
};
struct KsFinishMod {
  u64 q, s, sp;
};
struct KsFinish {
  KsFinishMod mod[kKsMaxDecomp];
};
// The finish stage of KeySwitch (key-switch-internal.cpp:180-196) riding on the store of the last
// forward transform (ntt_kernels.hip: fwd_copy_out_finish; round 6): polynomial p of the launch is
// (target * C + k) * D + i; its transformed words t are not stored but folded into
//   result[p][l] = (result[p][l] + ((prod[i][target * C + k][l] - t) mod q_i) * s_i) mod q_i .
struct KsEpilogue {
  u64* result;
  const u64* prod;
  u32 decomp;  // D
  u32 tc;      // targets * key components
  u64 s[kKsMaxDecomp], sp[kKsMaxDecomp];  // modswitch factor of modulus i in [0, q_i), Shoup companion
};

hipError_t ks_gather_launch(u64* ntt_buf, const u64* t_target, const KsDims& d,
                            const KsGatherAll& g, hipStream_t st);
hipError_t ks_mac_launch(u64* prod, const u64* t_target_iter, const u64* ntt_buf, const KsDims& d,
                         const KsMacAll& m, hipStream_t st);
hipError_t ks_round_launch(u64* tbuf, const u64* prod, const KsDims& d, const KsRound& r,
                           hipStream_t st);
hipError_t ks_finish_launch(u64* result, const u64* prod, const u64* tbuf, const KsDims& d,
                            const KsFinish& f, hipStream_t st);

// *violations += number of words of data[0, n) that are >= bound (debug contract)
hipError_t count_out_of_bounds_launch(const u64* data, u64 n, u64 bound,
                                      unsigned long long* violations, hipStream_t st);

hipError_t completion_flag_launch(u32* flag, u32 seq, hipStream_t st);

hipError_t fill_splitmix_launch(u64* data, u64 n, u64 batch, u64 seed0, u64 bound,
                                hipStream_t st);

}  // namespace hexl_amd
