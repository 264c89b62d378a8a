// internal.h -- declarations shared by the kernel translation units and the
// C-ABI layer (capi.cpp).  Not installed.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>

#include "modarith.h"

namespace hexl_amd {

// Scalars of the last inverse stage: n1 = N^-1 mod q, n1w = N^-1 * R[1]^-1,
// each with its floor(x * 2^64 / q) companion
// (hexl/ntt/ntt-radix-2.cpp:490-497).
struct InvLast {
  u64 n1, n1p, n1w, n1wp;
};

// Moduli below this bound use the Lazy arithmetic policy (modarith.h); their
// device tables carry 63-bit Shoup factors.
constexpr u64 kLazyModulusBound = 1ull << 56;
// Moduli below this bound use the Small policy (32-bit arithmetic); their device
// tables carry 32-bit Shoup factors.
constexpr u64 kSmallModulusBound = 1ull << 30;

// Moduli in [kSmallModulusBound, kFp64ModulusBound) use the Fp64 policy (exact integers in
// doubles, one-word balanced twiddles).
constexpr u64 kFp64ModulusBound = 1ull << 50;

enum ArithPolicy : int { kPolicySmall = 0, kPolicyFp64 = 1, kPolicyLazy = 2, kPolicyStrict = 3 };
int choose_policy(u64 q);  // ntt_kernels.hip

// Device-resident state of one NTT plan.
struct NttTables {
  // heap-ordered twiddles.  Integer policies: (R[n], floor(R[n] 2^s / q)) pairs, s = 32
  // (Small), 63 (Lazy) or 64 (Strict).  Fp64: an array of doubles, R[n] balanced into
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
constexpr int kMaxMultiPlans = 32;

// One transform over num_plans * polys_per_plan polynomials, polynomial b using plan
// b / polys_per_plan (tabs[k]->dev must be set; all plans: same degree >= 4096, same
// arithmetic policy, num_plans <= kMaxMultiPlans -- hipErrorNotSupported otherwise, and
// the caller loops over the plans instead).
hipError_t ntt_multi_launch(bool forward, const struct NttTables* const* tabs, u32 num_plans,
                            u64 polys_per_plan, u64* result, const u64* operand, u64 out_mf,
                            hipStream_t st);

hipError_t ntt_forward_launch(const NttTables& t, u64* result, const u64* operand, u64 batch,
                              u64 out_mf, hipStream_t st);
hipError_t ntt_inverse_launch(const NttTables& t, u64* result, const u64* operand, u64 batch,
                              u64 out_mf, hipStream_t st);

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
hipError_t dyadic_multiply_launch(u64* result, const u64* op1, const u64* op2, u64 n,
                                  const u64* moduli_host, u64 num_moduli, hipStream_t st);
// ---- KeySwitch stages (keyswitch_kernels.hip); argument blocks travel as kernel
// arguments, so the number of decomposition moduli per call is bounded.
constexpr int kKsMaxDecomp = 32;
struct KsGather {
  u64 q, barrett;          // key modulus of this RNS index, floor(2^64 / q)
  u32 jmap[kKsMaxDecomp];  // slot -> decomposition modulus
  u32 reduce_mask;         // bit s: moduli[jmap[s]] > q, reduce
};
struct KsMac {
  const u64* keys[kKsMaxDecomp];
  u32 slot[kKsMaxDecomp];  // decomposition modulus -> slot of ntt_buf (unused for `self`)
  u32 decomp, self;        // self == decomp: no operand is taken from t_target_iter
  u64 key_component_stride, key_index_offset;
  u64 prod_component_stride, prod_offset;
  u64 q, barrett, two64_mod_q, mu;  // mu, shift: generalised Barrett of MultOp
  u32 shift;
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
hipError_t ks_gather_launch(u64* out, const u64* t_target, u64 n, u32 slots, const KsGather& g,
                            hipStream_t st);
hipError_t ks_mac_launch(u64* prod, const u64* t_target_iter, const u64* ntt_buf, u64 n,
                         u32 components, const KsMac& m, hipStream_t st);
hipError_t ks_round_launch(u64* tbuf, const u64* t_last, u64 n, u32 decomp, const KsRound& r,
                           hipStream_t st);
hipError_t ks_finish_launch(u64* result, const u64* prod, const u64* tbuf, u64 n, u32 decomp,
                            const KsFinish& f, hipStream_t st);

hipError_t fill_splitmix_launch(u64* data, u64 n, u64 batch, u64 seed0, u64 bound,
                                hipStream_t st);

}  // namespace hexl_amd
