// hexl_shim.cpp -- namespace intel::hexl over the C-ABI of include/hexl_amd.h.
//
// This translation unit is the whole of libhexl.so: it includes no HIP header
// and calls nothing but the functions declared in hexl_amd.h, which is the
// proof that the C-ABI is a sufficient drop-in boundary for the reference's
// NTT / Eltwise* / number-theory API.  Errors reported by the C-ABI (contract
// violations, HIP failures, no GPU) become std::runtime_error -- the reference
// throws the same type from HEXL_CHECK in debug builds
// (hexl/include/hexl/util/check.hpp:19-34) and has undefined behaviour in
// release builds.
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "hexl/hexl.hpp"
#include "hexl_amd.h"

namespace intel {
namespace hexl {

AllocatorStrategyPtr mallocStrategy = AllocatorStrategyPtr(new MallocStrategy);

namespace {

void check(int rc) {
  if (rc != HEXL_AMD_OK) throw std::runtime_error(std::string("hexl: ") + hexl_amd_last_error());
}

// The debug contract.  The reference's debug library (libhexl_debug: HEXL_DEBUG) checks every
// ELEMENT of an operand against its range and throws (HEXL_CHECK_BOUNDS,
// hexl/include/hexl/util/check.hpp:32-35; hexl/ntt/ntt-internal.cpp:198, :261;
// hexl/eltwise/eltwise-mult-mod.cpp:31-33, ...; exercised by test/test-ntt.cpp:20-94).  This
// file compiled with -DHEXL_DEBUG is libhexl_debug.so: the same checks, through
// hexl_amd_check_bounds -- a loop for the caller's host buffers, one reduction kernel for
// device memory.  Scalar arguments (null pointers, mod factors, modulus ranges) are validated
// by the C-ABI in every build.
#ifdef HEXL_DEBUG
void shim_check_bounds(const uint64_t* arg, uint64_t n, uint64_t bound, const char* what,
                       const char* function) {
  if (!arg || n == 0) return;  // (null / empty: reported by the operation itself)
  uint64_t violations = 0;
  check(hexl_amd_check_bounds(arg, n, bound, &violations));
  HEXL_CHECK(violations == 0, violations << " value(s) of " << what << " in " << function
                                         << " exceed bound " << bound);
}
#define HEXL_SHIM_CHECK_BOUNDS(arg, n, bound, what) \
  shim_check_bounds(arg, n, bound, what, __FUNCTION__)
#else
#define HEXL_SHIM_CHECK_BOUNDS(arg, n, bound, what) \
  {}
#endif

using Table = AlignedVector64<uint64_t>;

}  // namespace

// ---------------------------------------------------------------- device-mapped host memory
void* DeviceMappedAllocate(size_t bytes) {
  void* p = nullptr;
  check(hexl_amd_host_alloc(&p, bytes));
  return p;
}
void DeviceMappedFree(void* p) noexcept { (void)hexl_amd_host_free(p); }
void RegisterHostMemory(void* p, size_t bytes) { check(hexl_amd_host_register(p, bytes)); }
void UnregisterHostMemory(void* p) noexcept { (void)hexl_amd_host_unregister(p); }
void* DeviceMalloc(size_t bytes) {
  void* p = nullptr;
  check(hexl_amd_device_alloc(&p, bytes, -1));
  return p;
}
void DeviceFree(void* p) noexcept { (void)hexl_amd_device_free(p); }
void Copy(void* dst, const void* src, size_t bytes) { check(hexl_amd_copy(dst, src, bytes, nullptr, 1)); }
void DeviceSynchronize() { check(hexl_amd_synchronize(nullptr)); }
AllocatorStrategyPtr DeviceMappedStrategy() {
  static AllocatorStrategyPtr s = std::make_shared<DeviceMappedAllocator>();
  return s;
}

// ---------------------------------------------------------------- number theory
uint64_t ReverseBits(uint64_t x, uint64_t bit_width) { return hexl_amd_reverse_bits(x, bit_width); }
uint64_t InverseMod(uint64_t x, uint64_t modulus) {
  HEXL_CHECK(x % modulus != 0, x << " does not have a InverseMod");
  return hexl_amd_inverse_mod(x, modulus);
}
uint64_t MultiplyMod(uint64_t x, uint64_t y, uint64_t modulus) {
  HEXL_CHECK(modulus != 0, "modulus == 0");
  return hexl_amd_multiply_mod(x, y, modulus);
}
uint64_t MultiplyMod(uint64_t x, uint64_t y, uint64_t y_precon, uint64_t modulus) {
  const uint64_t r = x * y - MultiplyUInt64Hi<64>(x, y_precon) * modulus;
  return r >= modulus ? r - modulus : r;
}
uint64_t AddUIntMod(uint64_t x, uint64_t y, uint64_t modulus) {
  const uint64_t s = x + y;
  return s >= modulus ? s - modulus : s;
}
uint64_t SubUIntMod(uint64_t x, uint64_t y, uint64_t modulus) {
  const uint64_t d = x + modulus - y;
  return d >= modulus ? d - modulus : d;
}
uint64_t PowMod(uint64_t base, uint64_t exp, uint64_t modulus) {
  return hexl_amd_pow_mod(base, exp, modulus);
}
bool IsPrimitiveRoot(uint64_t root, uint64_t degree, uint64_t modulus) {
  return hexl_amd_is_primitive_root(root, degree, modulus) != 0;
}
uint64_t GeneratePrimitiveRoot(uint64_t degree, uint64_t modulus) {
  return hexl_amd_generate_primitive_root(degree, modulus);
}
uint64_t MinimalPrimitiveRoot(uint64_t degree, uint64_t modulus) {
  return hexl_amd_minimal_primitive_root(degree, modulus);
}
bool IsPrime(uint64_t n) { return hexl_amd_is_prime(n) != 0; }
std::vector<uint64_t> GeneratePrimes(size_t num_primes, size_t bit_size, bool prefer_small_primes,
                                     size_t ntt_size) {
  std::vector<uint64_t> out(num_primes);
  const size_t found = hexl_amd_generate_primes(out.data(), num_primes, bit_size,
                                                prefer_small_primes ? 1 : 0, ntt_size);
  if (found != num_primes) throw std::runtime_error("hexl: Failed to find enough primes");
  return out;
}

// ---------------------------------------------------------------- NTT
struct NTT::State {
  hexl_amd_ntt* plan = nullptr;
  uint64_t degree = 0, q = 0, w = 0;
  AlignedAllocator<uint64_t, 64> alloc;
  // index: 0..6 as hexl_amd_ntt_table; 7..10 the AVX512-layout variants
  mutable Table tables[11];
  mutable std::once_flag once[11];

  explicit State(std::shared_ptr<AllocatorBase> a)
      : alloc(a),
        tables{Table(alloc), Table(alloc), Table(alloc), Table(alloc), Table(alloc), Table(alloc),
               Table(alloc), Table(alloc), Table(alloc), Table(alloc), Table(alloc)} {}
  ~State() { hexl_amd_ntt_destroy(plan); }

  const Table& base_table(int which) const {
    std::call_once(once[which], [&] {
      const uint64_t* p = hexl_amd_ntt_table(plan, which);
      tables[which].assign(p, p + degree);
    });
    return tables[which];
  }

  // Root powers with entries [N/8, N/4) repeated 4x and [N/4, N/2) repeated 2x
  // (hexl/ntt/ntt-internal.cpp:77-111) and their Barrett factors.
  const Table& avx_table(int idx, uint64_t shift) const {
    std::call_once(once[idx], [&] {
      const Table& R = base_table(0);
      Table v(alloc);
      const size_t n = degree;
      for (size_t i = 0; i < n / 8; ++i) v.push_back(R[i]);
      for (size_t i = n / 8; i < n / 4; ++i)
        for (int k = 0; k < 4; ++k) v.push_back(R[i]);
      for (size_t i = n / 4; i < n / 2; ++i)
        for (int k = 0; k < 2; ++k) v.push_back(R[i]);
      for (size_t i = n / 2; i < n; ++i) v.push_back(R[i]);
      if (n < 8) v.assign(R.begin(), R.end());
      if (shift)
        for (auto& x : v) x = hexl_amd_multiply_factor(x, shift, q);
      tables[idx] = std::move(v);
    });
    return tables[idx];
  }
};

const NTT::State& NTT::state() const {
  if (!m_state) throw std::runtime_error("hexl: NTT object is empty");
  return *m_state;
}

NTT::NTT(uint64_t degree, uint64_t q, uint64_t root_of_unity,
         std::shared_ptr<AllocatorBase> alloc_ptr)
    : m_state(std::make_shared<State>(alloc_ptr)) {
  check(hexl_amd_ntt_create(&m_state->plan, degree, q, root_of_unity, -1));
  m_state->degree = degree;
  m_state->q = q;
  m_state->w = hexl_amd_ntt_root_of_unity(m_state->plan);
  // the reference fills its tables in the constructor through the allocator
  // (test/test-ntt.cpp:168-200 observes the allocation); do the same for the
  // four tables the transforms are defined by
  m_state->base_table(0);
  m_state->base_table(2);
  m_state->base_table(3);
  m_state->base_table(6);
}

NTT::NTT(uint64_t degree, uint64_t q, std::shared_ptr<AllocatorBase> alloc_ptr)
    : NTT(degree, q, uint64_t{0} /* 0 selects MinimalPrimitiveRoot(2N, q) */, alloc_ptr) {}

bool NTT::CheckArguments(uint64_t degree, uint64_t modulus) {
  return hexl_amd_ntt_check_arguments(degree, modulus) != 0;
}

void NTT::ComputeForwardBatch(uint64_t* result, const uint64_t* operand, uint64_t batch,
                              uint64_t input_mod_factor, uint64_t output_mod_factor) {
  const State& s = state();
  HEXL_SHIM_CHECK_BOUNDS(operand, s.degree * batch, s.q * input_mod_factor, "operand");
  if (hexl_amd_pointer_is_device(operand) && hexl_amd_pointer_is_device(result)) {
    check(hexl_amd_ntt_forward(s.plan, result, operand, batch, input_mod_factor,
                               output_mod_factor, nullptr));
  } else {
    check(hexl_amd_ntt_forward_host(s.plan, result, operand, batch, input_mod_factor,
                                    output_mod_factor));
  }
}

void NTT::ComputeInverseBatch(uint64_t* result, const uint64_t* operand, uint64_t batch,
                              uint64_t input_mod_factor, uint64_t output_mod_factor) {
  const State& s = state();
  HEXL_SHIM_CHECK_BOUNDS(operand, s.degree * batch, s.q * input_mod_factor, "operand");
  if (hexl_amd_pointer_is_device(operand) && hexl_amd_pointer_is_device(result)) {
    check(hexl_amd_ntt_inverse(s.plan, result, operand, batch, input_mod_factor,
                               output_mod_factor, nullptr));
  } else {
    check(hexl_amd_ntt_inverse_host(s.plan, result, operand, batch, input_mod_factor,
                                    output_mod_factor));
  }
}

void NTT::ComputeForward(uint64_t* result, const uint64_t* operand, uint64_t input_mod_factor,
                         uint64_t output_mod_factor) {
  ComputeForwardBatch(result, operand, 1, input_mod_factor, output_mod_factor);
}

void NTT::ComputeInverse(uint64_t* result, const uint64_t* operand, uint64_t input_mod_factor,
                         uint64_t output_mod_factor) {
  ComputeInverseBatch(result, operand, 1, input_mod_factor, output_mod_factor);
}

// Extension: polynomials of several moduli in one call (hexl_amd_ntt_*_map / _indexed).
const void* NTT::PlanHandle() const { return state().plan; }

namespace {
void compute_map(bool forward, const NTT* const* ntts, size_t num_ntts,
                 const uint8_t* plan_of_slot, uint64_t period, uint64_t inner,
                 const uint32_t* prime_index, uint64_t* result, const uint64_t* operand,
                 uint64_t polys, uint64_t in_mf, uint64_t out_mf) {
  if (!ntts || num_ntts == 0) throw std::invalid_argument("hexl: no NTT objects");
  if (!prime_index && (!plan_of_slot || period == 0 || inner == 0))
    throw std::invalid_argument("hexl: empty prime map");
  std::vector<const hexl_amd_ntt*> plans(num_ntts);
  for (size_t k = 0; k < num_ntts; ++k) {
    if (!ntts[k]) throw std::invalid_argument("hexl: null NTT object");
    plans[k] = static_cast<const hexl_amd_ntt*>(ntts[k]->PlanHandle());
  }
  const uint64_t n = ntts[0]->GetDegree();
  // polynomial i sits at operand + i * n whatever its modulus: one degree for all
  for (size_t k = 1; k < num_ntts; ++k)
    if (ntts[k]->GetDegree() != n)
      throw std::invalid_argument("hexl: the NTT objects of a prime map must share one degree");
  auto which = [&](uint64_t i) -> uint64_t {
    return prime_index ? prime_index[i] : plan_of_slot[(i / inner) % period];
  };
#ifdef HEXL_DEBUG
  for (uint64_t i = 0; i < polys; ++i)
    if (which(i) < num_ntts)
      HEXL_SHIM_CHECK_BOUNDS(operand + i * n, n, ntts[which(i)]->GetModulus() * in_mf, "operand");
#endif
  if (hexl_amd_pointer_is_device(operand) && hexl_amd_pointer_is_device(result)) {
    if (prime_index)
      check((forward ? hexl_amd_ntt_forward_indexed : hexl_amd_ntt_inverse_indexed)(
          plans.data(), num_ntts, prime_index, result, operand, polys, in_mf, out_mf, nullptr));
    else
      check((forward ? hexl_amd_ntt_forward_map : hexl_amd_ntt_inverse_map)(
          plans.data(), num_ntts, plan_of_slot, period, inner, result, operand, polys, in_mf,
          out_mf, nullptr));
    return;
  }
  // host memory: run by run through the staged single-modulus path
  for (uint64_t start = 0; start < polys;) {
    const uint64_t k = which(start);
    if (k >= num_ntts) throw std::invalid_argument("hexl: prime index out of range");
    uint64_t end = start + 1;
    while (end < polys && which(end) == k) ++end;
    check((forward ? hexl_amd_ntt_forward_host : hexl_amd_ntt_inverse_host)(
        plans[k], result + start * n, operand + start * n, end - start, in_mf, out_mf));
    start = end;
  }
}
}  // namespace

void NTT::ComputeForwardMap(const NTT* const* ntts, size_t num_ntts, const uint8_t* plan_of_slot,
                            uint64_t period, uint64_t inner, uint64_t* result,
                            const uint64_t* operand, uint64_t polys, uint64_t input_mod_factor,
                            uint64_t output_mod_factor) {
  compute_map(true, ntts, num_ntts, plan_of_slot, period, inner, nullptr, result, operand, polys,
              input_mod_factor, output_mod_factor);
}
void NTT::ComputeInverseMap(const NTT* const* ntts, size_t num_ntts, const uint8_t* plan_of_slot,
                            uint64_t period, uint64_t inner, uint64_t* result,
                            const uint64_t* operand, uint64_t polys, uint64_t input_mod_factor,
                            uint64_t output_mod_factor) {
  compute_map(false, ntts, num_ntts, plan_of_slot, period, inner, nullptr, result, operand, polys,
              input_mod_factor, output_mod_factor);
}
void NTT::ComputeForwardIndexed(const NTT* const* ntts, size_t num_ntts,
                                const uint32_t* prime_index, uint64_t* result,
                                const uint64_t* operand, uint64_t polys,
                                uint64_t input_mod_factor, uint64_t output_mod_factor) {
  if (!prime_index) throw std::invalid_argument("hexl: prime_index == nullptr");
  compute_map(true, ntts, num_ntts, nullptr, 0, 0, prime_index, result, operand, polys,
              input_mod_factor, output_mod_factor);
}
void NTT::ComputeInverseIndexed(const NTT* const* ntts, size_t num_ntts,
                                const uint32_t* prime_index, uint64_t* result,
                                const uint64_t* operand, uint64_t polys,
                                uint64_t input_mod_factor, uint64_t output_mod_factor) {
  if (!prime_index) throw std::invalid_argument("hexl: prime_index == nullptr");
  compute_map(false, ntts, num_ntts, nullptr, 0, 0, prime_index, result, operand, polys,
              input_mod_factor, output_mod_factor);
}

uint64_t NTT::GetMinimalRootOfUnity() const { return state().w; }
uint64_t NTT::GetDegree() const { return state().degree; }
uint64_t NTT::GetModulus() const { return state().q; }

const Table& NTT::GetRootOfUnityPowers() const { return state().base_table(0); }
const Table& NTT::GetPrecon32RootOfUnityPowers() const { return state().base_table(1); }
const Table& NTT::GetPrecon64RootOfUnityPowers() const { return state().base_table(2); }
const Table& NTT::GetInvRootOfUnityPowers() const { return state().base_table(3); }
const Table& NTT::GetPrecon32InvRootOfUnityPowers() const { return state().base_table(4); }
const Table& NTT::GetPrecon52InvRootOfUnityPowers() const { return state().base_table(5); }
const Table& NTT::GetPrecon64InvRootOfUnityPowers() const { return state().base_table(6); }
const Table& NTT::GetAVX512RootOfUnityPowers() const { return state().avx_table(7, 0); }
const Table& NTT::GetAVX512Precon32RootOfUnityPowers() const { return state().avx_table(8, 32); }
const Table& NTT::GetAVX512Precon52RootOfUnityPowers() const { return state().avx_table(9, 52); }
const Table& NTT::GetAVX512Precon64RootOfUnityPowers() const { return state().avx_table(10, 64); }

// ---------------------------------------------------------------- Eltwise
namespace {

bool on_device(const void* a, const void* b, const void* c) {
  return hexl_amd_pointer_is_device(a) && (!b || hexl_amd_pointer_is_device(b)) &&
         (!c || hexl_amd_pointer_is_device(c));
}

}  // namespace

void EltwiseAddMod(uint64_t* result, const uint64_t* operand1, const uint64_t* operand2,
                   uint64_t n, uint64_t modulus) {
  HEXL_SHIM_CHECK_BOUNDS(operand1, n, modulus, "operand1");
  HEXL_SHIM_CHECK_BOUNDS(operand2, n, modulus, "operand2");
  if (on_device(result, operand1, operand2))
    check(hexl_amd_eltwise_add_mod(result, operand1, operand2, n, modulus, nullptr));
  else
    check(hexl_amd_eltwise_host(0, result, operand1, operand2, 0, n, modulus, 1, 1));
}

void EltwiseAddMod(uint64_t* result, const uint64_t* operand1, uint64_t operand2, uint64_t n,
                   uint64_t modulus) {
  HEXL_SHIM_CHECK_BOUNDS(operand1, n, modulus, "operand1");
  if (on_device(result, operand1, nullptr))
    check(hexl_amd_eltwise_add_mod_scalar(result, operand1, operand2, n, modulus, nullptr));
  else
    check(hexl_amd_eltwise_host(1, result, operand1, nullptr, operand2, n, modulus, 1, 1));
}

void EltwiseSubMod(uint64_t* result, const uint64_t* operand1, const uint64_t* operand2,
                   uint64_t n, uint64_t modulus) {
  HEXL_SHIM_CHECK_BOUNDS(operand1, n, modulus, "operand1");
  HEXL_SHIM_CHECK_BOUNDS(operand2, n, modulus, "operand2");
  if (on_device(result, operand1, operand2))
    check(hexl_amd_eltwise_sub_mod(result, operand1, operand2, n, modulus, nullptr));
  else
    check(hexl_amd_eltwise_host(2, result, operand1, operand2, 0, n, modulus, 1, 1));
}

void EltwiseSubMod(uint64_t* result, const uint64_t* operand1, uint64_t operand2, uint64_t n,
                   uint64_t modulus) {
  HEXL_SHIM_CHECK_BOUNDS(operand1, n, modulus, "operand1");
  if (on_device(result, operand1, nullptr))
    check(hexl_amd_eltwise_sub_mod_scalar(result, operand1, operand2, n, modulus, nullptr));
  else
    check(hexl_amd_eltwise_host(3, result, operand1, nullptr, operand2, n, modulus, 1, 1));
}

void EltwiseMultMod(uint64_t* result, const uint64_t* operand1, const uint64_t* operand2,
                    uint64_t n, uint64_t modulus, uint64_t input_mod_factor) {
  HEXL_SHIM_CHECK_BOUNDS(operand1, n, input_mod_factor * modulus, "operand1");
  HEXL_SHIM_CHECK_BOUNDS(operand2, n, input_mod_factor * modulus, "operand2");
  if (on_device(result, operand1, operand2))
    check(hexl_amd_eltwise_mult_mod(result, operand1, operand2, n, modulus, input_mod_factor,
                                    nullptr));
  else
    check(hexl_amd_eltwise_host(4, result, operand1, operand2, 0, n, modulus, input_mod_factor,
                                1));
}

void EltwiseFMAMod(uint64_t* result, const uint64_t* arg1, uint64_t arg2, const uint64_t* arg3,
                   uint64_t n, uint64_t modulus, uint64_t input_mod_factor) {
  HEXL_SHIM_CHECK_BOUNDS(arg1, n, input_mod_factor * modulus, "arg1");
  HEXL_SHIM_CHECK_BOUNDS(arg3, n, input_mod_factor * modulus, "arg3");
  if (on_device(result, arg1, arg3))
    check(hexl_amd_eltwise_fma_mod(result, arg1, arg2, arg3, n, modulus, input_mod_factor,
                                   nullptr));
  else
    check(hexl_amd_eltwise_host(5, result, arg1, arg3, arg2, n, modulus, input_mod_factor, 1));
}

void EltwiseReduceMod(uint64_t* result, const uint64_t* operand, uint64_t n, uint64_t modulus,
                      uint64_t input_mod_factor, uint64_t output_mod_factor) {
  if (input_mod_factor != modulus)  // (input_mod_factor == modulus: any 64-bit input is legal)
    HEXL_SHIM_CHECK_BOUNDS(operand, n, input_mod_factor * modulus, "operand");
  if (on_device(result, operand, nullptr))
    check(hexl_amd_eltwise_reduce_mod(result, operand, n, modulus, input_mod_factor,
                                      output_mod_factor, nullptr));
  else
    check(hexl_amd_eltwise_host(6, result, operand, nullptr, 0, n, modulus, input_mod_factor,
                                output_mod_factor));
}

void DyadicMultiply(uint64_t* result, const uint64_t* operand1, const uint64_t* operand2,
                    uint64_t n, const uint64_t* moduli, uint64_t num_moduli) {
  if (on_device(result, operand1, operand2))
    check(hexl_amd_dyadic_multiply(result, operand1, operand2, n, moduli, num_moduli, nullptr));
  else
    check(hexl_amd_dyadic_multiply_host(result, operand1, operand2, n, moduli, num_moduli));
}

void KeySwitch(uint64_t* result, const uint64_t* t_target_iter_ptr, uint64_t n,
               uint64_t decomp_modulus_size, uint64_t key_modulus_size,
               uint64_t rns_modulus_size, uint64_t key_component_count, const uint64_t* moduli,
               const uint64_t** k_switch_keys, const uint64_t* modswitch_factors,
               const uint64_t* root_of_unity_powers_ptr) {
  if (root_of_unity_powers_ptr != nullptr)  // key-switch-internal.cpp:31-34
    throw std::invalid_argument("Parameter root_of_unity_powers_ptr is not supported yet.");
  // (host result / target with device-resident keys goes through the host entry point, which
  // uses such key blocks where they lie)
  const bool dev = on_device(result, t_target_iter_ptr,
                             k_switch_keys ? k_switch_keys[0] : nullptr);
  if (dev)
    check(hexl_amd_key_switch(result, t_target_iter_ptr, n, decomp_modulus_size, key_modulus_size,
                              rns_modulus_size, key_component_count, moduli, k_switch_keys,
                              modswitch_factors, nullptr));
  else
    check(hexl_amd_key_switch_host(result, t_target_iter_ptr, n, decomp_modulus_size,
                                   key_modulus_size, rns_modulus_size, key_component_count,
                                   moduli, k_switch_keys, modswitch_factors));
}

void KeySwitchBatch(uint64_t* result, const uint64_t* t_target_iter_ptr, uint64_t num_targets,
                    uint64_t n, uint64_t decomp_modulus_size, uint64_t key_modulus_size,
                    uint64_t rns_modulus_size, uint64_t key_component_count,
                    const uint64_t* moduli, const uint64_t** k_switch_keys,
                    const uint64_t* modswitch_factors) {
  if (!on_device(result, t_target_iter_ptr, k_switch_keys ? k_switch_keys[0] : nullptr))
    throw std::invalid_argument("KeySwitchBatch takes device memory");
  check(hexl_amd_key_switch_batch(result, t_target_iter_ptr, num_targets, n, decomp_modulus_size,
                                  key_modulus_size, rns_modulus_size, key_component_count, moduli,
                                  k_switch_keys, modswitch_factors, nullptr));
}

void DyadicMultiplyBatch(uint64_t* result, const uint64_t* operand1, const uint64_t* operand2,
                         uint64_t num_pairs, uint64_t n, const uint64_t* moduli,
                         uint64_t num_moduli) {
  if (!on_device(result, operand1, operand2))
    throw std::invalid_argument("DyadicMultiplyBatch takes device memory");
  check(hexl_amd_dyadic_multiply_batch(result, operand1, operand2, num_pairs, n, moduli,
                                       num_moduli, nullptr));
}

void EltwiseCmpAdd(uint64_t* result, const uint64_t* operand1, uint64_t n, CMPINT cmp,
                   uint64_t bound, uint64_t diff) {
  if (on_device(result, operand1, nullptr))
    check(hexl_amd_eltwise_cmp_add(result, operand1, n, static_cast<int>(cmp), bound, diff,
                                   nullptr));
  else
    check(hexl_amd_eltwise_cmp_host(result, operand1, n, 0, static_cast<int>(cmp), bound, diff));
}

void EltwiseCmpSubMod(uint64_t* result, const uint64_t* operand1, uint64_t n, uint64_t modulus,
                      CMPINT cmp, uint64_t bound, uint64_t diff) {
  if (modulus <= 1) throw std::runtime_error("EltwiseCmpSubMod: modulus must be > 1");
  if (on_device(result, operand1, nullptr))
    check(hexl_amd_eltwise_cmp_sub_mod(result, operand1, n, modulus, static_cast<int>(cmp), bound,
                                       diff, nullptr));
  else
    check(hexl_amd_eltwise_cmp_host(result, operand1, n, modulus, static_cast<int>(cmp), bound,
                                    diff));
}

// The reference's public DyadicMultiply / KeySwitch are one-line forwards to these
// (hexl/experimental/seal/dyadic-multiply.cpp, key-switch.cpp); callers that name the internal
// entry points (hexl.hpp includes their headers) get the same GPU path.
namespace internal {

void DyadicMultiply(uint64_t* result, const uint64_t* operand1, const uint64_t* operand2,
                    uint64_t n, const uint64_t* moduli, uint64_t num_moduli) {
  ::intel::hexl::DyadicMultiply(result, operand1, operand2, n, moduli, num_moduli);
}

void KeySwitch(uint64_t* result, const uint64_t* t_target_iter_ptr, uint64_t n,
               uint64_t decomp_modulus_size, uint64_t key_modulus_size,
               uint64_t rns_modulus_size, uint64_t key_component_count, const uint64_t* moduli,
               const uint64_t** k_switch_keys, const uint64_t* modswitch_factors,
               const uint64_t* root_of_unity_powers_ptr) {
  ::intel::hexl::KeySwitch(result, t_target_iter_ptr, n, decomp_modulus_size, key_modulus_size,
                           rns_modulus_size, key_component_count, moduli, k_switch_keys,
                           modswitch_factors, root_of_unity_powers_ptr);
}

}  // namespace internal

}  // namespace hexl
}  // namespace intel
