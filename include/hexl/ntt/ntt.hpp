// hexl/ntt/ntt.hpp -- intel::hexl::NTT backed by the MI355X kernels.
// Drop-in for hexl/include/hexl/ntt/ntt.hpp:22-293: same constructors,
// ComputeForward / ComputeInverse, getters and constants; source compatible
// (the class layout differs: the state lives behind a shared pointer, so copies
// share one immutable set of tables and one device plan).
#pragma once
#include <stdint.h>

#include <memory>
#include <utility>
#include <vector>

#include "hexl/util/aligned-allocator.hpp"
#include "hexl/util/allocator.hpp"
#include "hexl/util/check.hpp"

namespace intel {
namespace hexl {

/// Negacyclic number-theoretic transform over Z_q[X]/(X^N + 1), bit-reversed
/// output order, Harvey lazy ranges.  Thread-safe: an object may be shared by
/// concurrent callers.
class NTT {
 public:
  /// Wraps any object with allocate(size_t) / deallocate(void*, size_t) as an
  /// AllocatorBase (SEAL passes its MemoryPool through this).
  template <class Adaptee, class... Args>
  struct AllocatorAdapter : public AllocatorInterface<AllocatorAdapter<Adaptee, Args...>> {
    explicit AllocatorAdapter(Adaptee&& _a, Args&&... args);
    AllocatorAdapter(const Adaptee& _a, Args&... args);
    void* allocate_impl(size_t bytes_count);
    void deallocate_impl(void* p, size_t n);

   private:
    Adaptee alloc;
  };

  /// Empty object; every other member requires a constructed one.
  NTT() = default;
  ~NTT() = default;

  /// degree: power of two; q: prime with q == 1 (mod 2 * degree).  Uses the
  /// minimal primitive 2N-th root of unity.  alloc_ptr supplies the host memory
  /// of the tables the getters expose.
  NTT(uint64_t degree, uint64_t q, std::shared_ptr<AllocatorBase> alloc_ptr = {});

  template <class Allocator, class... AllocatorArgs>
  NTT(uint64_t degree, uint64_t q, Allocator&& a, AllocatorArgs&&... args)
      : NTT(degree, q,
            std::static_pointer_cast<AllocatorBase>(
                std::make_shared<AllocatorAdapter<Allocator, AllocatorArgs...>>(
                    std::move(a), std::forward<AllocatorArgs>(args)...))) {}

  /// As above with a caller-chosen primitive 2N-th root of unity.
  NTT(uint64_t degree, uint64_t q, uint64_t root_of_unity,
      std::shared_ptr<AllocatorBase> alloc_ptr = {});

  template <class Allocator, class... AllocatorArgs>
  NTT(uint64_t degree, uint64_t q, uint64_t root_of_unity, Allocator&& a,
      AllocatorArgs&&... args)
      : NTT(degree, q, root_of_unity,
            std::static_pointer_cast<AllocatorBase>(
                std::make_shared<AllocatorAdapter<Allocator, AllocatorArgs...>>(
                    std::move(a), std::forward<AllocatorArgs>(args)...))) {}

  /// True iff (degree, modulus) is a legal parameter set.
  static bool CheckArguments(uint64_t degree, uint64_t modulus);

  /// Forward transform; result in bit-reversed order.  operand in
  /// [0, input_mod_factor * q), input_mod_factor in {1, 2, 4}; result in
  /// [0, output_mod_factor * q), output_mod_factor in {1, 4}.  Pointers may be
  /// host or device memory; result may alias operand.
  void ComputeForward(uint64_t* result, const uint64_t* operand, uint64_t input_mod_factor,
                      uint64_t output_mod_factor);

  /// Inverse transform of a bit-reversed operand.  input_mod_factor in {1, 2},
  /// output_mod_factor in {1, 2}.
  void ComputeInverse(uint64_t* result, const uint64_t* operand, uint64_t input_mod_factor,
                      uint64_t output_mod_factor);

  /// Extension: `batch` polynomials back to back in one call (device or host
  /// pointers).  ComputeForward(r, o, i, f) == ComputeForwardBatch(r, o, 1, i, f).
  void ComputeForwardBatch(uint64_t* result, const uint64_t* operand, uint64_t batch,
                           uint64_t input_mod_factor, uint64_t output_mod_factor);
  void ComputeInverseBatch(uint64_t* result, const uint64_t* operand, uint64_t batch,
                           uint64_t input_mod_factor, uint64_t output_mod_factor);

  /// Extension: polynomials of SEVERAL moduli in one call (the per-modulus loops of RNS
  /// callers, hexl/experimental/seal/key-switch-internal.cpp:51-90, as one launch sequence on
  /// the GPU).  `polys` polynomials back to back; polynomial i is transformed by
  /// *ntts[plan_of_slot[(i / inner) % period]] (Map) or *ntts[prime_index[i]] (Indexed).
  /// SEAL's [ciphertext][component][modulus][N] layout is inner = 1, period = k,
  /// plan_of_slot = {0 .. k-1}; prime-major blocks are inner = polynomials per prime.  All
  /// objects share one degree and one device.  Device or host pointers.
  static void ComputeForwardMap(const NTT* const* ntts, size_t num_ntts,
                                const uint8_t* plan_of_slot, uint64_t period, uint64_t inner,
                                uint64_t* result, const uint64_t* operand, uint64_t polys,
                                uint64_t input_mod_factor, uint64_t output_mod_factor);
  static void ComputeInverseMap(const NTT* const* ntts, size_t num_ntts,
                                const uint8_t* plan_of_slot, uint64_t period, uint64_t inner,
                                uint64_t* result, const uint64_t* operand, uint64_t polys,
                                uint64_t input_mod_factor, uint64_t output_mod_factor);
  static void ComputeForwardIndexed(const NTT* const* ntts, size_t num_ntts,
                                    const uint32_t* prime_index, uint64_t* result,
                                    const uint64_t* operand, uint64_t polys,
                                    uint64_t input_mod_factor, uint64_t output_mod_factor);
  static void ComputeInverseIndexed(const NTT* const* ntts, size_t num_ntts,
                                    const uint32_t* prime_index, uint64_t* result,
                                    const uint64_t* operand, uint64_t polys,
                                    uint64_t input_mod_factor, uint64_t output_mod_factor);
  /// The C-ABI plan behind this object (a `const hexl_amd_ntt*`, include/hexl_amd.h), for
  /// callers that mix the C++ API with the C one.
  const void* PlanHandle() const;

  uint64_t GetMinimalRootOfUnity() const;
  uint64_t GetDegree() const;
  uint64_t GetModulus() const;

  /// Powers of the root in bit-reversed order and their Barrett factors.
  const AlignedVector64<uint64_t>& GetRootOfUnityPowers() const;
  uint64_t GetRootOfUnityPower(size_t i) { return GetRootOfUnityPowers()[i]; }
  const AlignedVector64<uint64_t>& GetPrecon32RootOfUnityPowers() const;
  const AlignedVector64<uint64_t>& GetPrecon64RootOfUnityPowers() const;
  /// The reference's AVX512 layouts (entries [N/8,N/4) x4 and [N/4,N/2) x2
  /// duplicated), kept for callers that read them; the GPU does not use them.
  const AlignedVector64<uint64_t>& GetAVX512RootOfUnityPowers() const;
  const AlignedVector64<uint64_t>& GetAVX512Precon32RootOfUnityPowers() const;
  const AlignedVector64<uint64_t>& GetAVX512Precon52RootOfUnityPowers() const;
  const AlignedVector64<uint64_t>& GetAVX512Precon64RootOfUnityPowers() const;
  /// Inverse powers in the reference's stage order and their Barrett factors.
  const AlignedVector64<uint64_t>& GetInvRootOfUnityPowers() const;
  uint64_t GetInvRootOfUnityPower(size_t i) { return GetInvRootOfUnityPowers()[i]; }
  const AlignedVector64<uint64_t>& GetPrecon32InvRootOfUnityPowers() const;
  const AlignedVector64<uint64_t>& GetPrecon52InvRootOfUnityPowers() const;
  const AlignedVector64<uint64_t>& GetPrecon64InvRootOfUnityPowers() const;

  static size_t MaxDegreeBits() { return 20; }
  static size_t MaxModulusBits() { return 62; }
  static const size_t s_default_shift_bits{64};
  static const size_t s_ifma_shift_bits{52};
  static const size_t s_max_fwd_32_modulus{1ULL << (32 - 2)};
  static const size_t s_max_inv_32_modulus{1ULL << (32 - 2)};
  static const size_t s_max_fwd_ifma_modulus{1ULL << (s_ifma_shift_bits - 2)};
  static const size_t s_max_inv_ifma_modulus{1ULL << (s_ifma_shift_bits - 2)};
  static const size_t s_max_inv_dq_modulus{1ULL << (s_default_shift_bits - 2)};
  static size_t s_max_fwd_modulus(int bit_shift) {
    if (bit_shift == 32) return s_max_fwd_32_modulus;
    if (bit_shift == 52) return s_max_fwd_ifma_modulus;
    if (bit_shift == 64) return 1ULL << MaxModulusBits();
    HEXL_CHECK(false, "Invalid bit_shift " << bit_shift);
    return 0;
  }
  static size_t s_max_inv_modulus(int bit_shift) {
    if (bit_shift == 32) return s_max_inv_32_modulus;
    if (bit_shift == 52) return s_max_inv_ifma_modulus;
    if (bit_shift == 64) return 1ULL << MaxModulusBits();
    HEXL_CHECK(false, "Invalid bit_shift " << bit_shift);
    return 0;
  }

 private:
  struct State;
  std::shared_ptr<State> m_state;
  const State& state() const;
};

template <class Adaptee, class... Args>
NTT::AllocatorAdapter<Adaptee, Args...>::AllocatorAdapter(Adaptee&& _a, Args&&...)
    : alloc(std::move(_a)) {}
template <class Adaptee, class... Args>
NTT::AllocatorAdapter<Adaptee, Args...>::AllocatorAdapter(const Adaptee& _a, Args&...)
    : alloc(_a) {}
template <class Adaptee, class... Args>
void* NTT::AllocatorAdapter<Adaptee, Args...>::allocate_impl(size_t bytes_count) {
  return alloc.allocate(bytes_count);
}
template <class Adaptee, class... Args>
void NTT::AllocatorAdapter<Adaptee, Args...>::deallocate_impl(void* p, size_t n) {
  alloc.deallocate(p, n);
}

}  // namespace hexl
}  // namespace intel
