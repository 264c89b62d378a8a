// hexl/util/aligned-allocator.hpp -- 64-byte aligned std::vector over a pluggable
// AllocatorBase (reference: hexl/include/hexl/util/aligned-allocator.hpp:18-107).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <type_traits>
#include <vector>

#include "hexl/util/allocator.hpp"
#include "hexl/util/defines.hpp"

namespace intel {
namespace hexl {

/// Default strategy: malloc / free.
struct MallocStrategy : AllocatorBase {
  void* allocate(size_t bytes_count) final { return std::malloc(bytes_count); }
  void deallocate(void* p, size_t) final { std::free(p); }
};

using AllocatorStrategyPtr = std::shared_ptr<AllocatorBase>;
extern AllocatorStrategyPtr mallocStrategy;

/// std-compatible allocator returning Alignment-byte aligned storage carved out
/// of blocks obtained from an AllocatorBase.  The pointer to the enclosing block
/// is kept in the word just below the aligned address.
template <typename T, uint64_t Alignment>
class AlignedAllocator {
 public:
  template <typename, uint64_t>
  friend class AlignedAllocator;
  using value_type = T;

  explicit AlignedAllocator(AllocatorStrategyPtr strategy = nullptr) noexcept
      : m_strategy(strategy ? std::move(strategy) : mallocStrategy) {}
  AlignedAllocator(const AlignedAllocator&) = default;
  AlignedAllocator& operator=(const AlignedAllocator&) = default;
  template <typename U>
  AlignedAllocator(const AlignedAllocator<U, Alignment>& other) : m_strategy(other.m_strategy) {}

  template <typename U>
  struct rebind {
    using other = AlignedAllocator<U, Alignment>;
  };
  // Two allocators are interchangeable iff they draw from the same strategy.  (The reference
  // answers "equal" unconditionally, aligned-allocator.hpp:60-61: a vector move-assigned
  // across strategies then frees one strategy's block through the other.  With strategies
  // that differ in kind -- malloc and pinned device-mapped memory -- that is a crash, so the
  // comparison is real here and the allocator travels with its buffer.)
  bool operator==(const AlignedAllocator& other) const { return m_strategy == other.m_strategy; }
  bool operator!=(const AlignedAllocator& other) const { return !(*this == other); }
  using propagate_on_container_move_assignment = std::true_type;
  using propagate_on_container_copy_assignment = std::true_type;
  using propagate_on_container_swap = std::true_type;

  T* allocate(size_t n) {
    static_assert((Alignment & (Alignment - 1)) == 0, "Alignment must be a power of two");
    const size_t payload = sizeof(T) * n;
    const size_t block = payload + Alignment + sizeof(void*);
    char* raw = static_cast<char*>(m_strategy->allocate(block));
    if (!raw) return nullptr;
    uintptr_t addr = reinterpret_cast<uintptr_t>(raw + sizeof(void*));
    addr = (addr + Alignment - 1) & ~static_cast<uintptr_t>(Alignment - 1);
    reinterpret_cast<void**>(addr)[-1] = raw;
    return reinterpret_cast<T*>(addr);
  }

  void deallocate(T* p, size_t n) {
    if (!p) return;
    m_strategy->deallocate(reinterpret_cast<void**>(p)[-1], n);
  }

 private:
  AllocatorStrategyPtr m_strategy;
};

template <typename T>
using AlignedVector64 = std::vector<T, AlignedAllocator<T, 64> >;

}  // namespace hexl
}  // namespace intel
