// hexl/util/allocator.hpp -- pluggable host memory allocation for the tables an
// NTT object owns (reference: hexl/include/hexl/util/allocator.hpp:12-51).
#pragma once
#include <cstddef>

#include "hexl/util/defines.hpp"

namespace intel {
namespace hexl {

/// Interface a caller implements to supply the host memory of NTT tables
/// (SEAL plugs its MemoryPool in through NTT::AllocatorAdapter).
struct AllocatorBase {
  virtual ~AllocatorBase() noexcept {}
  virtual void* allocate(size_t bytes_count) = 0;
  virtual void deallocate(void* p, size_t n) = 0;
};

/// CRTP helper: forwards the virtual interface to Impl::allocate_impl /
/// Impl::deallocate_impl.
template <class Impl>
struct AllocatorInterface : public AllocatorBase {
  void* allocate(size_t bytes_count) override {
    return static_cast<Impl*>(this)->allocate_impl(bytes_count);
  }
  void deallocate(void* p, size_t n) override {
    static_cast<Impl*>(this)->deallocate_impl(p, n);
  }

 private:
  // defaults used when Impl does not provide its own
  void* allocate_impl(size_t) { return nullptr; }
  void deallocate_impl(void*, size_t) {}
};

}  // namespace hexl
}  // namespace intel
