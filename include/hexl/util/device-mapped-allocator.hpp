// hexl/util/device-mapped-allocator.hpp -- host memory the GPU kernels can address.
//
// Extension (no counterpart in the reference).  The reference's functions take host pointers
// and a caller that keeps doing so pays two PCIe copies and their synchronisation per call.
// Buffers that are pinned AND mapped into the device's address space need none of that: the
// one-kernel transforms (degree <= 2^13, 2^14 from 192 polynomials) and the element-wise
// kernels run straight on them over the link.  Two ways to get such buffers:
//   * allocate them here -- DeviceMappedAllocator is an intel::hexl::AllocatorBase
//     (hexl/include/hexl/util/allocator.hpp:12-51), so
//         AlignedVector64<uint64_t> v(n, AlignedAllocator<uint64_t, 64>(DeviceMappedStrategy()));
//     or DeviceMappedVector(n);
//   * register an existing allocation once (a caller's memory pool):
//         RegisterHostMemory(pool_base, pool_bytes);  ...  UnregisterHostMemory(pool_base);
// Nothing else changes at the call sites: intel::hexl::NTT / Eltwise* recognise the memory.
#pragma once
#include <cstddef>
#include <cstdint>

#include "hexl/util/aligned-allocator.hpp"
#include "hexl/util/allocator.hpp"

namespace intel {
namespace hexl {

/// Pinned, device-mapped host memory (throws std::runtime_error on failure / without a GPU).
void* DeviceMappedAllocate(size_t bytes);
void DeviceMappedFree(void* p) noexcept;
/// Pins and maps [p, p + bytes) -- memory the caller already owns.
void RegisterHostMemory(void* p, size_t bytes);
void UnregisterHostMemory(void* p) noexcept;

/// Device memory for callers without a HIP toolchain: buffers for the device-pointer forms of
/// NTT / Eltwise* (the throughput path).  The memory is NOT host-accessible: move data with
/// Copy (either side may be host, mapped or device memory; synchronous).
void* DeviceMalloc(size_t bytes);
void DeviceFree(void* p) noexcept;
void Copy(void* dst, const void* src, size_t bytes);
void DeviceSynchronize();

struct DeviceMappedAllocator : AllocatorBase {
  void* allocate(size_t bytes_count) final { return DeviceMappedAllocate(bytes_count); }
  void deallocate(void* p, size_t) final { DeviceMappedFree(p); }
};

/// One shared DeviceMappedAllocator.
AllocatorStrategyPtr DeviceMappedStrategy();

/// n zero-initialised words in device-mapped host memory.
inline AlignedVector64<uint64_t> DeviceMappedVector(size_t n) {
  return AlignedVector64<uint64_t>(n, 0, AlignedAllocator<uint64_t, 64>(DeviceMappedStrategy()));
}

}  // namespace hexl
}  // namespace intel
