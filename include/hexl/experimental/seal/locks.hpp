// hexl/experimental/seal/locks.hpp -- reader/writer lock vocabulary of the SEAL helpers.
// Drop-in for hexl/include/hexl/experimental/seal/locks.hpp:12-37 (the names are the interface).
#pragma once

#include <mutex>
#include <shared_mutex>

namespace intel {
namespace hexl {

using Lock = std::shared_mutex;
using WriteLock = std::unique_lock<Lock>;
using ReadLock = std::shared_lock<Lock>;

/// A shared mutex handed out as scoped read / write locks.
class RWLock {
 public:
  RWLock() = default;
  RWLock(const RWLock&) = delete;
  RWLock& operator=(const RWLock&) = delete;

  ReadLock AcquireRead() { return ReadLock(mutex_); }
  WriteLock AcquireWrite() { return WriteLock(mutex_); }
  ReadLock TryAcquireRead() noexcept { return ReadLock(mutex_, std::try_to_lock); }
  WriteLock TryAcquireWrite() noexcept { return WriteLock(mutex_, std::try_to_lock); }

 private:
  Lock mutex_;
};

}  // namespace hexl
}  // namespace intel
