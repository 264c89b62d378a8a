// hexl/experimental/seal/ntt-cache.hpp -- GetNTT(N, modulus): one NTT object per (degree,
// modulus) for the life of the process, shared between threads.
// Drop-in for hexl/include/hexl/experimental/seal/ntt-cache.hpp:13-53.  (The reference's header
// includes the private ntt/ntt-internal.hpp; this one needs only the public class.)  The objects
// are thin handles: the device tables behind them are built once per (N, modulus, device) by
// hexl_amd_ntt_create, and the library's own composites (KeySwitch) keep a plan cache of their
// own inside libhexl_amd.so.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <unordered_map>
#include <utility>

#include "hexl/experimental/seal/locks.hpp"
#include "hexl/ntt/ntt.hpp"

namespace intel {
namespace hexl {

/// Hash of a pair, as the cache keys need it.
struct HashPair {
  template <class T1, class T2>
  std::size_t operator()(const std::pair<T1, T2>& p) const {
    return hash_combine(std::hash<T1>{}(p.first), std::hash<T2>{}(p.second));
  }
  static std::size_t hash_combine(std::size_t lhs, std::size_t rhs) {
    return lhs ^ (rhs + 0x9e3779b9 + (lhs << 6) + (lhs >> 2));
  }
};

/// The process-wide NTT for (N, modulus); built on first use.  References stay valid: entries
/// are never erased and live in node storage.
inline NTT& GetNTT(size_t N, uint64_t modulus) {
  using Key = std::pair<uint64_t, uint64_t>;
  static std::unordered_map<Key, std::unique_ptr<NTT>, HashPair> cache;
  static RWLock guard;
  const Key key{static_cast<uint64_t>(N), modulus};
  {
    ReadLock shared = guard.AcquireRead();
    auto hit = cache.find(key);
    if (hit != cache.end()) return *hit->second;
  }
  // built outside the lock (table construction and upload take milliseconds); a thread that
  // lost the race drops its copy
  std::unique_ptr<NTT> fresh(new NTT(static_cast<uint64_t>(N), modulus));
  WriteLock exclusive = guard.AcquireWrite();
  auto& slot = cache[key];
  if (!slot) slot = std::move(fresh);
  return *slot;
}

}  // namespace hexl
}  // namespace intel
