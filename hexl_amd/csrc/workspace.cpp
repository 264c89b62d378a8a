// workspace.cpp -- see workspace.h
#include "workspace.h"

#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <utility>

namespace hexl_amd {
namespace {
struct Entry {
  void* ptr = nullptr;
  size_t cap = 0;
};
std::mutex g_mu;
std::map<std::tuple<int, hipStream_t, int>, Entry>& table() {
  static auto* t = new std::map<std::tuple<int, hipStream_t, int>, Entry>;  // never destroyed:
  return *t;  // HIP may already be torn down when static destructors run
}
std::map<std::pair<int, hipStream_t>, std::unique_ptr<std::mutex>>& sequence_table() {
  static auto* t = new std::map<std::pair<int, hipStream_t>, std::unique_ptr<std::mutex>>;
  return *t;
}
}  // namespace

StreamSequenceLock::StreamSequenceLock(hipStream_t stream) : mu_(nullptr) {
  int device = 0;
  (void)hipGetDevice(&device);
  std::mutex* mu;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto& slot = sequence_table()[std::make_pair(device, stream)];
    if (!slot) slot.reset(new std::mutex);
    mu = slot.get();  // entries are never erased: the pointer stays valid
  }
  mu->lock();
  mu_ = mu;
}

StreamSequenceLock::~StreamSequenceLock() { static_cast<std::mutex*>(mu_)->unlock(); }

hipError_t stream_workspace(WorkspacePurpose purpose, hipStream_t stream, size_t bytes,
                            void** out) {
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(g_mu);
  Entry& en = table()[std::make_tuple(device, stream, (int)purpose)];
  if (en.cap < bytes) {
    if (en.ptr) {
      // hipFree waits for the device, so no kernel still reads the old buffer
      e = hipFree(en.ptr);
      en.ptr = nullptr;
      en.cap = 0;
      if (e != hipSuccess) return e;
    }
    size_t want = bytes < 4096 ? 4096 : bytes;
    e = hipMalloc(&en.ptr, want);
    if (e != hipSuccess) {
      en.ptr = nullptr;
      return e;
    }
    en.cap = want;
  }
  *out = en.ptr;
  return hipSuccess;
}

void release_workspaces() {
  std::lock_guard<std::mutex> lock(g_mu);
  int prev = 0;
  (void)hipGetDevice(&prev);
  for (auto& kv : table()) {
    if (!kv.second.ptr) continue;
    (void)hipSetDevice(std::get<0>(kv.first));
    (void)hipFree(kv.second.ptr);
  }
  table().clear();
  (void)hipSetDevice(prev);
}

}  // namespace hexl_amd
