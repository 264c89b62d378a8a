// workspace.cpp -- see workspace.h
#include "workspace.h"

#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <utility>
#include <vector>

namespace hexl_amd {
namespace {
struct Entry {
  void* ptr = nullptr;
  size_t cap = 0;
};
// g_mu guards the two tables (lookup / insert / erase) and nothing else: no HIP call is made
// under it, so one stream growing its scratch (hipFree waits for the device) does not stall
// the enqueues of every other stream.
std::mutex g_mu;
std::map<std::tuple<int, hipStream_t, int>, Entry>& table() {
  static auto* t = new std::map<std::tuple<int, hipStream_t, int>, Entry>;  // never destroyed:
  return *t;  // HIP may already be torn down when static destructors run
}
// The per-(device, stream) sequence locks.  Recursive: a composite that holds the lock of its
// stream while it enqueues (KeySwitch) may call a transform that takes it again for its own
// scratch (the one-launch fused transform).
typedef std::recursive_mutex SeqMutex;
std::map<std::pair<int, hipStream_t>, std::shared_ptr<SeqMutex>>& sequence_table() {
  static auto* t = new std::map<std::pair<int, hipStream_t>, std::shared_ptr<SeqMutex>>;
  return *t;
}
// Replay cache (workspace.h).  Entries of one (device, stream) are only touched under that
// stream's sequence lock; g_mu covers the outer map.
struct GraphEntry {
  std::vector<uint64_t> key;
  hipGraphExec_t exec = nullptr;
  bool poisoned = false;
  uint64_t last_use = 0;
};
struct GraphSet {
  std::vector<GraphEntry> entries;
  uint64_t clock = 0;
};
std::map<std::pair<int, hipStream_t>, GraphSet>& graph_table() {
  static auto* t = new std::map<std::pair<int, hipStream_t>, GraphSet>;
  return *t;
}
GraphSet* graph_set(int device, hipStream_t stream, bool create) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto& tab = graph_table();
  auto it = tab.find(std::make_pair(device, stream));
  if (it != tab.end()) return &it->second;  // node addresses are stable
  if (!create) return nullptr;
  return &tab[std::make_pair(device, stream)];
}
// Under the stream's sequence lock.  The graphs may still be executing: wait for the stream.
void drop_graphs(int device, hipStream_t stream) {
  GraphSet* gs = graph_set(device, stream, false);
  if (!gs) return;
  bool waited = false;
  for (auto& en : gs->entries)
    if (en.exec) {
      if (!waited) (void)hipStreamSynchronize(stream);
      waited = true;
      (void)hipGraphExecDestroy(en.exec);
    }
  gs->entries.clear();
}

std::shared_ptr<SeqMutex> sequence_mutex(int device, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto& slot = sequence_table()[std::make_pair(device, stream)];
  if (!slot) slot = std::make_shared<SeqMutex>();
  return slot;
}
}  // namespace

struct StreamSequenceLock::Held {
  std::shared_ptr<SeqMutex> mu;  // keeps the mutex alive if its table entry is released meanwhile
};

StreamSequenceLock::StreamSequenceLock(hipStream_t stream) : held_(new Held) {
  int device = 0;
  (void)hipGetDevice(&device);
  held_->mu = sequence_mutex(device, stream);
  held_->mu->lock();
}

StreamSequenceLock::~StreamSequenceLock() {
  held_->mu->unlock();
  delete held_;
}

hipError_t stream_workspace(WorkspacePurpose purpose, hipStream_t stream, size_t bytes,
                            void** out) {
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  // Same-stream callers are serialised by the stream's sequence lock (taken here as well, it
  // is recursive), so the entry is ours while we grow it; g_mu only covers the map itself.
  StreamSequenceLock sequence(stream);
  Entry* en;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    en = &table()[std::make_tuple(device, stream, (int)purpose)];  // node addresses are stable
  }
  if (en->cap < bytes) {
    // (hipFree / hipMalloc are illegal under stream capture: the first call of a size class
    // must happen outside a capture, as tools/graph_replay.py does with its warm-up call)
    // captured sequences hold the old buffer's address
    drop_graphs(device, stream);
    if (en->ptr) {
      // hipFree waits for the device, so no kernel still reads the old buffer
      e = hipFree(en->ptr);
      en->ptr = nullptr;
      en->cap = 0;
      if (e != hipSuccess) return e;
    }
    size_t want = bytes < 4096 ? 4096 : bytes;
    e = hipMalloc(&en->ptr, want);
    if (e != hipSuccess) {
      en->ptr = nullptr;
      return e;
    }
    en->cap = want;
  }
  *out = en->ptr;
  return hipSuccess;
}

SequenceGraphAction lookup_sequence_graph(hipStream_t stream, const std::vector<uint64_t>& key,
                                          hipGraphExec_t* exec) {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return kGraphEager;
  GraphSet* gs = graph_set(device, stream, true);
  ++gs->clock;
  for (auto& en : gs->entries)
    if (en.key == key) {
      en.last_use = gs->clock;
      if (en.poisoned) return kGraphEager;
      if (en.exec) {
        *exec = en.exec;
        return kGraphReplay;
      }
      return kGraphCapture;  // second sight
    }
  // first sight: remember the key, evicting the least recently used one
  if ((int)gs->entries.size() >= kMaxSequenceGraphs) {
    size_t victim = 0;
    for (size_t i = 1; i < gs->entries.size(); ++i)
      if (gs->entries[i].last_use < gs->entries[victim].last_use) victim = i;
    if (gs->entries[victim].exec) {
      (void)hipStreamSynchronize(stream);  // it may still be executing
      (void)hipGraphExecDestroy(gs->entries[victim].exec);
    }
    gs->entries.erase(gs->entries.begin() + (long)victim);
  }
  GraphEntry en;
  en.key = key;
  en.last_use = gs->clock;
  gs->entries.push_back(std::move(en));
  return kGraphEager;
}

void store_sequence_graph(hipStream_t stream, const std::vector<uint64_t>& key, hipGraphExec_t exec) {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return;
  GraphSet* gs = graph_set(device, stream, true);
  for (auto& en : gs->entries)
    if (en.key == key) {
      if (en.exec && en.exec != exec) (void)hipGraphExecDestroy(en.exec);
      en.exec = exec;
      return;
    }
  (void)hipGraphExecDestroy(exec);  // the key was evicted meanwhile
}

void poison_sequence_graph(hipStream_t stream, const std::vector<uint64_t>& key) {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return;
  GraphSet* gs = graph_set(device, stream, true);
  for (auto& en : gs->entries)
    if (en.key == key) {
      if (en.exec) {
        (void)hipStreamSynchronize(stream);
        (void)hipGraphExecDestroy(en.exec);
      }
      en.exec = nullptr;
      en.poisoned = true;
    }
}

void release_stream_workspaces(hipStream_t stream) {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return;
  std::vector<void*> victims;
  {
    StreamSequenceLock sequence(stream);  // nobody is enqueueing against the buffers
    drop_graphs(device, stream);
    std::lock_guard<std::mutex> lock(g_mu);
    graph_table().erase(std::make_pair(device, stream));
    for (auto it = table().begin(); it != table().end();) {
      if (std::get<0>(it->first) == device && std::get<1>(it->first) == stream) {
        if (it->second.ptr) victims.push_back(it->second.ptr);
        it = table().erase(it);
      } else {
        ++it;
      }
    }
  }
  for (void* p : victims) (void)hipFree(p);  // waits for the device: queued kernels finish first
  // The stream's sequence mutex goes only when nobody else holds a reference to it (references
  // are taken under g_mu, so the count cannot rise while we look): a thread that holds or
  // waits for it keeps sequencing on the SAME mutex as everybody who comes after.
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = sequence_table().find(std::make_pair(device, stream));
  if (it != sequence_table().end() && it->second.use_count() == 1) sequence_table().erase(it);
}

int release_workspaces() {
  // Key by key under the stream's sequence lock (the order every user takes: sequence lock,
  // then g_mu): a composite call that is enqueueing against a buffer finishes its sequence
  // first; buffers whose stream is busy on ANOTHER thread right now are left alone and counted.
  std::vector<std::tuple<int, hipStream_t, int>> keys;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto& kv : table()) keys.push_back(kv.first);
  }
  std::vector<std::pair<int, void*>> victims;
  int busy = 0;
  for (auto& key : keys) {
    std::shared_ptr<SeqMutex> mu = sequence_mutex(std::get<0>(key), std::get<1>(key));
    if (!mu->try_lock()) {
      ++busy;
      continue;
    }
    {
      int cur = 0;
      (void)hipGetDevice(&cur);
      if (cur != std::get<0>(key)) (void)hipSetDevice(std::get<0>(key));
      drop_graphs(std::get<0>(key), std::get<1>(key));
      if (cur != std::get<0>(key)) (void)hipSetDevice(cur);
    }
    {
      std::lock_guard<std::mutex> lock(g_mu);
      graph_table().erase(std::make_pair(std::get<0>(key), std::get<1>(key)));
      auto it = table().find(key);
      if (it != table().end()) {
        if (it->second.ptr) victims.emplace_back(std::get<0>(key), it->second.ptr);
        table().erase(it);
      }
    }
    mu->unlock();
  }
  int prev = 0;
  (void)hipGetDevice(&prev);
  for (auto& v : victims) {
    (void)hipSetDevice(v.first);
    (void)hipFree(v.second);  // waits for the device: kernels already queued finish first
  }
  (void)hipSetDevice(prev);
  return busy;
}

}  // namespace hexl_amd
