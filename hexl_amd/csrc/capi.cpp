// capi.cpp -- the C-ABI of include/hexl_amd.h: plan construction (host table
// builder + upload), argument validation, device selection, host-pointer
// staging.  All compute is in ntt_kernels.hip / eltwise_kernels.hip; there is
// no CPU fallback -- if HIP is unusable every compute entry point fails.
#include <hip/hip_runtime_api.h>
#include <time.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hexl_amd.h"
#include "internal.h"
#include "number_theory.h"
#include "workspace.h"

using namespace hexl_amd;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  return fail(e == hipErrorNoDevice || e == hipErrorInvalidDevice ? HEXL_AMD_ERR_NO_DEVICE
                                                                   : HEXL_AMD_ERR_HIP,
              "%s: %s", what, hipGetErrorString(e));
}

#define HX_HIP(call)                                 \
  do {                                               \
    hipError_t e_ = (call);                          \
    if (e_ != hipSuccess) return hip_fail(e_, #call); \
  } while (0)

// Devices this process has used through the library (bit d = device d): what
// hexl_amd_host_unregister waits for -- synchronising EVERY visible device would create a primary
// context on each of them in every rank of a one-process-per-GPU job.
std::atomic<uint64_t> g_devices_used{0};
inline void note_device(int device) {
  if (device >= 0 && device < 64) g_devices_used.fetch_or(1ull << device, std::memory_order_relaxed);
}

// Makes `device` current for the scope if it is not already.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  DeviceScope() = default;
  explicit DeviceScope(int device) { enter(device); }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
  void enter(int device) {
    note_device(device);
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) {
      err = hipSetDevice(device);
      switched = (err == hipSuccess);
    }
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
};

// The device a launch on `st` runs on: the stream's own, or the calling thread's current
// device for the default stream(s).
hipError_t stream_device(hipStream_t st, int* device) {
  if (st == nullptr || st == hipStreamPerThread || st == hipStreamLegacy) return hipGetDevice(device);
  return hipStreamGetDevice(st, device);
}
// Makes the device that owns `st` current for the scope (every stream-taking entry point
// without a plan: a worker thread of a multi-GPU caller need not set a current device).
struct StreamDeviceScope : DeviceScope {
  int device = 0;
  explicit StreamDeviceScope(hipStream_t st) {
    err = stream_device(st, &device);
    if (err == hipSuccess) enter(device);
  }
};
#define HX_ON_STREAM_DEVICE(st)                          \
  StreamDeviceScope stream_scope_((hipStream_t)(st));    \
  if (stream_scope_.err != hipSuccess) return hip_fail(stream_scope_.err, "selecting the stream's device")

// Per-thread staging buffers for the host-pointer entry points.
// "host_poll": 1 (default) = a host-pointer call learns that its stream's work is done from a sequence
// number that a one-thread kernel behind the work stores into device-mapped host memory and the calling
// thread polls (Staging::finish); 0 = hipStreamSynchronize.  The runtime's completion path (signal,
// its bookkeeping on the host) costs 3-4 us more than the extra dispatch: N = 4096 one polynomial on
// mapped memory 17.2 -> 13.8 us, N = 16384 24.3 -> 21.1 (tools/completion_probe.hip; an in-kernel flag
// measured the same as the flag kernel).
std::atomic<u32> g_host_poll{1};
// hexl_amd_get_counter: waits that the flag ended / that ran out of polling time and went to the runtime
std::atomic<u64> g_host_polls{0}, g_host_poll_timeouts{0};
// Polls *flag until it has reached `seq` (sequence numbers only grow: wrap-safe comparison), for at most
// about a millisecond; false: not seen in that time (the caller then waits in the runtime).
static bool poll_completion_flag(const u32* flag, u32 seq) {
  timespec t0{};
  for (u32 spins = 1;; ++spins) {
    if ((int32_t)(__atomic_load_n(flag, __ATOMIC_ACQUIRE) - seq) >= 0) {
      g_host_polls.fetch_add(1, std::memory_order_relaxed);
      return true;
    }
    __builtin_ia32_pause();
    if ((spins & 0x3ff) == 0) {
      timespec t{};
      clock_gettime(CLOCK_MONOTONIC, &t);
      if (spins == 0x400) {
        t0 = t;
      } else if ((t.tv_sec - t0.tv_sec) * 1000000000ll + (t.tv_nsec - t0.tv_nsec) > 1000000) {
        g_host_poll_timeouts.fetch_add(1, std::memory_order_relaxed);
        return false;
      }
    }
  }
}

struct Staging {
  int device = -1;
  void* buf = nullptr;
  size_t cap = 0;
  hipStream_t stream = nullptr;
  // Small calls on ordinary host memory: a pinned, device-mapped bounce buffer.  The caller's
  // words are copied into it on the host (a 32 KiB memcpy is ~1 us), the one-kernel transform
  // or element-wise kernel runs straight on it over the link, the result is copied out:
  // one launch and one synchronisation instead of two staged hipMemcpy calls and a launch
  // (N = 4096: 34 -> ~21 us per call).
  void* bounce = nullptr;      // host address
  void* bounce_dev = nullptr;  // the address kernels use
  size_t bounce_cap = 0;
  // completion flag of `stream` (pinned, device-mapped) and the last sequence number asked for
  u32* done = nullptr;
  u32* done_dev = nullptr;
  u32 done_seq = 0;
  // Copies between ordinary (pageable) caller memory and the device go through these pinned
  // slots, never through the runtime's own handling of pageable memory: from about 1 MiB the HIP
  // runtime pins the CALLER's pages for the duration of a copy instead of staging them, and
  // that is the path the round-4 / round-5 aborts sat in -- a GPU memory access fault on an
  // address inside the process's heap, raised while a thread was inside such a copy
  // (EXPERIMENTS.md section 10).  staged_h2d / staged_d2h below.  Four slots, allocated on first
  // use and sized by the copies that come: 1 MiB each until a copy of 4 MiB or more is seen, 4 MiB
  // from then on (round 6: with 4 MiB pieces filled by the copy pool below the staged path moves
  // data at the link's rate; with two 1 MiB slots and one thread's memcpy it ran at half of it).
  static constexpr int kSlots = 4;
  static constexpr size_t kSmallSlot = (size_t)1 << 20, kBigSlot = (size_t)4 << 20;
  size_t slot_bytes = kSmallSlot;  // size of the slots allocated from now on (and of every live one)
  void* slot[kSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t slot_ev[kSlots] = {nullptr, nullptr, nullptr, nullptr};
  // A slot is free, or in use by a DMA whose end is marked by its event (chunks in the middle of
  // a pipeline) or simply by everything enqueued on a stream so far (the last chunk of a call:
  // the caller synchronises that stream anyway, no event needed).
  enum SlotState { kSlotFree = 0, kSlotEvent = 1, kSlotStream = 2 };
  SlotState slot_state[kSlots] = {kSlotFree, kSlotFree, kSlotFree, kSlotFree};
  hipStream_t slot_stream[kSlots] = {nullptr, nullptr, nullptr, nullptr};
  int slot_device = -1;  // the device the events belong to
  // Second stream + "kernels of chunk k done" events of the chunk pipeline of the *_host
  // transforms (ntt_host_pipelined): copies out run beside copies in.
  hipStream_t stream_out = nullptr;
  hipEvent_t chunk_ev[2] = {nullptr, nullptr};
  // A thread that ends gives its buffer and streams back (callers that run every task on
  // a fresh std::thread would otherwise leak one staging area per task).  Thread-local
  // destructors run when the thread ends and, for the main thread, at exit() BEFORE
  // static destructors and atexit handlers, i.e. while the HIP runtime is still up;
  // errors are ignored all the same.
  ~Staging() {
    if (stream) (void)hipStreamSynchronize(stream);
    // the composites' scratch keyed by these streams (KeySwitch through host pointers)
    if (device >= 0 && stream) {
      int cur = -1;
      if (hipGetDevice(&cur) == hipSuccess && (cur == device || hipSetDevice(device) == hipSuccess)) {
        if (stream) release_stream_workspaces(stream);
        if (cur != device && cur >= 0) (void)hipSetDevice(cur);
      }
    }
    if (buf) (void)hipFree(buf);
    if (bounce) (void)hipHostFree(bounce);
    if (done) (void)hipHostFree(done);
    for (int b = 0; b < kSlots; ++b) {
      (void)slot_free(b);
      if (slot_ev[b]) (void)hipEventDestroy(slot_ev[b]);
      if (slot[b]) (void)hipHostFree(slot[b]);
    }
    for (int b = 0; b < 2; ++b)
      if (chunk_ev[b]) (void)hipEventDestroy(chunk_ev[b]);
    if (stream_out) (void)hipStreamDestroy(stream_out);
    if (stream) (void)hipStreamDestroy(stream);
  }
  // (the current device is the one the copies' stream belongs to)  `bytes`: the copy at hand.
  int ensure_slots(size_t bytes) {
    int dev = 0;
    HX_HIP(hipGetDevice(&dev));
    if (bytes >= kBigSlot && slot_bytes < kBigSlot) {  // from now on: big slots
      for (int b = 0; b < kSlots; ++b) {
        (void)slot_free(b);
        if (slot[b]) HX_HIP(hipHostFree(slot[b]));
        slot[b] = nullptr;
      }
      slot_bytes = kBigSlot;
    }
    for (int b = 0; b < kSlots; ++b) {
      if (slot_ev[b] && slot_device != dev) {  // an event records on streams of its own device only
        (void)slot_free(b);
        (void)hipEventDestroy(slot_ev[b]);
        slot_ev[b] = nullptr;
      }
    }
    slot_device = dev;
    return HEXL_AMD_OK;
  }
  // slot b, allocated (with its event) on first use
  int slot_ready(int b) {
    if (!slot[b]) HX_HIP(hipHostMalloc(&slot[b], slot_bytes, hipHostMallocPortable));
    if (!slot_ev[b]) HX_HIP(hipEventCreateWithFlags(&slot_ev[b], hipEventDisableTiming));
    return HEXL_AMD_OK;
  }
  // waits until the DMA that last used slot b is done (the host may then read or write it)
  int slot_free(int b) {
    const SlotState was = slot_state[b];
    slot_state[b] = kSlotFree;
    if (was == kSlotEvent) HX_HIP(hipEventSynchronize(slot_ev[b]));
    if (was == kSlotStream) HX_HIP(hipStreamSynchronize(slot_stream[b]));
    return HEXL_AMD_OK;
  }
  // Returns when everything enqueued on st so far is done.  The thread's own stream: by polling the
  // completion flag ("host_poll"), for at most about a millisecond -- work that long is left to
  // hipStreamSynchronize, which is also what reports an error of the stream.  For calls whose last
  // operation on the stream is a KERNEL (the bounce buffer, mapped caller memory): behind a copy out
  // the flag kernel's dependency on the copy costs more than the runtime's wait (N = 65536 staged:
  // 84 -> 100 us when the copy's wait polled), so the staged paths keep hipStreamSynchronize.
  int finish(hipStream_t st) {
    if (st != nullptr && st == stream && g_host_poll.load(std::memory_order_relaxed) != 0) {
      if (!done) {
        // (explicitly fine-grained: the store must reach the host without waiting for a cache write-back)
        HX_HIP(hipHostMalloc((void**)&done, 64,
                             hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent));
        *done = 0;
        HX_HIP(hipHostGetDevicePointer((void**)&done_dev, done, 0));
      }
      const u32 seq = ++done_seq;
      if (completion_flag_launch(done_dev, seq, st) == hipSuccess) {
        if (poll_completion_flag(done, seq)) return HEXL_AMD_OK;
      } else {
        (void)hipGetLastError();
      }
    }
    HX_HIP(hipStreamSynchronize(st));
    return HEXL_AMD_OK;
  }
  // marks slot b in use by the DMA just enqueued on st; `last`: no more chunks follow in this call
  int slot_used(int b, hipStream_t st, bool last) {
    slot_stream[b] = st;
    // (only the thread's own staging stream: a caller's stream may be gone by the time the
    // slot is wanted again, an event outlives its stream)
    if (last && stream != nullptr && st == stream) {
      slot_state[b] = kSlotStream;
    } else {
      HX_HIP(hipEventRecord(slot_ev[b], st));
      slot_state[b] = kSlotEvent;
    }
    return HEXL_AMD_OK;
  }
  // pinned + mapped host memory of at least `bytes` (any device may address it: portable)
  int ensure_bounce(size_t bytes) {
    if (bounce_cap >= bytes) return HEXL_AMD_OK;
    if (bounce) HX_HIP(hipHostFree(bounce));
    bounce = nullptr;
    bounce_cap = 0;
    size_t want = bytes < (256u << 10) ? (256u << 10) : bytes;
    HX_HIP(hipHostMalloc(&bounce, want, hipHostMallocMapped | hipHostMallocPortable));
    HX_HIP(hipHostGetDevicePointer(&bounce_dev, bounce, 0));
    bounce_cap = want;
    return HEXL_AMD_OK;
  }
  // `dev` is the current device.
  int ensure(int dev, size_t bytes) {
    if (device != dev) {
      if (device >= 0 && (buf || stream)) {  // release what belongs to the previous device
        if (hipSetDevice(device) == hipSuccess) {
          for (int b = 0; b < kSlots; ++b) (void)slot_free(b);
          if (stream) (void)hipStreamSynchronize(stream);
          if (stream_out) (void)hipStreamSynchronize(stream_out);
          if (stream) release_stream_workspaces(stream);
          if (buf) (void)hipFree(buf);
          if (stream) (void)hipStreamDestroy(stream);
          if (stream_out) (void)hipStreamDestroy(stream_out);
          for (int b = 0; b < 2; ++b) {
            if (chunk_ev[b]) (void)hipEventDestroy(chunk_ev[b]);
            chunk_ev[b] = nullptr;
          }
        }
        HX_HIP(hipSetDevice(dev));
      }
      buf = nullptr;
      cap = 0;
      stream = nullptr;
      stream_out = nullptr;
      device = dev;
    }
    if (!stream) HX_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (cap < bytes) {
      if (buf) HX_HIP(hipFree(buf));
      buf = nullptr;
      cap = 0;
      HX_HIP(hipMalloc(&buf, bytes));
      cap = bytes;
    }
    return HEXL_AMD_OK;
  }
};
thread_local Staging g_staging;

// "host_direct_copy": 0 (default) = copies between ordinary host memory and the device go through
// the calling thread's pinned slots (staged_h2d / staged_d2h: a host copy per slot overlapped with
// the DMA of the others); 1 = they are handed to hipMemcpyAsync as they are (the runtime pins the
// caller's pages on the fly: for callers on a runtime they trust).
std::atomic<u32> g_host_direct_copy{0};

// Host-side copies between caller memory and the pinned slots.  One thread's memcpy moves
// 12-25 GB/s, the link 54 GB/s each way: copies of 1 MiB and more are cut into 256 KiB pieces that a
// small pool of helper threads (created at the first such copy, asleep otherwise, "host_copy_threads"
// of them including the caller: default 6, 1 = the calling thread alone) takes from a shared counter.
// One parallel copy at a time per process: a second calling thread that finds the pool busy copies
// by itself.  The reference library has no threads of its own; these only ever touch the
// caller's buffer between the call's entry and its return.
std::atomic<u32> g_host_copy_threads{6};
class CopyPool {
 public:
  static CopyPool& get() {
    static CopyPool pool;
    return pool;
  }
  // dst <- src, and (bytes2 != 0) dst2 <- src2 in the same job (a pipeline's copy in and copy out)
  void copy(void* dst, const void* src, size_t bytes, void* dst2 = nullptr, const void* src2 = nullptr,
            size_t bytes2 = 0) {
    const u32 want = g_host_copy_threads.load(std::memory_order_relaxed);
    if (bytes + bytes2 < kMinParallel || want <= 1 || !busy_.try_lock()) {
      memcpy(dst, src, bytes);
      if (bytes2) memcpy(dst2, src2, bytes2);
      return;
    }
    grow(want - 1);
    // (a job object of its own per copy: a helper that wakes late works on the counters of the
    // job it woke for, never on those of the next one)
    auto job = std::make_shared<Job>();
    job->dst[0] = (char*)dst;
    job->src[0] = (const char*)src;
    job->bytes[0] = bytes;
    job->dst[1] = (char*)dst2;
    job->src[1] = (const char*)src2;
    job->bytes[1] = bytes2;
    job->pieces0 = (bytes + kPiece - 1) / kPiece;
    job->pieces = job->pieces0 + (bytes2 + kPiece - 1) / kPiece;
    job->left.store(job->pieces, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = job;
      generation_.fetch_add(1, std::memory_order_release);
    }
    if (sleepers_.load(std::memory_order_acquire) != 0) cv_.notify_all();
    work(*job);
    // (the pieces other threads took: they are microseconds from done)
    while (job->left.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    busy_.unlock();
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_.store(true, std::memory_order_release);
    }
    cv_.notify_all();
    for (std::thread& t : threads_) t.join();
  }

 private:
  static constexpr size_t kPiece = (size_t)128 << 10, kMinParallel = (size_t)1 << 20;
  struct Job {
    char* dst[2] = {nullptr, nullptr};
    const char* src[2] = {nullptr, nullptr};
    size_t bytes[2] = {0, 0}, pieces0 = 0, pieces = 0;
    std::atomic<size_t> next{0}, left{0};
  };
  void grow(u32 helpers) {  // (under busy_)
    while (threads_.size() < helpers && threads_.size() < 63) threads_.emplace_back([this] { helper(); });
  }
  static void work(Job& j) {
    for (;;) {
      size_t i = j.next.fetch_add(1, std::memory_order_relaxed);
      if (i >= j.pieces) return;
      const int w = i >= j.pieces0 ? 1 : 0;
      if (w) i -= j.pieces0;
      const size_t off = i * kPiece, nb = j.bytes[w] - off < kPiece ? j.bytes[w] - off : kPiece;
      memcpy(j.dst[w] + off, j.src[w] + off, nb);
      j.left.fetch_sub(1, std::memory_order_release);
    }
  }
  // A helper that has just worked polls for the next job for a short while before it goes to
  // sleep: the copies of one call follow each other within tens of microseconds, and a wake-up
  // through the condition variable costs about as much as copying a piece.
  void helper() {
    uint64_t seen = 0;
    for (;;) {
      bool have = false;
      for (int spin = 0; spin < kSpins; ++spin) {
        if (generation_.load(std::memory_order_acquire) != seen || stop_.load(std::memory_order_acquire)) {
          have = true;
          break;
        }
        __builtin_ia32_pause();
      }
      std::shared_ptr<Job> job;
      {
        std::unique_lock<std::mutex> lk(m_);
        if (!have) {
          sleepers_.fetch_add(1, std::memory_order_release);
          cv_.wait(lk, [&] { return stop_.load() || generation_.load() != seen; });
          sleepers_.fetch_sub(1, std::memory_order_release);
        }
        if (stop_.load()) return;
        seen = generation_.load();
        job = job_;
      }
      if (job) work(*job);
    }
  }
  static constexpr int kSpins = 20000;  // ~100-200 us of polling
  std::mutex busy_, m_;
  std::condition_variable cv_;
  std::vector<std::thread> threads_;
  std::shared_ptr<Job> job_;
  std::atomic<uint64_t> generation_{0};
  std::atomic<int> sleepers_{0};
  std::atomic<bool> stop_{false};
};
inline void host_copy(void* dst, const void* src, size_t bytes) { CopyPool::get().copy(dst, src, bytes); }
// Piece size of a staged copy of `bytes` through slots of `slot_bytes`: one piece below 2 MiB (a
// piece costs a hipMemcpyAsync and an event, ~15 us), two pieces up to two slots' worth (the host
// copy of the second overlaps the DMA of the first: 4 MiB in 235 instead of 314 us), whole slots above.
inline size_t staged_piece(size_t bytes, size_t slot_bytes) {
  if (bytes < ((size_t)2 << 20)) return bytes < slot_bytes ? bytes : slot_bytes;
  const size_t half = ((bytes + 1) / 2 + 0xfff) & ~(size_t)0xfff;
  return half < slot_bytes ? half : slot_bytes;
}

// dst (device) <- src (ordinary host memory), enqueued on st.  Returns when the last chunk has
// been copied out of src (the DMA of the last chunks may still be running).
int staged_h2d(void* dst, const void* src, size_t bytes, hipStream_t st) {
  Staging& s = g_staging;
  if (int rc = s.ensure_slots(bytes)) return rc;
  const size_t piece = staged_piece(bytes, s.slot_bytes);
  int k = 0;
  for (size_t off = 0; off < bytes; ++k) {
    const int b = k % Staging::kSlots;
    const size_t nb = bytes - off < piece ? bytes - off : piece;
    if (int rc = s.slot_ready(b)) return rc;
    if (int rc = s.slot_free(b)) return rc;  // the host is about to write the slot
    host_copy(s.slot[b], (const char*)src + off, nb);
    HX_HIP(hipMemcpyAsync((char*)dst + off, s.slot[b], nb, hipMemcpyHostToDevice, st));
    off += nb;
    if (int rc = s.slot_used(b, st, off == bytes)) return rc;
  }
  return HEXL_AMD_OK;
}

// dst (ordinary host memory) <- src (device), after everything enqueued on st so far.
// Synchronous: the data is in dst on return.
int staged_d2h(void* dst, const void* src, size_t bytes, hipStream_t st) {
  Staging& s = g_staging;
  if (int rc = s.ensure_slots(bytes)) return rc;
  constexpr int kS = Staging::kSlots;
  size_t pend_off[kS] = {}, pend_nb[kS] = {};
  auto drain = [&](int b) -> int {
    if (pend_nb[b]) {
      if (int rc = s.slot_free(b)) return rc;
      host_copy((char*)dst + pend_off[b], s.slot[b], pend_nb[b]);
      pend_nb[b] = 0;
    }
    return HEXL_AMD_OK;
  };
  const size_t piece = staged_piece(bytes, s.slot_bytes);
  int k = 0;
  for (size_t off = 0; off < bytes; ++k) {
    const int b = k % kS;
    const size_t nb = bytes - off < piece ? bytes - off : piece;
    if (int rc = s.slot_ready(b)) return rc;
    if (int rc = drain(b)) return rc;
    // the DEVICE is about to write the slot: a DMA still reading it on this very stream (an
    // earlier staged_h2d of the same call) is ordered before it by the stream itself
    if (s.slot_state[b] != Staging::kSlotFree && s.slot_stream[b] != st)
      if (int rc = s.slot_free(b)) return rc;
    HX_HIP(hipMemcpyAsync(s.slot[b], (const char*)src + off, nb, hipMemcpyDeviceToHost, st));
    pend_off[b] = off;
    pend_nb[b] = nb;
    off += nb;
    if (int rc = s.slot_used(b, st, off == bytes)) return rc;
  }
  for (int i = 0; i < kS; ++i)  // the oldest chunk first
    if (int rc = drain((k + i) % kS)) return rc;
  return HEXL_AMD_OK;
}

// One side of a host-pointer call: `host` is caller memory whose first byte is of `kind`
// (pointer_kind: 0 ordinary host, 1 device, 2 pinned mapped host).
int copy_to_device(void* dst, const void* host, size_t bytes, int kind, hipStream_t st) {
  if (kind == 0 && g_host_direct_copy.load() == 0) return staged_h2d(dst, host, bytes, st);
  HX_HIP(hipMemcpyAsync(dst, host, bytes, hipMemcpyDefault, st));
  return HEXL_AMD_OK;
}
int copy_from_device(void* host, const void* src, size_t bytes, int kind, hipStream_t st) {
  if (kind == 0 && g_host_direct_copy.load() == 0) return staged_d2h(host, src, bytes, st);
  HX_HIP(hipMemcpyAsync(host, src, bytes, hipMemcpyDefault, st));
  return HEXL_AMD_OK;
}

}  // namespace

struct hexl_amd_ntt {
  u64 n = 0, q = 0, w = 0;
  u32 log_n = 0;
  int device = 0;
  // Reference-layout host tables (hexl_amd_ntt_table): built lazily except 0/2/3/6.
  mutable std::vector<u64> host[7];
  mutable std::once_flag once[7];
  ulonglong2* d_fwd = nullptr;
  ulonglong2* d_inv = nullptr;
  PlanDev* d_self = nullptr;  // device copy of the kernel parameters (multi-plan launches)
  NttTables t{};
};

extern "C" {

const char* hexl_amd_last_error(void) { return g_last_error.c_str(); }

int hexl_amd_device_count(int* count) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (count) *count = (e == hipSuccess) ? c : 0;
  if (e != hipSuccess || c == 0) return fail(HEXL_AMD_ERR_NO_DEVICE, "no HIP device visible");
  return HEXL_AMD_OK;
}

// 0 = ordinary host memory (or unknown), 1 = device / managed memory, 2 = pinned host memory
// mapped into the device's address space (hipHostMalloc / hipHostRegister: kernels can read
// and write it over the link).  *dev_alias: the address kernels use (kinds 1 and 2).
static int pointer_kind(const void* p, void** dev_alias, int* owner = nullptr) {
  if (dev_alias) *dev_alias = nullptr;
  if (owner) *owner = -1;
  if (!p) return 0;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // unregistered host memory reports an error: clear it
    return 0;
  }
  if (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) {
    if (dev_alias) *dev_alias = const_cast<void*>(p);
    if (owner) *owner = attr.device;  // the device the allocation lives on
    return 1;
  }
  if (attr.type == hipMemoryTypeHost && attr.devicePointer) {
    if (dev_alias) {
      // (attr.devicePointer is the alias of the address that was asked about)
      *dev_alias = attr.devicePointer;
    }
    return 2;
  }
  return 0;
}

// Kind of the whole range [p, p + bytes): 2 only if its LAST byte is mapped too and maps to
// the alias of the first plus the same offset (a buffer that starts inside a registered pool
// and runs past its end, or straddles two registrations, is ordinary host memory to the
// zero-copy paths: they stage it instead of faulting on the device); 1 likewise for device
// memory; otherwise 0.  `first` is the kind of the first byte alone (0 = the host can
// dereference it: what the bounce path asks).  One runtime query for ordinary host memory, two
// for device / mapped memory.
struct RangeKind {
  int kind = 0;
  int first = 0;
  void* alias = nullptr;
};
static RangeKind classify_range(const void* p, size_t bytes) {
  RangeKind r;
  void* last = nullptr;
  r.first = r.kind = pointer_kind(p, &r.alias);
  if (r.kind != 0 && bytes > 1) {
    const int kl = pointer_kind((const char*)p + bytes - 1, &last);
    if (kl != r.kind || (char*)last - (char*)r.alias != (ptrdiff_t)(bytes - 1)) {
      r.kind = 0;
      r.alias = nullptr;
      // (starts in mapped host memory and leaves it: host memory all the same -- the copies
      // treat it as ordinary, i.e. it goes through the pinned slots like any pageable buffer)
      if (r.first == 2) r.first = 0;
    }
  }
  return r;
}

int hexl_amd_pointer_is_device(const void* p) { return pointer_kind(p, nullptr) == 1 ? 1 : 0; }
int hexl_amd_pointer_kind(const void* p) { return pointer_kind(p, nullptr); }

int hexl_amd_host_alloc(void** p, uint64_t bytes) {
  if (!p) return fail(HEXL_AMD_ERR_INVALID_ARG, "p == nullptr");
  *p = nullptr;
  if (bytes == 0) return HEXL_AMD_OK;
  HX_HIP(hipHostMalloc(p, (size_t)bytes, hipHostMallocMapped | hipHostMallocPortable));
  return HEXL_AMD_OK;
}
int hexl_amd_host_free(void* p) {
  if (!p) return HEXL_AMD_OK;
  HX_HIP(hipHostFree(p));
  return HEXL_AMD_OK;
}
int hexl_amd_host_register(void* p, uint64_t bytes) {
  if (!p || bytes == 0) return fail(HEXL_AMD_ERR_INVALID_ARG, "p == nullptr or bytes == 0");
  HX_HIP(hipHostRegister(p, (size_t)bytes, hipHostRegisterMapped | hipHostRegisterPortable));
  return HEXL_AMD_OK;
}
int hexl_amd_host_unregister(void* p) {
  if (!p) return HEXL_AMD_OK;
  // Nothing may still be reading or writing the range when its device mapping goes away: the
  // *_host entry points return only after their own work is done, but the caller may have handed
  // the mapped alias to kernels or copies of its own on any stream of any device.
  // Only the devices this process has used through the library, plus the current one, are waited
  // for (a caller that handed the alias to a device it drives entirely by itself synchronises that
  // device itself): touching every visible device would create a context on each.
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess) {
    note_device(cur);
    const uint64_t used = g_devices_used.load(std::memory_order_relaxed);
    for (int d = 0; d < 64; ++d) {
      if (!((used >> d) & 1)) continue;
      if (d == cur || hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
    }
    if (hipSetDevice(cur) != hipSuccess) return fail(HEXL_AMD_ERR_HIP, "restoring the current device failed");
  }
  HX_HIP(hipHostUnregister(p));
  return HEXL_AMD_OK;
}

int hexl_amd_device_alloc(void** p, uint64_t bytes, int device) {
  if (!p) return fail(HEXL_AMD_ERR_INVALID_ARG, "p == nullptr");
  *p = nullptr;
  if (bytes == 0) return HEXL_AMD_OK;
  if (device < 0) HX_HIP(hipGetDevice(&device));
  DeviceScope scope(device);
  if (scope.err != hipSuccess) return hip_fail(scope.err, "hipSetDevice");
  HX_HIP(hipMalloc(p, (size_t)bytes));
  return HEXL_AMD_OK;
}
int hexl_amd_device_free(void* p) {
  if (!p) return HEXL_AMD_OK;
  HX_HIP(hipFree(p));
  return HEXL_AMD_OK;
}
int hexl_amd_copy(void* dst, const void* src, uint64_t bytes, void* stream, int blocking) {
  if (bytes == 0) return HEXL_AMD_OK;
  if (!dst || !src) return fail(HEXL_AMD_ERR_INVALID_ARG, "dst == nullptr or src == nullptr");
  HX_ON_STREAM_DEVICE(stream);
  // ordinary host memory on one side, device memory on the other: through the thread's pinned
  // slots (the runtime is never handed pageable caller memory; such a copy has always been
  // synchronous on the host side, and the device-to-host form now completes before it returns)
  const int kd = pointer_kind(dst, nullptr), ks = pointer_kind(src, nullptr);
  if (kd == 1 && ks == 0) {
    if (int rc = copy_to_device(dst, src, (size_t)bytes, 0, (hipStream_t)stream)) return rc;
  } else if (kd == 0 && ks == 1) {
    if (int rc = copy_from_device(dst, src, (size_t)bytes, 0, (hipStream_t)stream)) return rc;
  } else {
    HX_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDefault, (hipStream_t)stream));
  }
  if (blocking) HX_HIP(hipStreamSynchronize((hipStream_t)stream));
  return HEXL_AMD_OK;
}
int hexl_amd_set_device(int device) {
  HX_HIP(hipSetDevice(device));
  return HEXL_AMD_OK;
}
int hexl_amd_get_device(int* device) {
  if (!device) return fail(HEXL_AMD_ERR_INVALID_ARG, "device == nullptr");
  HX_HIP(hipGetDevice(device));
  return HEXL_AMD_OK;
}
// Streams of the library's own making (hexl_amd_stream_create): hexl_amd_synchronize learns that such a
// stream is done the way the host-pointer calls do -- a one-thread kernel behind the work publishes a
// sequence number in device-mapped host memory, the caller polls ("host_poll"; 3-4 us earlier than
// hipStreamSynchronize returns).  The flag of a stream is made at its first synchronisation.
namespace {
struct StreamFlag {
  std::mutex mu;  // sequence number and launch of one synchronisation: the flag only ever grows
  u32* host = nullptr;
  u32* dev = nullptr;
  u32 seq = 0;
};
std::mutex g_stream_flags_mu;
std::map<hipStream_t, std::shared_ptr<StreamFlag>> g_stream_flags;
}  // namespace

int hexl_amd_stream_create(void** stream, int device) {
  if (!stream) return fail(HEXL_AMD_ERR_INVALID_ARG, "stream == nullptr");
  *stream = nullptr;
  if (device < 0) HX_HIP(hipGetDevice(&device));
  DeviceScope scope(device);
  if (scope.err != hipSuccess) return hip_fail(scope.err, "hipSetDevice");
  hipStream_t st = nullptr;
  HX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  {
    std::lock_guard<std::mutex> lock(g_stream_flags_mu);
    g_stream_flags[st] = std::make_shared<StreamFlag>();
  }
  *stream = (void*)st;
  return HEXL_AMD_OK;
}
int hexl_amd_stream_destroy(void* stream) {
  if (!stream) return HEXL_AMD_OK;
  HX_ON_STREAM_DEVICE(stream);
  HX_HIP(hipStreamSynchronize((hipStream_t)stream));
  release_stream_workspaces((hipStream_t)stream);
  std::shared_ptr<StreamFlag> flag;
  {
    std::lock_guard<std::mutex> lock(g_stream_flags_mu);
    auto it = g_stream_flags.find((hipStream_t)stream);
    if (it != g_stream_flags.end()) {
      flag = it->second;
      g_stream_flags.erase(it);
    }
  }
  if (flag && flag->host) (void)hipHostFree(flag->host);
  HX_HIP(hipStreamDestroy((hipStream_t)stream));
  return HEXL_AMD_OK;
}
int hexl_amd_synchronize(void* stream) {
  if (!stream) {
    HX_HIP(hipDeviceSynchronize());
    return HEXL_AMD_OK;
  }
  std::shared_ptr<StreamFlag> flag;
  if (g_host_poll.load(std::memory_order_relaxed) != 0) {
    std::lock_guard<std::mutex> lock(g_stream_flags_mu);
    auto it = g_stream_flags.find((hipStream_t)stream);
    if (it != g_stream_flags.end()) flag = it->second;
  }
  if (flag) {
    HX_ON_STREAM_DEVICE(stream);
    u32 seq = 0;
    bool launched = false;
    {
      std::lock_guard<std::mutex> lock(flag->mu);
      if (!flag->host) {
        u32* h = nullptr;
        if (hipHostMalloc((void**)&h, 64, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) ==
            hipSuccess) {
          *h = 0;
          if (hipHostGetDevicePointer((void**)&flag->dev, h, 0) == hipSuccess) flag->host = h;
          else (void)hipHostFree(h);
        }
        if (!flag->host) (void)hipGetLastError();
      }
      if (flag->host) {
        seq = ++flag->seq;
        launched = completion_flag_launch(flag->dev, seq, (hipStream_t)stream) == hipSuccess;
        if (!launched) (void)hipGetLastError();
      }
    }
    if (launched && poll_completion_flag(flag->host, seq)) return HEXL_AMD_OK;
  }
  HX_HIP(hipStreamSynchronize((hipStream_t)stream));
  return HEXL_AMD_OK;
}

int hexl_amd_check_bounds(const uint64_t* data, uint64_t n, uint64_t bound,
                          uint64_t* violations) {
  if (!violations) return fail(HEXL_AMD_ERR_INVALID_ARG, "violations == nullptr");
  *violations = 0;
  if (n == 0) return HEXL_AMD_OK;
  if (!data) return fail(HEXL_AMD_ERR_INVALID_ARG, "data == nullptr");
  void* alias = nullptr;
  int owner = -1;
  const int kind = pointer_kind(data, &alias, &owner);
  if (kind == 0) {  // the caller's own host buffer: validating it is not computing on it
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i) bad += data[i] >= bound;
    *violations = bad;
    return HEXL_AMD_OK;
  }
  // the reduction kernel runs on the device that owns the data (mapped host memory: the
  // calling thread's current device)
  int device = 0;
  HX_HIP(hipGetDevice(&device));
  if (kind == 1 && owner >= 0) device = owner;
  DeviceScope scope(device);
  if (scope.err != hipSuccess) return hip_fail(scope.err, "hipSetDevice");
  if (int rc = g_staging.ensure(device, sizeof(unsigned long long))) return rc;
  unsigned long long* counter = (unsigned long long*)g_staging.buf;
  hipStream_t st = g_staging.stream;
  // the data may still be being produced on another stream: the check is a debug aid and
  // synchronises the device first
  HX_HIP(hipDeviceSynchronize());
  HX_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned long long), st));
  hipError_t e = count_out_of_bounds_launch((const u64*)alias, n, bound, counter, st);
  if (e != hipSuccess) return hip_fail(e, "bounds check launch");
  unsigned long long bad = 0;
  HX_HIP(hipMemcpyAsync(&bad, counter, sizeof bad, hipMemcpyDeviceToHost, st));
  HX_HIP(hipStreamSynchronize(st));
  *violations = bad;
  return HEXL_AMD_OK;
}

}  // extern "C"

// Plan tables go to the device through the calling thread's pinned slots like every other copy
// from pageable memory (staged_h2d): the library never hands its own pageable memory to the
// runtime either.  One-off cost per plan: a memcpy of the tables.
static hipError_t upload_table(void* dst, const void* src, size_t bytes, int device) {
  if (g_staging.ensure(device, 8) != HEXL_AMD_OK) return hipErrorOutOfMemory;
  if (staged_h2d(dst, src, bytes, g_staging.stream) != HEXL_AMD_OK) return hipErrorUnknown;
  return hipStreamSynchronize(g_staging.stream);
}

extern "C" {

int hexl_amd_ntt_check_arguments(uint64_t degree, uint64_t modulus) {
  return nt::ntt_check_arguments(degree, modulus) ? 1 : 0;
}

int hexl_amd_ntt_create(hexl_amd_ntt** out, uint64_t degree, uint64_t modulus,
                        uint64_t root_of_unity, int device) {
  if (!out) return fail(HEXL_AMD_ERR_INVALID_ARG, "plan == nullptr");
  *out = nullptr;
  if (!nt::ntt_check_arguments(degree, modulus))
    return fail(HEXL_AMD_ERR_INVALID_ARG,
                "degree %llu / modulus %llu: need a power-of-two degree in [2, 2^20] and a "
                "prime modulus == 1 mod 2*degree below 2^62",
                (unsigned long long)degree, (unsigned long long)modulus);
  const u64 n = degree, q = modulus;
  u64 w = root_of_unity;
  if (w == 0) w = nt::minimal_primitive_root(2 * n, q);
  if (!nt::is_primitive_root(w, 2 * n, q))
    return fail(HEXL_AMD_ERR_INVALID_ARG, "%llu is not a primitive 2*%llu-th root of unity",
                (unsigned long long)w, (unsigned long long)n);
  if (device < 0) HX_HIP(hipGetDevice(&device));
  int ndev = 0;
  HX_HIP(hipGetDeviceCount(&ndev));
  if (device >= ndev) return fail(HEXL_AMD_ERR_NO_DEVICE, "device %d of %d", device, ndev);

  hexl_amd_ntt* p = new (std::nothrow) hexl_amd_ntt;
  if (!p) return fail(HEXL_AMD_ERR_ALLOC, "out of host memory");
  p->n = n;
  p->q = q;
  p->w = w;
  p->log_n = (u32)nt::log2_floor(n);
  p->device = device;

  // R[bitrev(i)] = w^i and Rinv[bitrev(i)] = w^-i (hexl/ntt/ntt-internal.cpp:60-72;
  // the reference inverts each power with Euclid, powers of w^-1 give the same values).
  const u64 w_inv = nt::inverse_mod(w, q);
  std::vector<u64>& R = p->host[0];
  std::vector<u64> Rinv(n);
  R.assign(n, 0);
  {
    u64 pw = 1, pwi = 1;
    for (u64 i = 0; i < n; ++i) {
      const u64 idx = nt::reverse_bits(i, p->log_n);
      R[idx] = pw;
      Rinv[idx] = pwi;
      pw = nt::multiply_mod(pw, w, q);
      pwi = nt::multiply_mod(pwi, w_inv, q);
    }
  }
  // Stage-ordered inverse table of the reference (ntt-internal.cpp:143-154).
  std::vector<u64>& IR = p->host[3];
  IR.assign(n, 0);
  IR[0] = 1;
  {
    u64 idx = 1;
    for (u64 m = n >> 1; m > 0; m >>= 1)
      for (u64 i = 0; i < m; ++i) IR[idx++] = Rinv[m + i];
  }
  p->host[2].resize(n);
  p->host[6].resize(n);
  for (u64 i = 0; i < n; ++i) {
    p->host[2][i] = nt::multiply_factor(R[i], 64, q);
    p->host[6][i] = nt::multiply_factor(IR[i], 64, q);
  }
  // Device tables: heap-ordered (value, Shoup factor) pairs -- the factor has 32 / 63 / 64
  // fractional bits under the Small / Lazy and Harvey60 / Strict arithmetic policy -- or, under Fp64,
  // one double per twiddle: the value balanced into (-q/2, q/2].
  const int policy = choose_policy(q);
  // (doubled values: the Lazy family and Harvey60)
  const bool doubled = policy == kPolicyLazy || policy == kPolicyLazy32 || policy == kPolicyLazy16 ||
                       policy == kPolicyHarvey60;
  const u64 shoup_bits = policy == kPolicySmall ? 32 : doubled ? 63 : 64;
  const bool fp_tables = policy == kPolicyFp64 || policy == kPolicyFp64L;  // one double per twiddle
  auto balanced = [q](u64 w) { return w > q / 2 ? -(double)(q - w) : (double)w; };
  auto bits_of = [](double d) {
    u64 b;
    memcpy(&b, &d, sizeof b);
    return b;
  };
  const size_t entry = fp_tables ? sizeof(double) : sizeof(ulonglong2);
  std::vector<u64> hf, hi;  // raw table words
  hf.reserve(n * entry / 8);
  hi.reserve(n * entry / 8);
  for (u64 i = 0; i < n; ++i) {
    if (fp_tables) {
      hf.push_back(bits_of(balanced(R[i])));
      hi.push_back(bits_of(balanced(Rinv[i])));
    } else {
      hf.push_back(R[i]);
      hf.push_back(nt::multiply_factor(R[i], shoup_bits, q));
      hi.push_back(Rinv[i]);
      hi.push_back(nt::multiply_factor(Rinv[i], shoup_bits, q));
    }
  }
  InvLast il{};
  // the multiply-free N^-1 scaling of the last stage's sum branch (modarith.h)
  il.c2 = (q - 1) >> p->log_n;
  il.log_n = p->log_n;
  il.mont_mask = (u32)((doubled ? 2 * n : n) - 1);
  il.n1 = nt::inverse_mod(n, q);
  il.n1w = nt::multiply_mod(il.n1, Rinv[1], q);
  if (fp_tables) {
    il.n1 = bits_of(balanced(il.n1));
    il.n1w = bits_of(balanced(il.n1w));
    il.n1p = il.n1wp = 0;
  } else {
    il.n1p = nt::multiply_factor(il.n1, shoup_bits, q);
    il.n1wp = nt::multiply_factor(il.n1w, shoup_bits, q);
  }

  DeviceScope scope(device);
  hipError_t e = scope.err;
  if (e == hipSuccess) e = hipMalloc((void**)&p->d_fwd, n * entry);
  if (e == hipSuccess) e = hipMalloc((void**)&p->d_inv, n * entry);
  if (e == hipSuccess) e = upload_table(p->d_fwd, hf.data(), n * entry, device);
  if (e == hipSuccess) e = upload_table(p->d_inv, hi.data(), n * entry, device);
  if (e != hipSuccess) {
    if (p->d_fwd) (void)hipFree(p->d_fwd);
    if (p->d_inv) (void)hipFree(p->d_inv);
    delete p;
    return hip_fail(e, "uploading NTT tables");
  }
  p->t.policy = policy;
  p->t.fwd = p->d_fwd;
  p->t.inv = p->d_inv;
  p->t.mod = make_mod_const(q);
  p->t.log_n = p->log_n;
  p->t.inv_last = il;
  {
    PlanDev pd{p->d_fwd, p->d_inv, p->t.mod, il};
    e = hipMalloc((void**)&p->d_self, sizeof(PlanDev));
    if (e == hipSuccess) e = hipMemcpy(p->d_self, &pd, sizeof pd, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(p->d_fwd);
      (void)hipFree(p->d_inv);
      if (p->d_self) (void)hipFree(p->d_self);
      delete p;
      return hip_fail(e, "uploading NTT plan parameters");
    }
    p->t.dev = p->d_self;
  }
  *out = p;
  return HEXL_AMD_OK;
}

int hexl_amd_ntt_destroy(hexl_amd_ntt* p) {
  if (!p) return HEXL_AMD_OK;
  {
    DeviceScope scope(p->device);
    if (p->d_fwd) (void)hipFree(p->d_fwd);
    if (p->d_inv) (void)hipFree(p->d_inv);
    if (p->d_self) (void)hipFree(p->d_self);
  }
  delete p;
  return HEXL_AMD_OK;
}

uint64_t hexl_amd_ntt_degree(const hexl_amd_ntt* p) { return p ? p->n : 0; }
uint64_t hexl_amd_ntt_modulus(const hexl_amd_ntt* p) { return p ? p->q : 0; }
uint64_t hexl_amd_ntt_root_of_unity(const hexl_amd_ntt* p) { return p ? p->w : 0; }
int hexl_amd_ntt_device(const hexl_amd_ntt* p) { return p ? p->device : -1; }

const uint64_t* hexl_amd_ntt_table(const hexl_amd_ntt* p, int which) {
  if (!p || which < 0 || which > 6) return nullptr;
  // 1, 4, 5: 32- / 52-bit preconditioned variants, rarely wanted -> lazy.
  auto lazy = [&](int idx, int src, u64 shift) {
    std::call_once(p->once[idx], [&] {
      p->host[idx].resize(p->n);
      for (u64 i = 0; i < p->n; ++i)
        p->host[idx][i] = nt::multiply_factor(p->host[src][i], shift, p->q);
    });
  };
  if (which == 1) lazy(1, 0, 32);
  if (which == 4) lazy(4, 3, 32);
  if (which == 5) lazy(5, 3, 52);
  return p->host[which].data();
}

static int check_ntt_args(const hexl_amd_ntt* p, const void* result, const void* operand,
                          bool forward, uint64_t in_mf, uint64_t out_mf) {
  if (!p) return fail(HEXL_AMD_ERR_INVALID_ARG, "plan == nullptr");
  if (!result) return fail(HEXL_AMD_ERR_INVALID_ARG, "result == nullptr");
  if (!operand) return fail(HEXL_AMD_ERR_INVALID_ARG, "operand == nullptr");
  if (forward) {
    // ntt-internal.cpp:193-197
    if (!(in_mf == 1 || in_mf == 2 || in_mf == 4))
      return fail(HEXL_AMD_ERR_INVALID_ARG, "input_mod_factor must be 1, 2 or 4; got %llu",
                  (unsigned long long)in_mf);
    if (!(out_mf == 1 || out_mf == 4))
      return fail(HEXL_AMD_ERR_INVALID_ARG, "output_mod_factor must be 1 or 4; got %llu",
                  (unsigned long long)out_mf);
  } else {
    // ntt-internal.cpp:257-260
    if (!(in_mf == 1 || in_mf == 2))
      return fail(HEXL_AMD_ERR_INVALID_ARG, "input_mod_factor must be 1 or 2; got %llu",
                  (unsigned long long)in_mf);
    if (!(out_mf == 1 || out_mf == 2))
      return fail(HEXL_AMD_ERR_INVALID_ARG, "output_mod_factor must be 1 or 2; got %llu",
                  (unsigned long long)out_mf);
  }
  return HEXL_AMD_OK;
}

static int ntt_run(const hexl_amd_ntt* p, uint64_t* result, const uint64_t* operand,
                   uint64_t batch, bool forward, uint64_t in_mf, uint64_t out_mf,
                   void* stream) {
  if (int rc = check_ntt_args(p, result, operand, forward, in_mf, out_mf)) return rc;
  DeviceScope scope(p->device);
  if (scope.err != hipSuccess) return hip_fail(scope.err, "hipSetDevice");
  hipError_t e = forward ? ntt_forward_launch(p->t, result, operand, batch, out_mf,
                                              (hipStream_t)stream)
                         : ntt_inverse_launch(p->t, result, operand, batch, out_mf,
                                              (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, forward ? "forward NTT launch" : "inverse NTT launch");
  return HEXL_AMD_OK;
}

int hexl_amd_ntt_forward(const hexl_amd_ntt* p, uint64_t* result, const uint64_t* operand,
                         uint64_t batch, uint64_t in_mf, uint64_t out_mf, void* stream) {
  return ntt_run(p, result, operand, batch, true, in_mf, out_mf, stream);
}

int hexl_amd_ntt_inverse(const hexl_amd_ntt* p, uint64_t* result, const uint64_t* operand,
                         uint64_t batch, uint64_t in_mf, uint64_t out_mf, void* stream) {
  return ntt_run(p, result, operand, batch, false, in_mf, out_mf, stream);
}

// Polynomial i of `polys` uses plans[tab[(i / inner) % period]] (hexl_amd_ntt_*_map).
static int ntt_run_map(const hexl_amd_ntt* const* plans, uint64_t num_plans, const uint8_t* tab,
                       uint64_t period, uint64_t inner, uint64_t* result, const uint64_t* operand,
                       uint64_t polys, bool forward, uint64_t in_mf, uint64_t out_mf,
                       void* stream) {
  if (!plans || num_plans == 0) return fail(HEXL_AMD_ERR_INVALID_ARG, "no plans");
  if (!tab || period == 0 || inner == 0)
    return fail(HEXL_AMD_ERR_INVALID_ARG, "plan_of_slot == nullptr, period == 0 or inner == 0");
  for (uint64_t k = 0; k < num_plans; ++k) {
    if (!plans[k]) return fail(HEXL_AMD_ERR_INVALID_ARG, "plans[%llu] == nullptr",
                               (unsigned long long)k);
    if (plans[k]->n != plans[0]->n || plans[k]->device != plans[0]->device)
      return fail(HEXL_AMD_ERR_INVALID_ARG, "plans must share degree and device");
  }
  for (uint64_t s = 0; s < period; ++s)
    if (tab[s] >= num_plans)
      return fail(HEXL_AMD_ERR_INVALID_ARG, "plan_of_slot[%llu] = %u is not a plan index",
                  (unsigned long long)s, (unsigned)tab[s]);
  if (polys == 0) return HEXL_AMD_OK;
  if (int rc = check_ntt_args(plans[0], result, operand, forward, in_mf, out_mf)) return rc;
  const uint64_t n = plans[0]->n;
  // One launch sequence for the whole batch where the multi-plan kernels cover the shape.
  // Chunks of < 2^31 polynomials, cut at multiples of inner * period so that the map keeps
  // its phase.
  const uint64_t cycle = inner <= (1ull << 31) / period ? inner * period : 0;  // 0: too long
  if (num_plans <= (uint64_t)kMaxMultiPlans && period <= (uint64_t)kMaxMultiPeriod &&
      inner < (1ull << 31) && (polys < (1ull << 31) || (cycle && cycle < (1ull << 30)))) {
    DeviceScope scope(plans[0]->device);
    if (scope.err != hipSuccess) return hip_fail(scope.err, "hipSetDevice");
    const NttTables* tabs[kMaxMultiPlans];
    MultiMap map{};
    map.inner = (u32)inner;
    map.period = (u32)period;
    for (uint64_t k = 0; k < num_plans; ++k) tabs[k] = &plans[k]->t;
    for (uint64_t s = 0; s < period; ++s) map.plan_tab[s] = tab[s];
    const uint64_t chunk = polys < (1ull << 31) ? polys : ((1ull << 31) - 1) / cycle * cycle;
    bool refused = false;
    for (uint64_t off = 0; off < polys && !refused; off += chunk) {
      const uint64_t cnt = polys - off < chunk ? polys - off : chunk;
      const hipError_t e = ntt_multi_launch(forward, tabs, (u32)num_plans, map, cnt,
                                            result + off * n, operand + off * n, out_mf,
                                            (hipStream_t)stream);
      // (ntt_multi_launch validates the whole sequence -- every pass, every arithmetic
      // policy -- before its first launch: "not supported" means nothing was enqueued, and
      // the answer is the same for every chunk of the call)
      if (e == hipErrorNotSupported && off == 0)
        refused = true;
      else if (e != hipSuccess)
        return hip_fail(e, "multi-plan NTT launch");
    }
    if (!refused) return HEXL_AMD_OK;
  }
  // Run by run: maximal runs of consecutive polynomials that share a plan.
  uint64_t start = 0;
  while (start < polys) {
    const uint8_t k = tab[(start / inner) % period];
    uint64_t end = (start / inner + 1) * inner;
    while (end < polys && tab[(end / inner) % period] == k) end += inner;
    if (end > polys) end = polys;
    if (int rc = ntt_run(plans[k], result + start * n, operand + start * n, end - start, forward,
                         in_mf, out_mf, stream))
      return rc;
    start = end;
  }
  return HEXL_AMD_OK;
}

// prime_index[i] of polynomial i -> the (inner, period, table) form, or run by run.
static int ntt_run_indexed(const hexl_amd_ntt* const* plans, uint64_t num_plans,
                           const uint32_t* idx, uint64_t* result, const uint64_t* operand,
                           uint64_t polys, bool forward, uint64_t in_mf, uint64_t out_mf,
                           void* stream) {
  if (!plans || num_plans == 0) return fail(HEXL_AMD_ERR_INVALID_ARG, "no plans");
  if (polys == 0) return HEXL_AMD_OK;
  if (!idx) return fail(HEXL_AMD_ERR_INVALID_ARG, "prime_index == nullptr");
  for (uint64_t i = 0; i < polys; ++i)
    if (idx[i] >= num_plans)
      return fail(HEXL_AMD_ERR_INVALID_ARG, "prime_index[%llu] = %u is not a plan index",
                  (unsigned long long)i, (unsigned)idx[i]);
  // inner = gcd of the run lengths (the last run may be cut short by the end of the batch)
  auto gcd = [](uint64_t a, uint64_t b) {
    while (b) {
      const uint64_t t = a % b;
      a = b;
      b = t;
    }
    return a;
  };
  // (every complete run -- all but the last, which the end of the batch may cut short -- is a
  // whole number of slots; runs therefore start at multiples of inner and slots are uniform)
  uint64_t inner = 0, run_start = 0;
  for (uint64_t i = 1; i < polys; ++i)
    if (idx[i] != idx[i - 1]) {
      inner = gcd(inner, i - run_start);
      run_start = i;
    }
  if (inner == 0) inner = polys;  // one run
  const uint64_t slots = (polys + inner - 1) / inner;
  // shortest period of the slot sequence
  uint64_t period = 0;
  const uint64_t max_period = slots < (uint64_t)kMaxMultiPeriod ? slots : (uint64_t)kMaxMultiPeriod;
  for (uint64_t p = 1; p <= max_period && !period; ++p) {
    bool ok = true;
    for (uint64_t s = p; s < slots && ok; ++s) ok = idx[s * inner] == idx[(s - p) * inner];
    if (ok) period = p;
  }
  if (period && num_plans <= 256) {
    uint8_t tab[kMaxMultiPeriod];
    for (uint64_t s = 0; s < period; ++s) tab[s] = (uint8_t)idx[s * inner];
    return ntt_run_map(plans, num_plans, tab, period, inner, result, operand, polys, forward,
                       in_mf, out_mf, stream);
  }
  // no periodic structure: run by run
  for (uint64_t k = 0; k < num_plans; ++k) {
    if (!plans[k]) return fail(HEXL_AMD_ERR_INVALID_ARG, "plans[%llu] == nullptr",
                               (unsigned long long)k);
    if (plans[k]->n != plans[0]->n || plans[k]->device != plans[0]->device)
      return fail(HEXL_AMD_ERR_INVALID_ARG, "plans must share degree and device");
  }
  const uint64_t n = plans[0]->n;
  for (uint64_t start = 0; start < polys;) {
    uint64_t end = start + 1;
    while (end < polys && idx[end] == idx[start]) ++end;
    if (int rc = ntt_run(plans[idx[start]], result + start * n, operand + start * n, end - start,
                         forward, in_mf, out_mf, stream))
      return rc;
    start = end;
  }
  return HEXL_AMD_OK;
}

// Prime-major blocks: polynomials [k * batch_per_plan, (k+1) * batch_per_plan) belong to
// plans[k] -- the map form with inner = batch_per_plan and the identity table.
static int ntt_run_rns(const hexl_amd_ntt* const* plans, uint64_t num_plans, uint64_t* result,
                       const uint64_t* operand, uint64_t batch_per_plan, bool forward,
                       uint64_t in_mf, uint64_t out_mf, void* stream) {
  if (!plans) return fail(HEXL_AMD_ERR_INVALID_ARG, "plans == nullptr");
  for (uint64_t k = 0; k < num_plans; ++k) {
    if (!plans[k]) return fail(HEXL_AMD_ERR_INVALID_ARG, "plans[%llu] == nullptr",
                               (unsigned long long)k);
    if (plans[k]->n != plans[0]->n || plans[k]->device != plans[0]->device)
      return fail(HEXL_AMD_ERR_INVALID_ARG, "plans must share degree and device");
  }
  if (num_plans == 0 || batch_per_plan == 0) return HEXL_AMD_OK;
  if (int rc = check_ntt_args(plans[0], result, operand, forward, in_mf, out_mf)) return rc;
  // One launch sequence over all moduli (groups of kMaxMultiPlans) through the map form.
  // With many polynomials per modulus the launches no longer matter and the single-plan
  // kernels are ~5 % faster (their plan is in the kernel arguments): plan by plan from 2^24
  // coefficients per modulus.
  if (num_plans > 1 && batch_per_plan * plans[0]->n < (1ull << 24)) {
    uint8_t identity[kMaxMultiPlans];
    for (int j = 0; j < kMaxMultiPlans; ++j) identity[j] = (uint8_t)j;
    for (uint64_t k = 0; k < num_plans; k += (uint64_t)kMaxMultiPlans) {
      const uint64_t cnt = num_plans - k < (uint64_t)kMaxMultiPlans ? num_plans - k
                                                                     : (uint64_t)kMaxMultiPlans;
      const uint64_t off = k * batch_per_plan * plans[0]->n;
      if (int rc = ntt_run_map(plans + k, cnt, identity, cnt, batch_per_plan, result + off,
                               operand + off, cnt * batch_per_plan, forward, in_mf, out_mf, stream))
        return rc;
    }
    return HEXL_AMD_OK;
  }
  for (uint64_t k = 0; k < num_plans; ++k) {
    const uint64_t off = k * batch_per_plan * plans[k]->n;
    if (int rc = ntt_run(plans[k], result + off, operand + off, batch_per_plan, forward, in_mf,
                         out_mf, stream))
      return rc;
  }
  return HEXL_AMD_OK;
}

int hexl_amd_ntt_forward_map(const hexl_amd_ntt* const* plans, uint64_t num_plans,
                             const uint8_t* plan_of_slot, uint64_t period, uint64_t inner,
                             uint64_t* result, const uint64_t* operand, uint64_t polys,
                             uint64_t in_mf, uint64_t out_mf, void* stream) {
  return ntt_run_map(plans, num_plans, plan_of_slot, period, inner, result, operand, polys, true,
                     in_mf, out_mf, stream);
}
int hexl_amd_ntt_inverse_map(const hexl_amd_ntt* const* plans, uint64_t num_plans,
                             const uint8_t* plan_of_slot, uint64_t period, uint64_t inner,
                             uint64_t* result, const uint64_t* operand, uint64_t polys,
                             uint64_t in_mf, uint64_t out_mf, void* stream) {
  return ntt_run_map(plans, num_plans, plan_of_slot, period, inner, result, operand, polys, false,
                     in_mf, out_mf, stream);
}
int hexl_amd_ntt_forward_indexed(const hexl_amd_ntt* const* plans, uint64_t num_plans,
                                 const uint32_t* prime_index, uint64_t* result,
                                 const uint64_t* operand, uint64_t polys, uint64_t in_mf,
                                 uint64_t out_mf, void* stream) {
  return ntt_run_indexed(plans, num_plans, prime_index, result, operand, polys, true, in_mf,
                         out_mf, stream);
}
int hexl_amd_ntt_inverse_indexed(const hexl_amd_ntt* const* plans, uint64_t num_plans,
                                 const uint32_t* prime_index, uint64_t* result,
                                 const uint64_t* operand, uint64_t polys, uint64_t in_mf,
                                 uint64_t out_mf, void* stream) {
  return ntt_run_indexed(plans, num_plans, prime_index, result, operand, polys, false, in_mf,
                         out_mf, stream);
}

int hexl_amd_ntt_forward_rns(const hexl_amd_ntt* const* plans, uint64_t num_plans,
                             uint64_t* result, const uint64_t* operand,
                             uint64_t batch_per_plan, uint64_t in_mf, uint64_t out_mf,
                             void* stream) {
  return ntt_run_rns(plans, num_plans, result, operand, batch_per_plan, true, in_mf, out_mf,
                     stream);
}

int hexl_amd_ntt_inverse_rns(const hexl_amd_ntt* const* plans, uint64_t num_plans,
                             uint64_t* result, const uint64_t* operand,
                             uint64_t batch_per_plan, uint64_t in_mf, uint64_t out_mf,
                             void* stream) {
  return ntt_run_rns(plans, num_plans, result, operand, batch_per_plan, false, in_mf, out_mf,
                     stream);
}

// Host buffers (what an unmodified intel::hexl caller hands over): stage to the device,
// transform, copy back.  Ordinary memory moves through the thread's pinned slots (copy_to_device /
// copy_from_device above): the link's rate up to 1 MiB, a memcpy's (26 GB/s in + out) beyond --
// "host_direct_copy" = 1 hands the buffers to the runtime instead, which runs at the link's
// 54 GB/s from pageable memory as from pinned (tools/pcie_probe.py, profiles/r5_host_copy_ab.txt).
// A two-stream chunked pipeline over caller pages pinned for the call was built in round 2,
// measured equal to the plain sequence and -- never on by default, never exercised -- removed
// in round 5.
// Knob of the host path (hexl_amd_set_tuning; no environment variable is read):
// largest call (bytes of operand) that goes through the mapped bounce buffer: beyond it the
// host-side copies cost as much as the DMA they replace (round 4 measured N = 65536, 512 KiB: 82 us
// against 88 staged; N = 131072: 185 against 144 -- with the completion flag polled, round 6: N = 65536
// 74 against 83 staged, EltwiseMultMod of 65536 words 66-70 against 91, of 131072 words 145 either
// way: the limit went from 256 to 512 KiB).  What the buffer costs beyond the two memcpys is the
// device reading lines the CPU has just written, out of its caches: +1 us at 32 KiB, +4 at 128 KiB,
// +17 at 512 KiB against memory only the device writes (tests/cpp/host_call_budget.cpp: `unaccounted`).
// "host_bounce_kb"; 0 switches the bounce path off.
static std::atomic<size_t> g_host_bounce_max_bytes{(size_t)512 << 10};
static size_t host_bounce_max_bytes() { return g_host_bounce_max_bytes.load(); }
// "ks_graph": 1 (default) = a KeySwitch of at most kKsGraphMaxTargets targets whose buffers, keys
// and moduli were seen before on the same stream is replayed from a captured HIP graph; 0 = the
// launches are always enqueued one by one.
static std::atomic<u32> g_ks_graph{1};
// "ks_fuse": 1 (default) = the rounding and finish stages of KeySwitch ride on the load / store of
// the forward transform between them (round 6), 0 = stage by stage (A/B).
static std::atomic<u32> g_ks_fuse{1};
// "ks_mac_onestep": 1 (default) = the multiply-accumulate of KeySwitch reduces a 128-bit sum below
// 2^(bits(q) + 61) in one generalised Barrett step, 0 = always as BarrettReduce128's two-word form (A/B).
static std::atomic<u32> g_ks_mac_onestep{1};
// Bumped by every hexl_amd_set_tuning call: part of the key of a captured KeySwitch sequence, so a
// graph captured under other tuning values (another kernel selection) is never replayed.
static std::atomic<u64> g_tuning_epoch{0};
constexpr int kKsGraphMaxTargets = 4;
// hexl_amd_get_counter
static std::atomic<u64> g_ks_graph_captures{0}, g_ks_graph_replays{0}, g_ks_eager{0};
static bool set_host_tuning(const char* key, uint64_t value) {
  if (strcmp(key, "host_bounce_kb") == 0 && value <= (1u << 20)) {
    g_host_bounce_max_bytes = (size_t)value << 10;
    return true;
  }
  if (strcmp(key, "ks_graph") == 0 && value <= 1) {
    g_ks_graph = (u32)value;
    return true;
  }
  if (strcmp(key, "ks_fuse") == 0 && value <= 1) {
    g_ks_fuse = (u32)value;
    return true;
  }
  if (strcmp(key, "ks_mac_onestep") == 0 && value <= 1) {
    g_ks_mac_onestep = (u32)value;
    return true;
  }
  if (strcmp(key, "host_copy_threads") == 0 && value >= 1 && value <= 64) {
    g_host_copy_threads = (u32)value;
    return true;
  }
  if (strcmp(key, "host_direct_copy") == 0 && value <= 1) {
    g_host_direct_copy = (u32)value;
    return true;
  }
  if (strcmp(key, "host_poll") == 0 && value <= 1) {
    g_host_poll = (u32)value;
    return true;
  }
  return false;
}

static int ntt_run_host(const hexl_amd_ntt* p, uint64_t* result, const uint64_t* operand,
                        uint64_t batch, bool forward, uint64_t in_mf, uint64_t out_mf) {
  if (int rc = check_ntt_args(p, result, operand, forward, in_mf, out_mf)) return rc;
  if (batch == 0) return HEXL_AMD_OK;
  DeviceScope scope(p->device);
  if (scope.err != hipSuccess) return hip_fail(scope.err, "hipSetDevice");
  const size_t poly_bytes = (size_t)p->n * sizeof(u64);
  const size_t bytes = (size_t)batch * poly_bytes;
  auto run = [&](u64* d, u64 polys, hipStream_t st, u64* mid = nullptr) {
    return forward ? ntt_forward_launch(p->t, d, d, polys, out_mf, st, mid)
                   : ntt_inverse_launch(p->t, d, d, polys, out_mf, st, mid);
  };
  // Caller memory the kernels can address (pinned and mapped: hexl_amd_host_alloc /
  // hexl_amd_host_register, the intel::hexl allocator built on them): a one-kernel transform
  // reads the operand and writes the result straight over the link -- no staging copies, one
  // launch, one synchronisation; a multi-pass transform reads the operand in its first pass
  // (no H2D copy) and copies the result back.
  // (one classification per distinct buffer: in place is the common call)
  const RangeKind op_range = classify_range(operand, bytes);
  const RangeKind res_range =
      (const void*)result == (const void*)operand ? op_range : classify_range(result, bytes);
  // (in place or disjoint only: a result that overlaps the operand at an offset -- not a call the
  // reference defines either -- keeps the plain sequence, which reads all of the operand first)
  const bool same_or_disjoint = (const void*)result == (const void*)operand ||
                                (const char*)result + bytes <= (const char*)operand ||
                                (const char*)operand + bytes <= (const char*)result;
  void *op_dev = op_range.alias, *res_dev = res_range.alias;
  const int op_kind = op_range.kind, res_kind = res_range.kind;
  if (op_kind == 2) {
    if (int rc = g_staging.ensure(p->device, 8)) return rc;  // (the stream)
    hipStream_t st = g_staging.stream;
    auto launch = [&](u64* dst, u64* mid = nullptr) {
      return forward ? ntt_forward_launch(p->t, dst, (const u64*)op_dev, batch, out_mf, st, mid)
                     : ntt_inverse_launch(p->t, dst, (const u64*)op_dev, batch, out_mf, st, mid);
    };
    if (res_kind == 2 && ntt_is_single_kernel(p->t, batch, true)) {
      hipError_t e = launch((u64*)res_dev);
      if (e != hipSuccess) return hip_fail(e, "NTT launch");
      if (int rc = g_staging.finish(st)) return rc;
      return HEXL_AMD_OK;
    }
    if (res_kind == 2 && ntt_is_two_pass(p->t, batch, true)) {
      // Two passes, both buffers mapped: the first pass reads the operand over the link and hands over
      // in a DEVICE buffer, the second reads that and writes the result over the link -- every word
      // crosses the link once per direction, no copy (round 6; before: pass 1 into the device buffer,
      // pass 2 there, a D2H copy -- or, for small calls, both passes in place over the link:
      // N = 16384 one polynomial 21.1 us, N = 65536 48.6)
      if (int rc = g_staging.ensure(p->device, bytes)) return rc;
      hipError_t e = launch((u64*)res_dev, (u64*)g_staging.buf);
      if (e != hipSuccess) return hip_fail(e, "NTT launch");
      if (int rc = g_staging.finish(st)) return rc;
      return HEXL_AMD_OK;
    }
    if (int rc = g_staging.ensure(p->device, bytes)) return rc;
    u64* d = (u64*)g_staging.buf;
    hipError_t e = launch(d);
    if (e != hipSuccess) return hip_fail(e, "NTT launch");
    if (int rc = copy_from_device(result, d, bytes, res_range.first, st)) return rc;
    HX_HIP(hipStreamSynchronize(st));
    return HEXL_AMD_OK;
  }
  if (op_kind == 0 && res_kind == 0 && bytes <= host_bounce_max_bytes() &&
      op_range.first != 1 && res_range.first != 1) {
    // ordinary host memory, small call: the kernels (one or two passes) run in place on the
    // mapped bounce buffer
    // (a two-pass transform hands over between its passes in device memory: the words cross the link
    // once per direction)
    const bool two_pass = ntt_is_two_pass(p->t, batch, true);
    if (int rc = g_staging.ensure(p->device, two_pass ? bytes : 8)) return rc;  // (the stream)
    if (int rc = g_staging.ensure_bounce(bytes)) return rc;
    hipStream_t st = g_staging.stream;
    memcpy(g_staging.bounce, operand, bytes);
    hipError_t e = run((u64*)g_staging.bounce_dev, batch, st, two_pass ? (u64*)g_staging.buf : nullptr);
    if (e != hipSuccess) return hip_fail(e, "NTT launch");
    if (int rc = g_staging.finish(st)) return rc;
    memcpy(result, g_staging.bounce, bytes);
    return HEXL_AMD_OK;
  }
  if (int rc = g_staging.ensure(p->device, bytes)) return rc;
  u64* d = (u64*)g_staging.buf;
  hipStream_t st = g_staging.stream;
  if (op_range.first == 0 && res_range.first == 0 && batch > 1 && bytes >= 2 * Staging::kBigSlot &&
      same_or_disjoint && g_host_direct_copy.load() == 0) {
    // Ordinary host memory on both sides, several polynomials, 8 MiB or more: a pipeline over
    // chunks of about 4 MiB of whole polynomials (round 6) --
    //   host copy in (k) + out (k - 2)  |  H2D, kernels of chunk k on one stream  |  D2H on another
    // so that the link carries both directions at once and the host copies hide behind it.
    // Slots 0, 1 carry the chunks in, slots 2, 3 the chunks out.
    Staging& s = g_staging;
    if (int rc = s.ensure_slots(bytes)) return rc;
    if (!s.stream_out) HX_HIP(hipStreamCreateWithFlags(&s.stream_out, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b)
      if (!s.chunk_ev[b]) HX_HIP(hipEventCreateWithFlags(&s.chunk_ev[b], hipEventDisableTiming));
    for (int b = 0; b < Staging::kSlots; ++b)
      if (int rc = s.slot_ready(b)) return rc;
    const u64 chunk_polys = s.slot_bytes / poly_bytes ? s.slot_bytes / poly_bytes : 0;
    if (chunk_polys >= 1) {
      const u64 nchunks = (batch + chunk_polys - 1) / chunk_polys;
      auto first = [&](u64 k) { return k * chunk_polys; };
      auto count = [&](u64 k) { return first(k) + chunk_polys <= batch ? chunk_polys : batch - first(k); };
      for (u64 k = 0; k < nchunks + 2; ++k) {
        // host side of this step, as ONE job of the copy pool: chunk k into its slot, chunk k - 2
        // out of the slot that chunk k's D2H is about to take (its own D2H was enqueued two steps
        // ago: waiting for it does not stall the step)
        const bool in = k < nchunks, out = k >= 2;
        const int b = (int)(k & 1), ob = 2 + b;
        const size_t off = in ? (size_t)first(k) * p->n : 0, nb = in ? (size_t)count(k) * poly_bytes : 0;
        const u64 j = out ? k - 2 : 0;
        const size_t joff = out ? (size_t)first(j) * p->n : 0, jnb = out ? (size_t)count(j) * poly_bytes : 0;
        if (in)
          if (int rc = s.slot_free(b)) return rc;  // (the H2D of chunk k - 2 is done)
        if (out)
          if (int rc = s.slot_free(ob)) return rc;  // (the D2H of chunk k - 2 is done)
        if (in && out)
          CopyPool::get().copy(s.slot[b], operand + off, nb, result + joff, s.slot[ob], jnb);
        else if (in)
          host_copy(s.slot[b], operand + off, nb);
        else
          host_copy(result + joff, s.slot[ob], jnb);
        if (in) {
          HX_HIP(hipMemcpyAsync(d + off, s.slot[b], nb, hipMemcpyHostToDevice, st));
          if (int rc = s.slot_used(b, st, false)) return rc;
          hipError_t e = run(d + off, count(k), st);
          if (e != hipSuccess) return hip_fail(e, "NTT launch");
          HX_HIP(hipEventRecord(s.chunk_ev[b], st));
          HX_HIP(hipStreamWaitEvent(s.stream_out, s.chunk_ev[b], 0));
          HX_HIP(hipMemcpyAsync(s.slot[ob], d + off, nb, hipMemcpyDeviceToHost, s.stream_out));
          if (int rc = s.slot_used(ob, s.stream_out, false)) return rc;
        }
      }
      HX_HIP(hipStreamSynchronize(st));
      return HEXL_AMD_OK;
    }
  }
  // (a mixed argument set -- device operand, host result -- lands here too: each side by its kind)
  if (int rc = copy_to_device(d, operand, bytes, op_range.first, st)) return rc;
  hipError_t e = run(d, batch, st);
  if (e != hipSuccess) return hip_fail(e, "NTT launch");
  if (int rc = copy_from_device(result, d, bytes, res_range.first, st)) return rc;
  HX_HIP(hipStreamSynchronize(st));
  return HEXL_AMD_OK;
}

int hexl_amd_ntt_forward_host(const hexl_amd_ntt* p, uint64_t* result, const uint64_t* operand,
                              uint64_t batch, uint64_t in_mf, uint64_t out_mf) {
  return ntt_run_host(p, result, operand, batch, true, in_mf, out_mf);
}

int hexl_amd_ntt_inverse_host(const hexl_amd_ntt* p, uint64_t* result, const uint64_t* operand,
                              uint64_t batch, uint64_t in_mf, uint64_t out_mf) {
  return ntt_run_host(p, result, operand, batch, false, in_mf, out_mf);
}

// ------------------------------------------------------------------ eltwise

static int check_elt(EltOp op, const EltArgs& g) {
  if (!g.result) return fail(HEXL_AMD_ERR_INVALID_ARG, "result == nullptr");
  if (!g.a) return fail(HEXL_AMD_ERR_INVALID_ARG, "operand1 == nullptr");
  if ((op == ELT_ADD || op == ELT_SUB || op == ELT_MULT) && !g.b)
    return fail(HEXL_AMD_ERR_INVALID_ARG, "operand2 == nullptr");
  if (g.n == 0) return fail(HEXL_AMD_ERR_INVALID_ARG, "n == 0");
  if (op != ELT_CMP_ADD && g.q <= 1)
    return fail(HEXL_AMD_ERR_INVALID_ARG, "modulus must be > 1");
  switch (op) {
    case ELT_ADD:
    case ELT_SUB:
    case ELT_ADD_SCALAR:
    case ELT_SUB_SCALAR:
      if (g.q >= (1ull << 63)) return fail(HEXL_AMD_ERR_INVALID_ARG, "modulus must be < 2^63");
      if ((op == ELT_ADD_SCALAR || op == ELT_SUB_SCALAR) && g.scalar >= g.q)
        return fail(HEXL_AMD_ERR_INVALID_ARG, "scalar operand must be < modulus");
      break;
    case ELT_MULT:
      if (!(g.in_mf == 1 || g.in_mf == 2 || g.in_mf == 4))
        return fail(HEXL_AMD_ERR_INVALID_ARG, "input_mod_factor must be 1, 2 or 4");
      if (g.q >= (1ull << 62) || g.in_mf * g.q >= (1ull << 63))
        return fail(HEXL_AMD_ERR_INVALID_ARG,
                    "need modulus < 2^62 and input_mod_factor * modulus < 2^63");
      break;
    case ELT_FMA:
      if (!(g.in_mf == 1 || g.in_mf == 2 || g.in_mf == 4 || g.in_mf == 8))
        return fail(HEXL_AMD_ERR_INVALID_ARG, "input_mod_factor must be 1, 2, 4 or 8");
      if (g.q >= (1ull << 61)) return fail(HEXL_AMD_ERR_INVALID_ARG, "modulus must be < 2^61");
      // eltwise-fma-mod.cpp:29-31 (HEXL_CHECK in debug builds): arg2 < input_mod_factor * q
      if (g.scalar >= g.in_mf * g.q)
        return fail(HEXL_AMD_ERR_INVALID_ARG, "arg2 must be < input_mod_factor * modulus");
      break;
    case ELT_REDUCE:
      if (!(g.in_mf == g.q || g.in_mf == 2 || g.in_mf == 4))
        return fail(HEXL_AMD_ERR_INVALID_ARG, "input_mod_factor must be modulus, 2 or 4");
      if (!(g.out_mf == 1 || g.out_mf == 2))
        return fail(HEXL_AMD_ERR_INVALID_ARG, "output_mod_factor must be 1 or 2");
      break;
    case ELT_REDUCE_FMA:
      if (g.q >= (1ull << 61)) return fail(HEXL_AMD_ERR_INVALID_ARG, "modulus must be < 2^61");
      break;
    case ELT_CMP_ADD:
    case ELT_CMP_SUB_MOD:
      if (g.cmp < 0 || g.cmp > 7) return fail(HEXL_AMD_ERR_INVALID_ARG, "cmp must be a CMPINT (0..7)");
      if (g.scalar == 0) return fail(HEXL_AMD_ERR_INVALID_ARG, "diff == 0");
      if (op == ELT_CMP_SUB_MOD && g.scalar >= g.q)
        return fail(HEXL_AMD_ERR_INVALID_ARG, "diff must be < modulus");
      break;
  }
  return HEXL_AMD_OK;
}

static int elt_run(EltOp op, const EltArgs& g, void* stream) {
  if (int rc = check_elt(op, g)) return rc;
  HX_ON_STREAM_DEVICE(stream);
  hipError_t e = eltwise_launch(op, g, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "eltwise launch");
  return HEXL_AMD_OK;
}

int hexl_amd_eltwise_add_mod(uint64_t* result, const uint64_t* a, const uint64_t* b, uint64_t n,
                             uint64_t q, void* stream) {
  return elt_run(ELT_ADD, EltArgs{result, a, b, 0, n, q, 1, 1}, stream);
}
int hexl_amd_eltwise_add_mod_scalar(uint64_t* result, const uint64_t* a, uint64_t b, uint64_t n,
                                    uint64_t q, void* stream) {
  return elt_run(ELT_ADD_SCALAR, EltArgs{result, a, nullptr, b, n, q, 1, 1}, stream);
}
int hexl_amd_eltwise_sub_mod(uint64_t* result, const uint64_t* a, const uint64_t* b, uint64_t n,
                             uint64_t q, void* stream) {
  return elt_run(ELT_SUB, EltArgs{result, a, b, 0, n, q, 1, 1}, stream);
}
int hexl_amd_eltwise_sub_mod_scalar(uint64_t* result, const uint64_t* a, uint64_t b, uint64_t n,
                                    uint64_t q, void* stream) {
  return elt_run(ELT_SUB_SCALAR, EltArgs{result, a, nullptr, b, n, q, 1, 1}, stream);
}
int hexl_amd_eltwise_mult_mod(uint64_t* result, const uint64_t* a, const uint64_t* b,
                              uint64_t n, uint64_t q, uint64_t in_mf, void* stream) {
  return elt_run(ELT_MULT, EltArgs{result, a, b, 0, n, q, in_mf, 1}, stream);
}
int hexl_amd_eltwise_fma_mod(uint64_t* result, const uint64_t* arg1, uint64_t arg2,
                             const uint64_t* arg3, uint64_t n, uint64_t q, uint64_t in_mf,
                             void* stream) {
  return elt_run(ELT_FMA, EltArgs{result, arg1, arg3, arg2, n, q, in_mf, 1}, stream);
}
int hexl_amd_eltwise_reduce_mod(uint64_t* result, const uint64_t* operand, uint64_t n,
                                uint64_t q, uint64_t in_mf, uint64_t out_mf, void* stream) {
  return elt_run(ELT_REDUCE, EltArgs{result, operand, nullptr, 0, n, q, in_mf, out_mf}, stream);
}
int hexl_amd_eltwise_reduce_fma_mod(uint64_t* result, const uint64_t* arg1, uint64_t arg2,
                                    const uint64_t* arg3, uint64_t n, uint64_t q,
                                    uint64_t in_mf, void* stream) {
  if (in_mf == q)
    return elt_run(ELT_REDUCE_FMA, EltArgs{result, arg1, arg3, arg2, n, q, in_mf, 1}, stream);
  return elt_run(ELT_FMA, EltArgs{result, arg1, arg3, arg2, n, q, in_mf, 1}, stream);
}

static EltArgs cmp_args(uint64_t* result, const uint64_t* a, uint64_t n, uint64_t q, int cmp,
                        uint64_t bound, uint64_t diff) {
  EltArgs g{result, a, nullptr, diff, n, q, 1, 1};
  g.cmp = cmp;
  g.bound = bound;
  return g;
}
int hexl_amd_eltwise_cmp_add(uint64_t* result, const uint64_t* operand1, uint64_t n, int cmp,
                             uint64_t bound, uint64_t diff, void* stream) {
  return elt_run(ELT_CMP_ADD, cmp_args(result, operand1, n, 0, cmp, bound, diff), stream);
}
int hexl_amd_eltwise_cmp_sub_mod(uint64_t* result, const uint64_t* operand1, uint64_t n,
                                 uint64_t q, int cmp, uint64_t bound, uint64_t diff,
                                 void* stream) {
  return elt_run(ELT_CMP_SUB_MOD, cmp_args(result, operand1, n, q, cmp, bound, diff), stream);
}

static int eltwise_host_run(EltOp op, EltArgs g, uint64_t* result, const uint64_t* operand1,
                            const uint64_t* operand2);

int hexl_amd_eltwise_cmp_host(uint64_t* result, const uint64_t* operand1, uint64_t n,
                              uint64_t q, int cmp, uint64_t bound, uint64_t diff) {
  const EltOp op = q == 0 ? ELT_CMP_ADD : ELT_CMP_SUB_MOD;
  return eltwise_host_run(op, cmp_args(result, operand1, n, q, cmp, bound, diff), result,
                          operand1, nullptr);
}

int hexl_amd_eltwise_host(int op, uint64_t* result, const uint64_t* operand1,
                          const uint64_t* operand2, uint64_t scalar, uint64_t n, uint64_t q,
                          uint64_t in_mf, uint64_t out_mf) {
  if (op < 0 || op > 6) return fail(HEXL_AMD_ERR_INVALID_ARG, "unknown eltwise op %d", op);
  return eltwise_host_run((EltOp)op, EltArgs{result, operand1, operand2, scalar, n, q, in_mf, out_mf},
                          result, operand1, operand2);
}

// host buffers: stage to the device, run the kernel, copy the result back
static int eltwise_host_run(EltOp op, EltArgs g, uint64_t* result, const uint64_t* operand1,
                            const uint64_t* operand2) {
  if (int rc = check_elt(op, g)) return rc;
  const uint64_t n = g.n;
  int device = 0;
  HX_HIP(hipGetDevice(&device));
  const size_t bytes = (size_t)n * sizeof(u64);
  const bool has_b = operand2 != nullptr;
  // (one classification per distinct buffer)
  const RangeKind ra = classify_range(operand1, bytes);
  const RangeKind rb = !has_b                                        ? RangeKind{}
                       : (const void*)operand2 == (const void*)operand1 ? ra
                                                                     : classify_range(operand2, bytes);
  const RangeKind rr = (const void*)result == (const void*)operand1              ? ra
                       : has_b && (const void*)result == (const void*)operand2 ? rb
                                                                                : classify_range(result, bytes);
  void *r = rr.alias, *a = ra.alias, *b = rb.alias;
  const int kr = rr.kind, ka = ra.kind, kb = rb.kind;
  // mapped caller memory: the streaming kernel runs straight on it (see ntt_run_host)
  if (kr == 2 && ka == 2 && (!has_b || kb == 2)) {
    if (int rc = g_staging.ensure(device, 8)) return rc;
    g.result = (u64*)r;
    g.a = (const u64*)a;
    g.b = (const u64*)b;
    hipError_t e = eltwise_launch(op, g, g_staging.stream);
    if (e != hipSuccess) return hip_fail(e, "eltwise launch");
    if (int rc = g_staging.finish(g_staging.stream)) return rc;
    return HEXL_AMD_OK;
  }
  // small call, every buffer ordinary host memory: the mapped bounce buffer (ntt_run_host).
  // (The shim sends mixed argument sets here too -- a device operand with a host result: those
  // are not host-dereferenceable and take the staged copies below, whose direction is detected.)
  if (bytes <= host_bounce_max_bytes() && kr == 0 && ka == 0 && kb == 0 && rr.first == 0 &&
      ra.first == 0 && rb.first == 0) {
    if (int rc = g_staging.ensure(device, 8)) return rc;
    if (int rc = g_staging.ensure_bounce(bytes * (has_b ? 2 : 1))) return rc;
    u64* ha = (u64*)g_staging.bounce;
    u64* da_ = (u64*)g_staging.bounce_dev;
    memcpy(ha, operand1, bytes);
    if (has_b) memcpy(ha + n, operand2, bytes);
    g.result = da_;
    g.a = da_;
    g.b = has_b ? da_ + n : nullptr;
    hipError_t e = eltwise_launch(op, g, g_staging.stream);
    if (e != hipSuccess) return hip_fail(e, "eltwise launch");
    if (int rc = g_staging.finish(g_staging.stream)) return rc;
    memcpy(result, ha, bytes);
    return HEXL_AMD_OK;
  }
  if (int rc = g_staging.ensure(device, bytes * (has_b ? 2 : 1))) return rc;
  u64* da = (u64*)g_staging.buf;
  u64* db = has_b ? da + n : nullptr;
  hipStream_t st = g_staging.stream;
  if (int rc = copy_to_device(da, operand1, bytes, ra.first, st)) return rc;
  if (has_b)
    if (int rc = copy_to_device(db, operand2, bytes, rb.first, st)) return rc;
  g.result = da;
  g.a = da;
  g.b = db;
  hipError_t e = eltwise_launch(op, g, st);
  if (e != hipSuccess) return hip_fail(e, "eltwise launch");
  if (int rc = copy_from_device(result, da, bytes, rr.first, st)) return rc;
  HX_HIP(hipStreamSynchronize(st));
  return HEXL_AMD_OK;
}

// ------------------------------------------------------------------ DyadicMultiply

static int check_dyadic(const uint64_t* result, const uint64_t* a, const uint64_t* b, uint64_t n,
                        const uint64_t* moduli, uint64_t num_moduli) {
  if (!result) return fail(HEXL_AMD_ERR_INVALID_ARG, "result == nullptr");
  if (!a) return fail(HEXL_AMD_ERR_INVALID_ARG, "operand1 == nullptr");
  if (!b) return fail(HEXL_AMD_ERR_INVALID_ARG, "operand2 == nullptr");
  if (!moduli) return fail(HEXL_AMD_ERR_INVALID_ARG, "moduli == nullptr");
  if (n == 0) return fail(HEXL_AMD_ERR_INVALID_ARG, "n == 0");
  for (uint64_t i = 0; i < num_moduli; ++i)
    if (moduli[i] <= 1 || moduli[i] >= (1ull << 62))
      return fail(HEXL_AMD_ERR_INVALID_ARG, "moduli[%llu] must be in (1, 2^62)",
                  (unsigned long long)i);
  return HEXL_AMD_OK;
}

int hexl_amd_dyadic_multiply(uint64_t* result, const uint64_t* operand1, const uint64_t* operand2,
                             uint64_t n, const uint64_t* moduli, uint64_t num_moduli,
                             void* stream) {
  if (int rc = check_dyadic(result, operand1, operand2, n, moduli, num_moduli)) return rc;
  if (num_moduli == 0) return HEXL_AMD_OK;
  HX_ON_STREAM_DEVICE(stream);
  hipError_t e = dyadic_multiply_launch(result, operand1, operand2, n, moduli, num_moduli, 1,
                                        (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "dyadic multiply launch");
  return HEXL_AMD_OK;
}

int hexl_amd_dyadic_multiply_batch(uint64_t* result, const uint64_t* operand1,
                                   const uint64_t* operand2, uint64_t num_pairs, uint64_t n,
                                   const uint64_t* moduli, uint64_t num_moduli, void* stream) {
  if (int rc = check_dyadic(result, operand1, operand2, n, moduli, num_moduli)) return rc;
  if (num_moduli == 0 || num_pairs == 0) return HEXL_AMD_OK;
  if (num_pairs > 65535) return fail(HEXL_AMD_ERR_INVALID_ARG, "num_pairs must be <= 65535");
  HX_ON_STREAM_DEVICE(stream);
  hipError_t e = dyadic_multiply_launch(result, operand1, operand2, n, moduli, num_moduli,
                                        num_pairs, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "dyadic multiply launch");
  return HEXL_AMD_OK;
}

int hexl_amd_dyadic_multiply_host(uint64_t* result, const uint64_t* operand1,
                                  const uint64_t* operand2, uint64_t n, const uint64_t* moduli,
                                  uint64_t num_moduli) {
  if (int rc = check_dyadic(result, operand1, operand2, n, moduli, num_moduli)) return rc;
  if (num_moduli == 0) return HEXL_AMD_OK;
  int device = 0;
  HX_HIP(hipGetDevice(&device));
  const size_t poly = (size_t)n * num_moduli * sizeof(u64);
  const int k1 = pointer_kind(operand1, nullptr), k2 = pointer_kind(operand2, nullptr),
            kr = pointer_kind(result, nullptr);
  if (k1 == 0 && k2 == 0 && kr == 0 && 7 * poly <= host_bounce_max_bytes()) {
    // Small call, ordinary host memory on every side: the operands are copied into the thread's pinned,
    // device-mapped bounce buffer, the kernel runs on it over the link, the call's end is polled
    // (ntt_run_host; n = 4096 x 2 moduli: 76 -> ~30 us against four staged copies).  Layout as below.
    if (int rc = g_staging.ensure(device, 8)) return rc;  // (the stream)
    if (int rc = g_staging.ensure_bounce(7 * poly)) return rc;
    u64* hx_ = (u64*)g_staging.bounce;
    u64* dxb = (u64*)g_staging.bounce_dev;
    const size_t w = (size_t)n * num_moduli;
    memcpy(hx_, operand1, 2 * poly);
    memcpy(hx_ + 2 * w, operand2, 2 * poly);
    // (coefficients the reference leaves untouched -- n > 512 not a multiple of 512 -- keep what the
    // caller's result buffer holds)
    if (n > 512 && (n & 511) != 0) memcpy(hx_ + 4 * w, result, 3 * poly);
    hipError_t e = dyadic_multiply_launch(dxb + 4 * w, dxb, dxb + 2 * w, n, moduli, num_moduli, 1,
                                          g_staging.stream);
    if (e != hipSuccess) return hip_fail(e, "dyadic multiply launch");
    if (int rc = g_staging.finish(g_staging.stream)) return rc;
    memcpy(result, hx_ + 4 * w, 3 * poly);
    return HEXL_AMD_OK;
  }
  // layout of the staging buffer: x (2 polys) | y (2 polys) | result (3 polys)
  if (int rc = g_staging.ensure(device, 7 * poly)) return rc;
  u64* dx = (u64*)g_staging.buf;
  u64* dy = dx + 2 * n * num_moduli;
  u64* dr = dy + 2 * n * num_moduli;
  hipStream_t st = g_staging.stream;
  if (int rc = copy_to_device(dx, operand1, 2 * poly, k1, st)) return rc;
  if (int rc = copy_to_device(dy, operand2, 2 * poly, k2, st)) return rc;
  // coefficients the reference leaves untouched (n > 512 not a multiple of 512) keep
  // whatever the caller's result buffer holds
  if (int rc = copy_to_device(dr, result, 3 * poly, kr, st)) return rc;
  hipError_t e = dyadic_multiply_launch(dr, dx, dy, n, moduli, num_moduli, 1, st);
  if (e != hipSuccess) return hip_fail(e, "dyadic multiply launch");
  if (int rc = copy_from_device(result, dr, 3 * poly, kr, st)) return rc;
  HX_HIP(hipStreamSynchronize(st));
  return HEXL_AMD_OK;
}

// ------------------------------------------------------------------ KeySwitch

namespace {

// Plans by (n, q, device), the counterpart of the reference's GetNTT cache
// (hexl/include/hexl/experimental/seal/ntt-cache.hpp:27-53).  Entries live for
// the life of the process.
const hexl_amd_ntt* cached_plan(u64 n, u64 q, int device) {
  // a per-thread front cache keeps the steady state off the global mutex
  thread_local std::map<std::tuple<u64, u64, int>, const hexl_amd_ntt*> local;
  const auto key = std::make_tuple(n, q, device);
  auto hit = local.find(key);
  if (hit != local.end()) return hit->second;
  static std::mutex mu;
  static std::map<std::tuple<u64, u64, int>, hexl_amd_ntt*> cache;
  const hexl_amd_ntt* plan = nullptr;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) {
      plan = it->second;
    } else {
      hexl_amd_ntt* p = nullptr;
      if (hexl_amd_ntt_create(&p, n, q, 0, device) != HEXL_AMD_OK) return nullptr;
      cache.emplace(key, p);
      plan = p;
    }
  }
  local.emplace(key, plan);
  return plan;
}

u64 floor_2_64_over(u64 q) { return (u64)((((unsigned __int128)1) << 64) / q); }

}  // namespace

static int check_key_switch(const uint64_t* result, const uint64_t* t_target, uint64_t n,
                            uint64_t D, uint64_t K, uint64_t R, uint64_t C,
                            const uint64_t* moduli, const uint64_t* const* keys,
                            const uint64_t* msf) {
  if (!result) return fail(HEXL_AMD_ERR_INVALID_ARG, "result == nullptr");
  if (!t_target) return fail(HEXL_AMD_ERR_INVALID_ARG, "t_target_iter_ptr == nullptr");
  if (!moduli) return fail(HEXL_AMD_ERR_INVALID_ARG, "moduli == nullptr");
  if (!keys) return fail(HEXL_AMD_ERR_INVALID_ARG, "k_switch_keys == nullptr");
  if (!msf) return fail(HEXL_AMD_ERR_INVALID_ARG, "modswitch_factors == nullptr");
  if (n == 0 || D == 0 || C == 0)
    return fail(HEXL_AMD_ERR_INVALID_ARG, "n, decomp_modulus_size, key_component_count must be > 0");
  if (D > (uint64_t)kKsMaxDecomp)
    return fail(HEXL_AMD_ERR_INVALID_ARG, "decomp_modulus_size > %d is not supported", kKsMaxDecomp);
  if (C > 65535) return fail(HEXL_AMD_ERR_INVALID_ARG, "key_component_count too large");
  // the algorithm indexes moduli[0..D-1] and moduli[K-1], and RNS index D uses key K-1
  if (K < D + 1 || R != D + 1)
    return fail(HEXL_AMD_ERR_INVALID_ARG,
                "need key_modulus_size > decomp_modulus_size and rns_modulus_size == "
                "decomp_modulus_size + 1");
  for (uint64_t j = 0; j < D; ++j)
    if (!keys[j]) return fail(HEXL_AMD_ERR_INVALID_ARG, "k_switch_keys[%llu] == nullptr",
                              (unsigned long long)j);
  // (the primality test of hexl_amd_ntt_check_arguments -- twelve Miller-Rabin rounds -- is 2-3 us per
  // modulus: 21 of the 55 us a call with eight moduli spent on the host before round 6 remembered, per
  // thread, the (degree, modulus) pairs that passed)
  thread_local std::vector<std::pair<uint64_t, uint64_t>> passed;
  auto known = [&](uint64_t q) {
    for (const auto& e : passed)
      if (e.first == n && e.second == q) return true;
    return false;
  };
  for (uint64_t i = 0; i < K; ++i)
    if (i < D || i == K - 1) {
      if (moduli[i] < (1ull << 61) && known(moduli[i])) continue;
      if (hexl_amd_ntt_check_arguments(n, moduli[i]) && moduli[i] < (1ull << 61)) {
        if (passed.size() >= 256) passed.clear();
        passed.emplace_back(n, moduli[i]);
        continue;
      }
      return fail(HEXL_AMD_ERR_INVALID_ARG,
                  "moduli[%llu] is not an NTT-friendly prime below 2^61 for degree %llu",
                  (unsigned long long)i, (unsigned long long)n);
    }
  // the factors go through EltwiseFMAMod(input_mod_factor = 8): key-switch-internal.cpp:190
  for (uint64_t i = 0; i < D; ++i)
    if (msf[i] >= 8 * moduli[i])
      return fail(HEXL_AMD_ERR_INVALID_ARG, "modswitch_factors[%llu] must be < 8 * moduli[%llu]",
                  (unsigned long long)i, (unsigned long long)i);
  return HEXL_AMD_OK;
}

// Transforms of `polys` consecutive polynomials whose moduli follow `map` (plan ids index
// `plans`): one multi-plan launch sequence where the shapes allow it, otherwise one launch
// per polynomial (small degrees -- the reference's own small test vectors).
static int ntt_mapped(bool forward, const std::vector<const hexl_amd_ntt*>& plans,
                      const MultiMap& map, u64 polys, u64* result, const u64* operand, u64 n,
                      u64 out_mf, hipStream_t st) {
  std::vector<const NttTables*> tabs(plans.size());
  for (size_t k = 0; k < plans.size(); ++k) tabs[k] = &plans[k]->t;
  hipError_t e = ntt_multi_launch(forward, tabs.data(), (u32)tabs.size(), map, polys, result,
                                  operand, out_mf, st);
  if (e == hipSuccess) return HEXL_AMD_OK;
  if (e != hipErrorNotSupported) return hip_fail(e, "KeySwitch multi-plan NTT");
  for (u64 b = 0; b < polys; ++b) {
    const NttTables& t = plans[map.plan_tab[(b / map.inner) % map.period]]->t;
    e = forward ? ntt_forward_launch(t, result + b * n, operand + b * n, 1, out_mf, st)
                : ntt_inverse_launch(t, result + b * n, operand + b * n, 1, out_mf, st);
    if (e != hipSuccess) return hip_fail(e, "KeySwitch NTT");
  }
  return HEXL_AMD_OK;
}

// Device buffers throughout; T targets that share moduli and keys.  `keys`: host array of
// D device pointers.  At most eleven launches whatever T, D and C are (degree >= 4096;
// seven where the transforms are one kernel, N = 8192 and N = 16384 with enough polynomials):
// inverse NTT of the targets (1-2), lazy forward NTT of all product operands, reading the
// coefficient-form targets through a source map and reducing them on load (1-2),
// multiply-accumulate (1), inverse NTT of the last components (1-2), rounding (1), forward
// NTT of the corrections (1-2), finish (1).
static int key_switch_device(u64* result, const u64* t_target_iter, u64 T, u64 n, u64 D, u64 K,
                             u64 R, u64 C, const u64* moduli, const u64* const* keys,
                             const u64* msf, hipStream_t st) {
  int device = 0;
  HX_HIP(hipGetDevice(&device));
  // plan ids: 0..D-1 the decomposition moduli, D the special modulus moduli[K-1]
  std::vector<const hexl_amd_ntt*> plan(D + 1, nullptr);
  for (u64 i = 0; i <= D; ++i) {
    plan[i] = cached_plan(n, moduli[i < D ? i : K - 1], device);
    if (!plan[i]) return HEXL_AMD_ERR_HIP;  // message set by hexl_amd_ntt_create
  }
  // workspace: t_target (T D) | ntt_buf (T D^2) | prod (R T C) | tbuf (T C D) polynomials.
  // Keyed by the caller's stream: calls on one stream are serialised and share it, calls
  // on different streams may overlap on the device and get separate buffers.
  // The sequence lock keeps another host thread's KeySwitch on the same stream from
  // interleaving its launches with these (or regrowing the buffer under them).
  const size_t words = (size_t)n * T * (D + D * D + R * C + C * D);
  void* ws = nullptr;
  StreamSequenceLock sequence(st);
  HX_HIP(stream_workspace(kWsKeySwitch, st, words * sizeof(u64), &ws));
  u64* t_target = (u64*)ws;
  u64* ntt_buf = t_target + T * D * n;
  u64* prod = ntt_buf + T * D * D * n;
  u64* tbuf = prod + R * T * C * n;
  const KsDims dims{n, (u32)D, (u32)T, (u32)C, (u32)K};
  // the launch sequence (no allocation, no synchronisation: capturable)
  auto enqueue = [&](hipStream_t s) -> int {
    hipError_t e;

    // key-switch-internal.cpp:38-56: coefficient form of the targets per decomposition modulus
    {
      MultiMap map{};
      map.inner = 1;
      map.period = (u32)D;
      for (u64 j = 0; j < D; ++j) map.plan_tab[j] = (uint8_t)j;
      if (int rc = ntt_mapped(false, plan, map, T * D, t_target, t_target_iter, n, 1, s)) return rc;
    }

    // :61-131 per RNS index i: operands to the key modulus, lazy forward NTT, MAC, reduce
    KsGatherAll g{};
    KsMacAll m{};
    for (u64 j = 0; j < D; ++j) m.keys[j] = keys[j];
    for (u64 i = 0; i < R; ++i) {
      const u64 key_index = (i == D) ? K - 1 : i;
      const u64 q = moduli[key_index];
      g.q[i] = m.q[i] = q;
      g.barrett[i] = m.barrett[i] = floor_2_64_over(q);
      for (u64 j = 0; j < D; ++j)
        if (j != i && moduli[j] > q) g.reduce_mask[i] |= 1u << j;
      m.key_index[i] = (u32)key_index;
      m.two64_mod_q[i] = (u64)((((unsigned __int128)1) << 64) % q);
      const u32 ceil_log = 64 - __builtin_clzll(q);
      m.shift[i] = ceil_log - 2;
      m.mu[i] = (u64)((((unsigned __int128)(1ull << (ceil_log + 62 - 64))) << 64) / q);
      // sums below 2^(bits(q) + 61) take one generalised Barrett step (ks_mac_kernel)
      m.hi_limit[i] = ceil_log >= 4 && g_ks_mac_onestep.load() != 0 ? 1ull << (ceil_log - 3) : 0;
    }
    {
      // The D^2 product operands of a target, ordered by RNS index i: target polynomial j
      // (coefficient form) reduced to q_i where moduli[j] is larger, for every j != i (i < D)
      // or every j (i == D).  The first pass of their forward transform reads them straight
      // from t_target through the source map (MultiMap); only where the multi-plan launch
      // does not apply (small degrees) are they gathered into ntt_buf first.
      MultiMap map{};
      map.inner = 1;
      map.period = (u32)(D * D);
      map.src_stride = (u32)D;
      for (u64 s = 0; s < D * D; ++s) {
        const u64 i = s < D * (D - 1) ? s / (D - 1) : D;
        const u64 r = s - i * (D - 1);
        const u64 j = i < D ? (r < i ? r : r + 1) : s - D * (D - 1);
        map.plan_tab[s] = (uint8_t)i;
        map.src_tab[s] = (uint8_t)(j | (((g.reduce_mask[i] >> j) & 1) ? 0x80 : 0));
      }
      std::vector<const NttTables*> tabs(plan.size());
      for (size_t k = 0; k < plan.size(); ++k) tabs[k] = &plan[k]->t;
      e = ntt_multi_launch(true, tabs.data(), (u32)tabs.size(), map, T * D * D, ntt_buf, t_target,
                           4, s);
      if (e == hipErrorNotSupported) {
        e = ks_gather_launch(ntt_buf, t_target, dims, g, s);
        if (e != hipSuccess) return hip_fail(e, "KeySwitch gather");
        map.src_stride = 0;
        if (int rc = ntt_mapped(true, plan, map, T * D * D, ntt_buf, ntt_buf, n, 4, s)) return rc;
      } else if (e != hipSuccess) {
        return hip_fail(e, "KeySwitch multi-plan NTT");
      }
    }
    e = ks_mac_launch(prod, t_target_iter, ntt_buf, dims, m, s);
    if (e != hipSuccess) return hip_fail(e, "KeySwitch multiply-accumulate");

    // :134-197 modulus switching from the special prime, all (target, key component) at once
    const u64 qk = moduli[K - 1];
    KsRound rd{};
    KsFinish fin{};
    rd.qk = qk;
    rd.barrett_k = floor_2_64_over(qk);
    rd.qk_half = qk >> 1;
    for (u64 i = 0; i < D; ++i) {
      const u64 qi = moduli[i];
      const u64 bf = floor_2_64_over(qi);
      u64 h = rd.qk_half - (u64)(((unsigned __int128)rd.qk_half * bf) >> 64) * qi;  // BarrettReduce64
      if (h >= qi) h -= qi;
      rd.mod[i] = KsRoundMod{qi, bf, qi - h, qk > qi ? 1u : 0u};
      u64 s = msf[i];  // FMAMod reduces its scalar from [0, 8q) (eltwise-fma-mod.cpp:60-64)
      if (s >= 4 * qi) s -= 4 * qi;
      if (s >= 2 * qi) s -= 2 * qi;
      if (s >= qi) s -= qi;
      fin.mod[i] = KsFinishMod{qi, s, (u64)((((unsigned __int128)s) << 64) / qi)};
    }
    u64* t_last = prod + D * T * C * n;  // prod[D][.][.]: T C contiguous polynomials
    e = ntt_inverse_launch(plan[D]->t, t_last, t_last, T * C, 2, s);
    if (e != hipSuccess) return hip_fail(e, "KeySwitch inverse NTT (last)");
    if (g_ks_fuse.load() != 0) {
      // Round 6: rounding and finish ride on the forward transform of the corrections -- its first
      // pass reads the last component through a source map and rounds it on load (kRoundFirst),
      // its last pass folds its output into the result (KsEpilogue): two launches and the write +
      // read of two T C D-polynomial buffers less.  Degrees the multi-plan kernels do not serve
      // (below 4096) take the stage-by-stage sequence below.
      MultiMap map{};
      map.inner = 1;
      map.period = (u32)D;
      map.src_stride = 1;
      map.rnd_qk = rd.qk;
      map.rnd_barrett = rd.barrett_k;
      map.rnd_half = rd.qk_half;
      KsEpilogue ep{};
      ep.result = result;
      ep.prod = prod;
      ep.decomp = (u32)D;
      ep.tc = (u32)(T * C);
      for (u64 i = 0; i < D; ++i) {
        map.plan_tab[i] = (uint8_t)i;
        map.src_tab[i] = (uint8_t)(rd.mod[i].reduce ? 0x80 : 0);
        ep.s[i] = fin.mod[i].s;
        ep.sp[i] = fin.mod[i].sp;
      }
      std::vector<const NttTables*> tabs(plan.size());
      for (size_t k = 0; k < plan.size(); ++k) tabs[k] = &plan[k]->t;
      e = ntt_multi_launch(true, tabs.data(), (u32)tabs.size(), map, T * C * D, tbuf, t_last, 4, s, &ep);
      if (e == hipSuccess) return HEXL_AMD_OK;
      if (e != hipErrorNotSupported) return hip_fail(e, "KeySwitch fused tail");
    }
    e = ks_round_launch(tbuf, prod, dims, rd, s);
    if (e != hipSuccess) return hip_fail(e, "KeySwitch rounding");
    {
      MultiMap map{};
      map.inner = 1;
      map.period = (u32)D;
      for (u64 i = 0; i < D; ++i) map.plan_tab[i] = (uint8_t)i;
      if (int rc = ntt_mapped(true, plan, map, T * C * D, tbuf, tbuf, n, 4, s)) return rc;
    }
    e = ks_finish_launch(result, prod, tbuf, dims, fin, s);
    if (e != hipSuccess) return hip_fail(e, "KeySwitch finish");
    return HEXL_AMD_OK;
  };

  // One ciphertext per call -- what a caller of the reference's KeySwitch does
  // (key-switch-internal.cpp:25-201 is one target) -- is launch-bound: ~6 us of transform work
  // behind eleven dependent launches.  A sequence that comes back with the very same buffers,
  // keys and moduli on the same stream is captured once into a HIP graph and replayed from then
  // on (workspace.h; hexl_amd_key_switch_host always qualifies: it stages into the thread's own
  // buffers).  First sight of a buffer set runs eagerly, so a caller that walks over different
  // ciphertexts never pays a capture.
  bool replayable = T <= (u64)kKsGraphMaxTargets && g_ks_graph.load() != 0 && g_profile == nullptr &&
                    st != nullptr && st != hipStreamLegacy && st != hipStreamPerThread;
  if (replayable) {
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &capturing) != hipSuccess) (void)hipGetLastError();
    replayable = capturing == hipStreamCaptureStatusNone;  // (inside a caller's own capture: eager)
  }
  if (!replayable) {
    g_ks_eager.fetch_add(1, std::memory_order_relaxed);
    return enqueue(st);
  }
  std::vector<uint64_t> key;
  key.reserve(11 + K + 2 * D);
  for (u64 v : {(u64)(uintptr_t)result, (u64)(uintptr_t)t_target_iter, (u64)(uintptr_t)ws, T, n, D, K, R, C,
                g_tuning_epoch.load(std::memory_order_relaxed)})
    key.push_back(v);
  for (u64 i = 0; i < K; ++i) key.push_back(moduli[i]);
  for (u64 i = 0; i < D; ++i) key.push_back(msf[i]);
  for (u64 j = 0; j < D; ++j) key.push_back((u64)(uintptr_t)keys[j]);
  hipGraphExec_t exec = nullptr;
  switch (lookup_sequence_graph(st, key, &exec)) {
    case kGraphReplay:
      if (hipGraphLaunch(exec, st) == hipSuccess) {
        g_ks_graph_replays.fetch_add(1, std::memory_order_relaxed);
        return HEXL_AMD_OK;
      }
      (void)hipGetLastError();
      poison_sequence_graph(st, key);
      return enqueue(st);
    case kGraphCapture: {
      // The sequence is captured on a stream of the library's own, never on the caller's: a capture
      // applies to the whole stream, and work another thread of the caller enqueued on `st` during
      // the capture window (a transform, a copy, a kernel of its own) would be recorded into the
      // graph -- not run now, run again on every replay -- while a synchronisation on `st` would
      // fail (round-5 advice).  The scratch and every pointer are those of `st`; the graph is
      // launched on `st`.
      hipStream_t cap = nullptr;
      if (hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) != hipSuccess ||
          hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        if (cap) (void)hipStreamDestroy(cap);
        poison_sequence_graph(st, key);
        return enqueue(st);
      }
      const int rc = enqueue(cap);
      hipGraph_t graph = nullptr;
      hipError_t ec = hipStreamEndCapture(cap, &graph);
      (void)hipStreamDestroy(cap);
      if (rc == HEXL_AMD_OK && ec == hipSuccess && graph)
        ec = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      else if (ec == hipSuccess)
        ec = hipErrorUnknown;
      if (graph) (void)hipGraphDestroy(graph);
      if (ec == hipSuccess) ec = hipGraphLaunch(exec, st);
      if (ec != hipSuccess) {  // nothing was enqueued: run the sequence as it is (and report its error)
        (void)hipGetLastError();
        if (exec) (void)hipGraphExecDestroy(exec);
        poison_sequence_graph(st, key);
        return enqueue(st);
      }
      store_sequence_graph(st, key, exec);
      g_ks_graph_captures.fetch_add(1, std::memory_order_relaxed);
      return HEXL_AMD_OK;
    }
    default:
      g_ks_eager.fetch_add(1, std::memory_order_relaxed);
      return enqueue(st);
  }
}

int hexl_amd_key_switch(uint64_t* result, const uint64_t* t_target_iter_ptr, uint64_t n,
                        uint64_t decomp_modulus_size, uint64_t key_modulus_size,
                        uint64_t rns_modulus_size, uint64_t key_component_count,
                        const uint64_t* moduli, const uint64_t* const* k_switch_keys,
                        const uint64_t* modswitch_factors, void* stream) {
  if (int rc = check_key_switch(result, t_target_iter_ptr, n, decomp_modulus_size,
                                key_modulus_size, rns_modulus_size, key_component_count, moduli,
                                k_switch_keys, modswitch_factors))
    return rc;
  HX_ON_STREAM_DEVICE(stream);
  return key_switch_device(result, t_target_iter_ptr, 1, n, decomp_modulus_size, key_modulus_size,
                           rns_modulus_size, key_component_count, moduli, k_switch_keys,
                           modswitch_factors, (hipStream_t)stream);
}

int hexl_amd_key_switch_batch(uint64_t* result, const uint64_t* t_target_iter_ptr,
                              uint64_t num_targets, uint64_t n, uint64_t decomp_modulus_size,
                              uint64_t key_modulus_size, uint64_t rns_modulus_size,
                              uint64_t key_component_count, const uint64_t* moduli,
                              const uint64_t* const* k_switch_keys,
                              const uint64_t* modswitch_factors, void* stream) {
  if (int rc = check_key_switch(result, t_target_iter_ptr, n, decomp_modulus_size,
                                key_modulus_size, rns_modulus_size, key_component_count, moduli,
                                k_switch_keys, modswitch_factors))
    return rc;
  if (num_targets == 0) return HEXL_AMD_OK;
  if (num_targets * key_component_count > 65535)
    return fail(HEXL_AMD_ERR_INVALID_ARG, "num_targets * key_component_count must be <= 65535");
  HX_ON_STREAM_DEVICE(stream);
  return key_switch_device(result, t_target_iter_ptr, num_targets, n, decomp_modulus_size,
                           key_modulus_size, rns_modulus_size, key_component_count, moduli,
                           k_switch_keys, modswitch_factors, (hipStream_t)stream);
}

int hexl_amd_key_switch_host(uint64_t* result, const uint64_t* t_target_iter_ptr, uint64_t n,
                             uint64_t D, uint64_t K, uint64_t R, uint64_t C,
                             const uint64_t* moduli, const uint64_t* const* k_switch_keys,
                             const uint64_t* modswitch_factors) {
  if (int rc = check_key_switch(result, t_target_iter_ptr, n, D, K, R, C, moduli, k_switch_keys,
                                modswitch_factors))
    return rc;
  int device = 0;
  HX_HIP(hipGetDevice(&device));
  // staging: result (C D n) | target (D n) | D key blocks of C K n words.  The keys are the bulk
  // of a call's bytes (n = 16384, 7 moduli: 14.7 MB against 0.9 MB of result) and long-lived on
  // the caller's side: a key block that already IS device memory (uploaded once with
  // hexl_amd_device_alloc + hexl_amd_copy, intel::hexl::DeviceMalloc + Copy) is used where it
  // lies -- host result / target with device-resident keys is the cheap way to call this from
  // an unmodified per-ciphertext loop.
  const size_t res_words = (size_t)C * D * n, key_words = (size_t)C * K * n;
  std::vector<const u64*> kp(D);
  size_t host_keys = 0;
  for (u64 j = 0; j < D; ++j) {
    void* alias = nullptr;
    if (pointer_kind(k_switch_keys[j], &alias) == 1)
      kp[j] = (const u64*)alias;
    else
      ++host_keys;
  }
  if (int rc = g_staging.ensure(device, (res_words + D * n + host_keys * key_words) * sizeof(u64)))
    return rc;
  u64* d_res = (u64*)g_staging.buf;
  u64* d_tgt = d_res + res_words;
  u64* d_keys = d_tgt + D * n;
  hipStream_t st = g_staging.stream;
  const int k_res = pointer_kind(result, nullptr), k_tgt = pointer_kind(t_target_iter_ptr, nullptr);
  if (int rc = copy_to_device(d_res, result, res_words * sizeof(u64), k_res, st)) return rc;
  if (int rc = copy_to_device(d_tgt, t_target_iter_ptr, D * n * sizeof(u64), k_tgt, st)) return rc;
  for (u64 j = 0, slot = 0; j < D; ++j) {
    if (kp[j]) continue;
    if (int rc = copy_to_device(d_keys + slot * key_words, k_switch_keys[j], key_words * sizeof(u64),
                                pointer_kind(k_switch_keys[j], nullptr), st))
      return rc;
    kp[j] = d_keys + slot * key_words;
    ++slot;
  }
  if (int rc = key_switch_device(d_res, d_tgt, 1, n, D, K, R, C, moduli, kp.data(),
                                 modswitch_factors, st))
    return rc;
  if (int rc = copy_from_device(result, d_res, res_words * sizeof(u64), k_res, st)) return rc;
  HX_HIP(hipStreamSynchronize(st));
  return HEXL_AMD_OK;
}

// ------------------------------------------------------------------ number theory

uint64_t hexl_amd_multiply_factor(uint64_t operand, uint64_t bit_shift, uint64_t modulus) {
  return nt::multiply_factor(operand, bit_shift, modulus);
}
uint64_t hexl_amd_inverse_mod(uint64_t x, uint64_t modulus) { return nt::inverse_mod(x, modulus); }
uint64_t hexl_amd_multiply_mod(uint64_t x, uint64_t y, uint64_t modulus) {
  return nt::multiply_mod(x, y, modulus);
}
uint64_t hexl_amd_pow_mod(uint64_t base, uint64_t exp, uint64_t modulus) {
  return nt::pow_mod(base, exp, modulus);
}
int hexl_amd_is_primitive_root(uint64_t root, uint64_t degree, uint64_t modulus) {
  return nt::is_primitive_root(root, degree, modulus) ? 1 : 0;
}
uint64_t hexl_amd_generate_primitive_root(uint64_t degree, uint64_t modulus) {
  return nt::generate_primitive_root(degree, modulus);
}
uint64_t hexl_amd_minimal_primitive_root(uint64_t degree, uint64_t modulus) {
  return nt::minimal_primitive_root(degree, modulus);
}
uint64_t hexl_amd_reverse_bits(uint64_t x, uint64_t bit_width) {
  return nt::reverse_bits(x, bit_width);
}
int hexl_amd_is_prime(uint64_t n) { return nt::is_prime(n) ? 1 : 0; }
size_t hexl_amd_generate_primes(uint64_t* out, size_t num_primes, size_t bit_size,
                                int prefer_small_primes, size_t ntt_size) {
  return nt::generate_primes(out, num_primes, bit_size, prefer_small_primes != 0, ntt_size);
}

// ------------------------------------------------------------------ profiling

namespace {
struct ProfileState {
  std::vector<ProfileRecord> records;
  ProfileSink sink{nullptr, 0, 0};
  std::vector<float> ms;
};
thread_local ProfileState g_prof_state;
}  // namespace

int hexl_amd_profile_start(int max_records) {
  ProfileState& ps = g_prof_state;
  if (max_records <= 0) return fail(HEXL_AMD_ERR_INVALID_ARG, "max_records <= 0");
  for (size_t i = ps.records.size(); i < (size_t)max_records; ++i) {
    ProfileRecord r{nullptr, nullptr, nullptr};
    HX_HIP(hipEventCreate(&r.start));
    HX_HIP(hipEventCreate(&r.stop));
    ps.records.push_back(r);
  }
  ps.sink.records = ps.records.data();
  ps.sink.capacity = max_records;
  ps.sink.count = 0;
  g_profile = &ps.sink;
  return HEXL_AMD_OK;
}

int hexl_amd_profile_stop(int* num_records) {
  ProfileState& ps = g_prof_state;
  g_profile = nullptr;
  const int n = ps.sink.count;
  ps.ms.assign(n, 0.f);
  for (int i = 0; i < n; ++i) {
    HX_HIP(hipEventSynchronize(ps.records[i].stop));
    HX_HIP(hipEventElapsedTime(&ps.ms[i], ps.records[i].start, ps.records[i].stop));
  }
  if (num_records) *num_records = n;
  return HEXL_AMD_OK;
}

int hexl_amd_profile_get(int i, const char** name, float* ms) {
  ProfileState& ps = g_prof_state;
  if (i < 0 || i >= (int)ps.ms.size()) return fail(HEXL_AMD_ERR_INVALID_ARG, "bad record index");
  if (name) *name = ps.records[i].name;
  if (ms) *ms = ps.ms[i];
  return HEXL_AMD_OK;
}

int hexl_amd_release_stream_workspaces(void* stream) {
  // (scratch is keyed by (device, stream): look on the device that owns the stream, like every
  // other stream-taking entry point)
  HX_ON_STREAM_DEVICE(stream);
  release_stream_workspaces((hipStream_t)stream);
  return HEXL_AMD_OK;
}
int hexl_amd_release_workspaces(void) {
  const int busy = release_workspaces();
  if (busy)
    return fail(HEXL_AMD_ERR_INVALID_ARG,
                "%d scratch buffer(s) not released: a composite call was enqueueing on their "
                "stream", busy);
  return HEXL_AMD_OK;
}

int hexl_amd_get_counter(const char* key, uint64_t* value) {
  if (!key || !value) return fail(HEXL_AMD_ERR_INVALID_ARG, "key == nullptr or value == nullptr");
  if (strcmp(key, "ks_graph_captures") == 0)
    *value = g_ks_graph_captures.load();
  else if (strcmp(key, "ks_graph_replays") == 0)
    *value = g_ks_graph_replays.load();
  else if (strcmp(key, "ks_eager") == 0)
    *value = g_ks_eager.load();
  else if (strcmp(key, "host_polls") == 0)
    *value = g_host_polls.load();
  else if (strcmp(key, "host_poll_timeouts") == 0)
    *value = g_host_poll_timeouts.load();
  else
    return fail(HEXL_AMD_ERR_INVALID_ARG, "unknown counter: %s", key);
  return HEXL_AMD_OK;
}

int hexl_amd_set_tuning(const char* key, uint64_t value) {
  if (!key) return fail(HEXL_AMD_ERR_INVALID_ARG, "key == nullptr");
  g_tuning_epoch.fetch_add(1, std::memory_order_relaxed);
  if (set_host_tuning(key, value)) return HEXL_AMD_OK;
  if (set_tuning(key, value) != 0)
    return fail(HEXL_AMD_ERR_INVALID_ARG, "unknown tuning key or value out of range: %s", key);
  return HEXL_AMD_OK;
}

int hexl_amd_fill_splitmix(uint64_t* data, uint64_t n, uint64_t batch, uint64_t seed0,
                           uint64_t bound, void* stream) {
  if (!data) return fail(HEXL_AMD_ERR_INVALID_ARG, "data == nullptr");
  HX_ON_STREAM_DEVICE(stream);
  hipError_t e = fill_splitmix_launch(data, n, batch, seed0, bound, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "fill launch");
  return HEXL_AMD_OK;
}

}  // extern "C"
