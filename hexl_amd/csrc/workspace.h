// workspace.h -- stream-keyed device scratch buffers.
//
// Work enqueued on one HIP stream executes in order, so successive operations on a
// stream may reuse one scratch buffer; operations on DIFFERENT streams may overlap
// on the device and must not share scratch.  Buffers are therefore keyed by
// (device, stream, purpose), grow on demand and live until released (release_workspaces,
// release_stream_workspaces) or the process ends.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace hexl_amd {

enum WorkspacePurpose : int {
  kWsFusedNtt = 0,   // (unused since round 4: was the scheduler state of the archived fused plan, experiments/)
  kWsKeySwitch = 1,  // t_target | ntt_buf | t_poly_prod of KeySwitch (capi.cpp)
};

// Device buffer of at least `bytes` bytes for `purpose` on (current device, stream).
// Growing synchronises the device (hipFree) -- it happens once per size class.
hipError_t stream_workspace(WorkspacePurpose purpose, hipStream_t stream, size_t bytes,
                            void** out);

// Held while a multi-launch sequence that uses a stream workspace is being ENQUEUED
// (KeySwitch: up to eleven launches over one scratch buffer): two host threads that issue such
// sequences on the same stream would otherwise interleave their launches -- stream order
// then serialises the kernels, but of two half-finished sequences over one buffer -- or
// free the buffer the other is still enqueuing against.  Keyed like the buffers by
// (current device, stream); enqueueing is microseconds, the kernels run outside the lock.
// Recursive: a composite that holds the lock while it enqueues (KeySwitch) may call a transform
// that takes it again for its own scratch (the one-launch fused transform).
class StreamSequenceLock {
 public:
  explicit StreamSequenceLock(hipStream_t stream);
  ~StreamSequenceLock();
  StreamSequenceLock(const StreamSequenceLock&) = delete;
  StreamSequenceLock& operator=(const StreamSequenceLock&) = delete;

 private:
  struct Held;
  Held* held_;
};

// Replay cache for launch sequences enqueued against a stream's scratch (KeySwitch, one
// ciphertext per call: at most eleven dependent launches of a few microseconds each, where the
// host-side launch overhead is most of the call).  A sequence is identified by EVERYTHING its
// launches depend on -- every pointer, size and per-modulus constant, the scratch address --
// as a vector of words compared exactly (no digest, no collisions).  Protocol, under the
// stream's StreamSequenceLock:
//   lookup_sequence_graph(...)  -> kGraphReplay: *exec is an instantiated graph of the sequence
//                                  (hipGraphLaunch it);
//                               -> kGraphCapture: the key was seen before: capture the sequence
//                                  now and hand the instantiated graph to store_sequence_graph;
//                               -> kGraphEager: first sight of the key (a caller that walks over
//                                  different buffers never pays a capture): enqueue eagerly.
// At most kMaxSequenceGraphs keys per (device, stream), least recently used evicted; all graphs of
// a stream go when its scratch is regrown or released.
enum SequenceGraphAction : int { kGraphEager = 0, kGraphCapture = 1, kGraphReplay = 2 };
constexpr int kMaxSequenceGraphs = 8;
SequenceGraphAction lookup_sequence_graph(hipStream_t stream, const std::vector<uint64_t>& key,
                                          hipGraphExec_t* exec);
void store_sequence_graph(hipStream_t stream, const std::vector<uint64_t>& key, hipGraphExec_t exec);
// the key could not be captured (or replay failed): stays eager for the life of the entry
void poison_sequence_graph(hipStream_t stream, const std::vector<uint64_t>& key);

// Frees every cached buffer of the current process (all devices), each under its stream's
// sequence lock.  Returns the number of buffers left alone because another thread was
// enqueueing against them at that moment (0: everything freed).  C-ABI:
// hexl_amd_release_workspaces.
int release_workspaces();

// Frees the buffers (and the sequence lock) keyed by `stream` on the current device; for
// owners about to destroy the stream (the per-thread staging streams of the host-pointer
// entry points; callers that create and destroy streams: hexl_amd_release_stream_workspaces).
void release_stream_workspaces(hipStream_t stream);

}  // namespace hexl_amd
