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

namespace hexl_amd {

enum WorkspacePurpose : int {
  kWsFusedNtt = 0,   // scheduler state of fused_pass (ntt_kernels.hip)
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
