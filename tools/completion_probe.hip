// completion_probe.hip -- what one synchronous call on mapped host memory costs under four ways of learning
// that the stream's work is done (developer probe for the *_host paths, capi.cpp):
//   sync      hipStreamSynchronize
//   event     hipEventRecord + hipEventSynchronize (blocking-sync and spinning flavours)
//   flagk     a one-thread kernel behind the work stores a sequence number into mapped host memory; the host spins
//   writev    hipStreamWriteValue32 behind the work; the host spins
// The work is the library's own forward transform of one polynomial in place on memory from
// hexl_amd_host_alloc (what the bounce path runs).
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/completion_probe.hip -Lhexl_amd/lib -lhexl_amd
//       -Wl,-rpath,$PWD/hexl_amd/lib -o tools/completion_probe
#include <hip/hip_runtime.h>
#include <time.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hexl_amd.h"

static double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
#define OK(call)                                                             \
  do {                                                                       \
    if ((call) != 0) {                                                       \
      std::fprintf(stderr, "%s failed: %s\n", #call, hexl_amd_last_error()); \
      std::exit(2);                                                          \
    }                                                                        \
  } while (0)
#define HIP(call)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));           \
      std::exit(2);                                                                    \
    }                                                                                  \
  } while (0)

__global__ void flag_kernel(volatile unsigned* flag, unsigned seq) {
  __atomic_store_n((unsigned*)flag, seq, __ATOMIC_RELEASE);  // system scope by default on fine-grained memory
  __threadfence_system();
}

// stand-in for a one-workgroup transform on mapped memory: 512 threads, 8 words each over the link, some
// arithmetic, 8 words back; `flag` != nullptr: the workgroup itself publishes `seq` behind its stores
__global__ void __launch_bounds__(512) surrogate_kernel(unsigned long long* data, int rounds, unsigned* flag, unsigned seq) {
  unsigned long long v[8];
  for (int i = 0; i < 8; ++i) v[i] = data[threadIdx.x + 512 * i];
  for (int r = 0; r < rounds; ++r)
    for (int i = 0; i < 8; ++i) v[i] = v[i] * 6364136223846793005ull + v[(i + 1) & 7];
  for (int i = 0; i < 8; ++i) data[threadIdx.x + 512 * i] = v[i];
  if (flag) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <class F>
static double median_us(int iters, F body) {
  for (int i = 0; i < 50; ++i) body();
  std::vector<double> t((size_t)iters);
  for (int i = 0; i < iters; ++i) {
    const double t0 = now_us();
    body();
    t[(size_t)i] = now_us() - t0;
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 1000;
  for (uint64_t n : {4096ull, 16384ull, 65536ull}) {
    uint64_t q = 0;
    if (hexl_amd_generate_primes(&q, 1, 54, 1, n) != 1) return 2;
    hexl_amd_ntt* plan = nullptr;
    OK(hexl_amd_ntt_create(&plan, n, q, 0, 0));
    void* host = nullptr;
    OK(hexl_amd_host_alloc(&host, n * 8));
    uint64_t* h = (uint64_t*)host;
    for (uint64_t i = 0; i < n; ++i) h[i] = i % q;
    void* dev = nullptr;
    HIP(hipHostGetDevicePointer(&dev, host, 0));
    unsigned* flag = nullptr;
    HIP(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped));
    *flag = 0;
    unsigned* flag_dev = nullptr;
    HIP(hipHostGetDevicePointer((void**)&flag_dev, flag, 0));
    hipStream_t st;
    HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev_block, ev_spin;
    HIP(hipEventCreateWithFlags(&ev_block, hipEventDisableTiming | hipEventBlockingSync));
    HIP(hipEventCreateWithFlags(&ev_spin, hipEventDisableTiming));
    auto work = [&] { OK(hexl_amd_ntt_forward(plan, (uint64_t*)dev, (const uint64_t*)dev, 1, 1, 4, st)); };
    unsigned seq = 0;
    const double t_sync = median_us(iters, [&] {
      work();
      HIP(hipStreamSynchronize(st));
    });
    const double t_evb = median_us(iters, [&] {
      work();
      HIP(hipEventRecord(ev_block, st));
      HIP(hipEventSynchronize(ev_block));
    });
    const double t_evs = median_us(iters, [&] {
      work();
      HIP(hipEventRecord(ev_spin, st));
      while (hipEventQuery(ev_spin) == hipErrorNotReady) {
      }
    });
    const double t_query = median_us(iters, [&] {
      work();
      while (hipStreamQuery(st) == hipErrorNotReady) {
      }
    });
    const double t_flagk = median_us(iters, [&] {
      work();
      ++seq;
      hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, st, flag_dev, seq);
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
    });
    HIP(hipStreamSynchronize(st));
    double t_writev = -1;
    {
      ++seq;
      hipError_t e = hipStreamWriteValue32(st, flag_dev, seq, 0);
      if (e == hipSuccess) {
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
        t_writev = median_us(iters, [&] {
          work();
          ++seq;
          HIP(hipStreamWriteValue32(st, flag_dev, seq, 0));
          while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
        });
      } else {
        std::fprintf(stderr, "hipStreamWriteValue32: %s\n", hipGetErrorString(e));
        (void)hipGetLastError();
      }
    }
    HIP(hipStreamSynchronize(st));
    if (n == 4096) {
      for (int rounds : {0, 40, 80}) {
        auto sur = [&](unsigned* f, unsigned sq) {
          hipLaunchKernelGGL(surrogate_kernel, dim3(1), dim3(512), 0, st, (unsigned long long*)dev, rounds, f, sq);
        };
        const double a = median_us(iters, [&] {
          sur(nullptr, 0);
          HIP(hipStreamSynchronize(st));
        });
        const double b = median_us(iters, [&] {
          sur(nullptr, 0);
          ++seq;
          hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, st, flag_dev, seq);
          while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
        });
        const double c = median_us(iters, [&] {
          ++seq;
          sur(flag_dev, seq);
          while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
        });
        HIP(hipStreamSynchronize(st));
        std::printf("{\"surrogate_rounds\": %d, \"sync_us\": %.2f, \"flag_kernel_spin_us\": %.2f, \"in_kernel_flag_spin_us\": %.2f}\n",
                    rounds, a, b, c);
      }
    }
    const double t_launch = median_us(iters, [&] { work(); });
    HIP(hipStreamSynchronize(st));
    std::printf(
        "{\"n\": %llu, \"sync_us\": %.2f, \"event_blocking_us\": %.2f, \"event_query_spin_us\": %.2f, "
        "\"stream_query_spin_us\": %.2f, \"flag_kernel_spin_us\": %.2f, \"write_value_spin_us\": %.2f, "
        "\"launch_only_us\": %.2f}\n",
        (unsigned long long)n, t_sync, t_evb, t_evs, t_query, t_flagk, t_writev, t_launch);
    std::fflush(stdout);
    hexl_amd_ntt_destroy(plan);
  }
  return 0;
}
