// host_call_budget.cpp -- where the microseconds of ONE synchronous intel::hexl::NTT call on
// ordinary host memory go (the unmodified one-polynomial caller, the
// reference's own per-call convention benchmark/bench-ntt.cpp:212-239), measured from C++ with
// clock_gettime -- no Python, no ctypes in the path.
//
// The call, for a one- or two-kernel transform of at most 256 KiB (capi.cpp: ntt_run_host), is
//     classify the pointers -> memcpy into the pinned mapped bounce buffer -> launch(es) ->
//     wait for the stream -> memcpy out
// Each phase is timed on its own through entry points that do only that phase:
//     classify      hexl_amd_pointer_kind on the caller's vector
//     memcpy        N words into / out of memory from hexl_amd_host_alloc (the bounce buffer's kind)
//     launch        hexl_amd_ntt_forward on a DEVICE buffer, no wait (enqueue cost alone)
//     launch+wait   hexl_amd_ntt_forward_host in place on mapped memory (kind 2: no staging copies,
//                   the kernel reads and writes over the link exactly as it does on the bounce buffer)
//     call          intel::hexl::NTT::ComputeForward on a std::vector, in place
// and the line reports call - (classify + 2 memcpy + launch+wait) as `unaccounted`.
//
//   g++ -std=c++17 -O2 -Iinclude tests/cpp/host_call_budget.cpp -Lhexl_amd/lib -lhexl -lhexl_amd
//       -Wl,-rpath,$PWD/hexl_amd/lib -pthread -o tests/cpp/host_call_budget
//   tests/cpp/host_call_budget [iterations [host_bounce_kb]]      -> one JSON line per degree
#include <time.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hexl/hexl.hpp"
#include "hexl_amd.h"

namespace {

double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

#define OK(call)                                                                         \
  do {                                                                                   \
    if ((call) != 0) {                                                                   \
      std::fprintf(stderr, "%s failed: %s\n", #call, hexl_amd_last_error());             \
      std::exit(2);                                                                      \
    }                                                                                    \
  } while (0)

// median of per-iteration times of `body`, after a warm-up
template <class F>
double median_us(int iters, F body) {
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

void budget(uint64_t n, int bits, int iters) {
  uint64_t q = 0;
  if (hexl_amd_generate_primes(&q, 1, (size_t)bits, 1, n) != 1) std::exit(2);
  intel::hexl::NTT ntt(n, q);
  std::vector<uint64_t> v(n), ref(n);
  uint64_t s = 12345;
  for (auto& x : v) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    x = (s >> 3) % q;
  }
  const auto input = v;
  ntt.ComputeForward(ref.data(), v.data(), 1, 1);

  // the whole call, as an unmodified caller makes it (in place; and out of place)
  const double call = median_us(iters, [&] { ntt.ComputeForward(v.data(), v.data(), 1, 1); });
  // the same call waiting in hipStreamSynchronize instead of polling the completion flag (round 6 A/B)
  OK(hexl_amd_set_tuning("host_poll", 0));
  const double call_sync = median_us(iters, [&] { ntt.ComputeForward(v.data(), v.data(), 1, 1); });
  OK(hexl_amd_set_tuning("host_poll", 1));
  std::vector<uint64_t> out(n);
  const double call_oop =
      median_us(iters, [&] { ntt.ComputeForward(out.data(), input.data(), 1, 1); });
  if (out != ref) {
    std::fprintf(stderr, "n=%llu: out-of-place result differs\n", (unsigned long long)n);
    std::exit(3);
  }

  // classify
  volatile int kind_sink = 0;
  const double classify = median_us(iters, [&] { kind_sink = hexl_amd_pointer_kind(v.data()); });
  (void)kind_sink;

  // memcpy into / out of pinned mapped memory
  void* pinned = nullptr;
  OK(hexl_amd_host_alloc(&pinned, n * 8));
  const double copy_in = median_us(iters, [&] { std::memcpy(pinned, input.data(), n * 8); });
  const double copy_out = median_us(iters, [&] { std::memcpy(out.data(), pinned, n * 8); });

  // launch + wait on mapped memory (what the call does between its two memcpys)
  hexl_amd_ntt* plan = nullptr;
  OK(hexl_amd_ntt_create(&plan, n, q, 0, -1));
  std::memcpy(pinned, input.data(), n * 8);
  const double launch_wait = median_us(iters, [&] {
    OK(hexl_amd_ntt_forward_host(plan, (uint64_t*)pinned, (const uint64_t*)pinned, 1, 1, 4));
  });

  // launch alone: device buffer, private stream, no wait (drained outside the timed region)
  void* dev = nullptr;
  void* stream = nullptr;
  OK(hexl_amd_device_alloc(&dev, n * 8, -1));
  OK(hexl_amd_stream_create(&stream, -1));
  OK(hexl_amd_copy(dev, input.data(), n * 8, stream, 1));
  std::vector<double> lt;
  for (int rep = 0; rep < iters / 16 + 4; ++rep) {
    const double t0 = now_us();
    for (int k = 0; k < 16; ++k)
      OK(hexl_amd_ntt_forward(plan, (uint64_t*)dev, (const uint64_t*)dev, 1, 4, 4, stream));
    lt.push_back((now_us() - t0) / 16);
    OK(hexl_amd_synchronize(stream));
  }
  std::sort(lt.begin(), lt.end());
  const double launch = lt[lt.size() / 2];
  // ... and the wait of an already finished stream (the floor of hipStreamSynchronize)
  const double idle_wait = median_us(iters, [&] { OK(hexl_amd_synchronize(stream)); });
  // device-resident call + wait: launch + kernel + completion, no link traffic
  const double dev_call = median_us(iters, [&] {
    OK(hexl_amd_ntt_forward(plan, (uint64_t*)dev, (const uint64_t*)dev, 1, 4, 4, stream));
    OK(hexl_amd_synchronize(stream));
  });

  const double parts = classify + copy_in + launch_wait + copy_out;
  std::printf(
      "{\"n\": %llu, \"bits\": %d, \"iterations\": %d, \"call_us\": %.2f, \"call_out_of_place_us\": %.2f, \"call_stream_synchronize_us\": %.2f, "
      "\"classify_us\": %.2f, \"memcpy_in_us\": %.2f, \"launch_and_wait_mapped_us\": %.2f, "
      "\"memcpy_out_us\": %.2f, \"sum_of_parts_us\": %.2f, \"unaccounted_us\": %.2f, "
      "\"launch_only_us\": %.2f, \"idle_synchronize_us\": %.2f, \"device_call_and_wait_us\": %.2f}\n",
      (unsigned long long)n, bits, iters, call, call_oop, call_sync, classify, copy_in, launch_wait, copy_out,
      parts, call - parts, launch, idle_wait, dev_call);
  std::fflush(stdout);
  OK(hexl_amd_stream_destroy(stream));
  OK(hexl_amd_device_free(dev));
  OK(hexl_amd_ntt_destroy(plan));
  OK(hexl_amd_host_free(pinned));
}

}  // namespace

int main(int argc, char** argv) {
  const int iters = argc >= 2 ? std::atoi(argv[1]) : 2000;
  if (argc >= 3) OK(hexl_amd_set_tuning("host_bounce_kb", (uint64_t)std::atoll(argv[2])));  // (A/B of the limit)
  budget(4096, 49, iters);
  budget(8192, 54, iters);
  budget(16384, 54, iters);
  budget(65536, 54, iters / 2);
  if (argc >= 4) budget(131072, 54, iters / 2);  // (any fourth argument: the first degree beyond the bounce limit)
  return 0;
}
