// eager_rate.cpp -- BASELINE configs[1] (N = 4096, 50-bit prime, 256 polynomials) as a C++ caller of the C-ABI issues it:
// calls back to back on one stream, no interpreter between them -- wall time per call of 2000 calls (the stream drained
// once at the end) for the forward and inverse transform and EltwiseMultMod.  bench.py reports it beside the same calls
// made through Python (whose per-call overhead is of the order of the 6-7 us these kernels take).
//   g++ -std=c++17 -O2 -Iinclude tests/cpp/eager_rate.cpp -Lhexl_amd/lib -lhexl_amd -Wl,-rpath,$PWD/hexl_amd/lib
//       -o tests/cpp/eager_rate
#include <time.h>

#include <cstdio>
#include <cstdlib>
#include <functional>

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

int main() {
  const uint64_t n = 4096, batch = 256, q = 562949954093057ull;
  void* stream = nullptr;
  OK(hexl_amd_stream_create(&stream, -1));
  hexl_amd_ntt* plan = nullptr;
  OK(hexl_amd_ntt_create(&plan, n, q, 0, -1));
  void *x = nullptr, *y = nullptr;
  OK(hexl_amd_device_alloc(&x, n * batch * 8, -1));
  OK(hexl_amd_device_alloc(&y, n * batch * 8, -1));
  OK(hexl_amd_fill_splitmix((uint64_t*)x, n, batch, 1, q, stream));
  uint64_t *X = (uint64_t*)x, *Y = (uint64_t*)y;
  auto wall = [&](const std::function<void()>& fn) {
    const int calls = 2000;
    for (int i = 0; i < 200; ++i) fn();
    OK(hexl_amd_synchronize(stream));
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      const double t0 = now_us();
      for (int i = 0; i < calls; ++i) fn();
      OK(hexl_amd_synchronize(stream));
      const double t = (now_us() - t0) / calls;
      if (t < best) best = t;
    }
    return best;
  };
  const double f = wall([&] { OK(hexl_amd_ntt_forward(plan, Y, X, batch, 1, 1, stream)); });
  const double i = wall([&] { OK(hexl_amd_ntt_inverse(plan, Y, X, batch, 1, 1, stream)); });
  const double m = wall([&] { OK(hexl_amd_eltwise_mult_mod(Y, X, X, n * batch, q, 1, stream)); });
  std::printf("{\"n\": %llu, \"batch\": %llu, \"fwd_us\": %.2f, \"inv_us\": %.2f, \"multmod_us\": %.2f}\n",
              (unsigned long long)n, (unsigned long long)batch, f, i, m);
  return 0;
}
