// ks_call_cost.cpp -- what one KeySwitch call (one ciphertext, device buffers: the reference's call,
// key-switch-internal.cpp:25-201) costs a C++ caller: the host time of the enqueue alone (the call returns without
// synchronising) and the wall time per call of a stream of calls -- launch by launch ("ks_graph" 0) and replayed from
// the captured graph, on one ciphertext and walking over eight (bench.py's composites block measures the same call
// through Python, whose per-call overhead is of the order of the difference between the two paths).
//   g++ -std=c++17 -O2 -Iinclude tests/cpp/ks_call_cost.cpp -Lhexl_amd/lib -lhexl_amd -Wl,-rpath,$PWD/hexl_amd/lib
//       -o tests/cpp/ks_call_cost
//   tests/cpp/ks_call_cost [n [D]]      -> one JSON line per mode
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

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::atoll(argv[1]) : 16384, D = argc > 2 ? std::atoll(argv[2]) : 7, C = 2, K = D + 1;
  const int ring = 8, calls = 400;
  std::vector<uint64_t> moduli(K);
  if (hexl_amd_generate_primes(moduli.data(), K, 54, 1, n) != K) return 2;
  void* stream = nullptr;
  OK(hexl_amd_stream_create(&stream, -1));
  std::vector<const uint64_t*> keys(D);
  for (uint64_t j = 0; j < D; ++j) {
    void* p = nullptr;
    OK(hexl_amd_device_alloc(&p, C * K * n * 8, -1));
    for (uint64_t c = 0; c < C; ++c)
      for (uint64_t i = 0; i < K; ++i)
        OK(hexl_amd_fill_splitmix((uint64_t*)p + (c * K + i) * n, n, 1, 100 + j * 31 + c * 7 + i, moduli[i], stream));
    keys[j] = (const uint64_t*)p;
  }
  std::vector<uint64_t> msf(D);
  for (uint64_t i = 0; i < D; ++i) msf[i] = 12345 + i;
  std::vector<uint64_t*> res(ring), tgt(ring);
  for (int r = 0; r < ring; ++r) {
    void *a = nullptr, *b = nullptr;
    OK(hexl_amd_device_alloc(&a, C * D * n * 8, -1));
    OK(hexl_amd_device_alloc(&b, D * n * 8, -1));
    res[r] = (uint64_t*)a;
    tgt[r] = (uint64_t*)b;
    for (uint64_t i = 0; i < D; ++i) {
      OK(hexl_amd_fill_splitmix(tgt[r] + i * n, n, 1, 7 + r * 13 + i, moduli[i], stream));
      for (uint64_t c = 0; c < C; ++c)
        OK(hexl_amd_fill_splitmix(res[r] + (c * D + i) * n, n, 1, 900 + r + c * 3 + i, moduli[i], stream));
    }
  }
  OK(hexl_amd_synchronize(stream));
  auto run = [&](int graph, int walk) {
    OK(hexl_amd_set_tuning("ks_graph", graph));
    auto call = [&](int k) {
      const int r = walk ? k % ring : 0;
      OK(hexl_amd_key_switch(res[r], tgt[r], n, D, K, D + 1, C, moduli.data(), keys.data(), msf.data(), stream));
    };
    for (int k = 0; k < 40; ++k) call(k);
    OK(hexl_amd_synchronize(stream));
    std::vector<double> host((size_t)calls);
    const double t0 = now_us();
    for (int k = 0; k < calls; ++k) {
      const double a = now_us();
      call(k);
      host[(size_t)k] = now_us() - a;
    }
    OK(hexl_amd_synchronize(stream));
    const double wall = (now_us() - t0) / calls;
    std::sort(host.begin(), host.end());
    // ... and the latency of a call that is waited for before the next one is made
    std::vector<double> lat((size_t)calls / 4);
    for (size_t k = 0; k < lat.size(); ++k) {
      const double a = now_us();
      call((int)k);
      OK(hexl_amd_synchronize(stream));
      lat[k] = now_us() - a;
    }
    std::sort(lat.begin(), lat.end());
    uint64_t rep = 0, eag = 0;
    hexl_amd_get_counter("ks_graph_replays", &rep);
    hexl_amd_get_counter("ks_eager", &eag);
    std::printf("{\"n\": %llu, \"D\": %llu, \"ks_graph\": %d, \"buffers\": \"%s\", \"host_enqueue_us\": %.1f, "
                "\"wall_us_per_call\": %.1f, \"call_and_wait_us\": %.1f, \"replays_so_far\": %llu, \"eager_so_far\": %llu}\n",
                (unsigned long long)n, (unsigned long long)D, graph, walk ? "walking over 8 ciphertexts" : "one ciphertext",
                host[host.size() / 2], wall, lat[lat.size() / 2], (unsigned long long)rep, (unsigned long long)eag);
    std::fflush(stdout);
  };
  if (argc > 3) {  // (developer: one mode only)
    run(std::atoi(argv[3]), 1);
    return 0;
  }
  run(0, 1);
  run(0, 0);
  run(1, 1);
  run(1, 0);
  return 0;
}
