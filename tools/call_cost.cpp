// call_cost.cpp -- host time of ONE enqueue of each device-pointer entry point, from C++ (median over 300 calls, the stream
// drained every 16 so that no call waits for queue space): anything far above a plain kernel launch (3-4 us) is host work
// of the library's own.  (Round 6: this is how KeySwitch's per-call primality tests were found.)
//   g++ -std=c++17 -O2 -Iinclude tools/call_cost.cpp -Lhexl_amd/lib -lhexl_amd -Wl,-rpath,$PWD/hexl_amd/lib -o tools/call_cost
#include <time.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
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

static void* g_stream;
static void cost(const char* name, const std::function<void()>& fn) {
  for (int i = 0; i < 32; ++i) fn();
  OK(hexl_amd_synchronize(g_stream));
  std::vector<double> t;
  for (int rep = 0; rep < 20; ++rep) {
    for (int i = 0; i < 16; ++i) {
      const double a = now_us();
      fn();
      t.push_back(now_us() - a);
    }
    OK(hexl_amd_synchronize(g_stream));
  }
  std::sort(t.begin(), t.end());
  std::printf("%-46s %6.2f us per enqueue (p90 %.2f)\n", name, t[t.size() / 2], t[t.size() * 9 / 10]);
  std::fflush(stdout);
}

int main() {
  const uint64_t n = 4096, polys = 8;
  OK(hexl_amd_stream_create(&g_stream, -1));
  std::vector<uint64_t> q(8);
  if (hexl_amd_generate_primes(q.data(), 8, 54, 1, 16384) != 8) return 2;
  void *a = nullptr, *b = nullptr, *r = nullptr;
  const uint64_t words = 16384 * 64;
  OK(hexl_amd_device_alloc(&a, words * 8, -1));
  OK(hexl_amd_device_alloc(&b, words * 8, -1));
  OK(hexl_amd_device_alloc(&r, words * 8, -1));
  OK(hexl_amd_fill_splitmix((uint64_t*)a, words, 1, 1, q[0], g_stream));
  OK(hexl_amd_fill_splitmix((uint64_t*)b, words, 1, 2, q[0], g_stream));
  uint64_t *A = (uint64_t*)a, *B = (uint64_t*)b, *R = (uint64_t*)r;
  std::vector<hexl_amd_ntt*> plans(8);
  for (int i = 0; i < 8; ++i) OK(hexl_amd_ntt_create(&plans[i], 16384, q[i], 0, -1));
  hexl_amd_ntt* p4096 = nullptr;
  uint64_t q4096 = 0;
  if (hexl_amd_generate_primes(&q4096, 1, 49, 1, 4096) != 1) return 2;
  OK(hexl_amd_ntt_create(&p4096, 4096, q4096, 0, -1));
  const uint64_t m = q[0], ne = n * polys;
  cost("ntt_forward N=4096 x 8", [&] { OK(hexl_amd_ntt_forward(p4096, A, A, polys, 1, 1, g_stream)); });
  cost("ntt_inverse N=4096 x 8", [&] { OK(hexl_amd_ntt_inverse(p4096, A, A, polys, 1, 1, g_stream)); });
  cost("ntt_forward N=16384 x 8 (two passes)", [&] { OK(hexl_amd_ntt_forward(plans[0], A, A, 8, 1, 1, g_stream)); });
  cost("ntt_forward_rns 8 plans x 8 (N=16384)",
       [&] { OK(hexl_amd_ntt_forward_rns(plans.data(), 8, R, A, 8, 1, 1, g_stream)); });
  std::vector<uint8_t> slot = {0, 1, 2, 3, 4, 5, 6, 7};
  cost("ntt_forward_map 8 plans, 64 polys (N=16384)",
       [&] { OK(hexl_amd_ntt_forward_map(plans.data(), 8, slot.data(), 8, 1, R, A, 64, 1, 1, g_stream)); });
  cost("eltwise_add_mod", [&] { OK(hexl_amd_eltwise_add_mod(R, A, B, ne, m, g_stream)); });
  cost("eltwise_sub_mod_scalar", [&] { OK(hexl_amd_eltwise_sub_mod_scalar(R, A, 5, ne, m, g_stream)); });
  cost("eltwise_mult_mod", [&] { OK(hexl_amd_eltwise_mult_mod(R, A, B, ne, m, 1, g_stream)); });
  cost("eltwise_fma_mod", [&] { OK(hexl_amd_eltwise_fma_mod(R, A, 12345, B, ne, m, 1, g_stream)); });
  cost("eltwise_reduce_mod (q -> 1)", [&] { OK(hexl_amd_eltwise_reduce_mod(R, A, ne, m, m, 1, g_stream)); });
  cost("eltwise_reduce_fma_mod", [&] { OK(hexl_amd_eltwise_reduce_fma_mod(R, A, 12345, B, ne, m, m, g_stream)); });
  cost("eltwise_cmp_add", [&] { OK(hexl_amd_eltwise_cmp_add(R, A, ne, 0, 17, 3, g_stream)); });
  cost("eltwise_cmp_sub_mod", [&] { OK(hexl_amd_eltwise_cmp_sub_mod(R, A, ne, m, 0, 17, 3, g_stream)); });
  cost("dyadic_multiply n=4096 x 8 moduli",
       [&] { OK(hexl_amd_dyadic_multiply(R, A, B, 4096, q.data(), 8, g_stream)); });
  return 0;
}
