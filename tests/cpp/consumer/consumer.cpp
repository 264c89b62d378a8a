// Minimal consumer of the installed package: the reference's example shape
// (example/example.cpp:110-144 ExampleNTT) through find_package(HEXL), plus one use of every
// header the reference's umbrella exposes beyond the hot path (hexl/include/hexl/hexl.hpp:6-26:
// logging, the internal SEAL entry points) and of ntt-cache.hpp / locks.hpp.
#include <cstdio>
#include <thread>
#include <vector>

#include "hexl/experimental/seal/ntt-cache.hpp"
#include "hexl/hexl.hpp"

int main(int argc, char** argv) {
  START_EASYLOGGINGPP(argc, argv);
  HEXL_VLOG(3, "consumer starting with " << argc << " argument(s)");
  const uint64_t N = 8, modulus = 769;
  std::vector<uint64_t> arg{1, 2, 3, 4, 5, 6, 7, 8};
  auto exp_out = arg;
  intel::hexl::NTT ntt(N, modulus);
  ntt.ComputeForward(arg.data(), arg.data(), 1, 1);
  const auto transformed = arg;
  ntt.ComputeInverse(arg.data(), arg.data(), 1, 1);
  if (arg != exp_out) {
    printf("round trip mismatch\n");
    return 1;
  }
  // GetNTT: the same object from every thread, the same transform as a fresh NTT
  intel::hexl::NTT* seen[2] = {nullptr, nullptr};
  std::thread t0([&] { seen[0] = &intel::hexl::GetNTT(N, modulus); });
  std::thread t1([&] { seen[1] = &intel::hexl::GetNTT(N, modulus); });
  t0.join();
  t1.join();
  intel::hexl::NTT& cached = intel::hexl::GetNTT(N, modulus);
  if (seen[0] != &cached || seen[1] != &cached || &intel::hexl::GetNTT(2 * N, modulus) == &cached) {
    printf("GetNTT does not cache per (N, modulus)\n");
    return 1;
  }
  auto again = exp_out;
  cached.ComputeForward(again.data(), again.data(), 1, 1);
  if (again != transformed) {
    printf("GetNTT transform mismatch\n");
    return 1;
  }
  {
    intel::hexl::RWLock lock;
    intel::hexl::ReadLock r = lock.AcquireRead();
    if (lock.TryAcquireWrite().owns_lock()) {
      printf("RWLock handed out a write lock under a reader\n");
      return 1;
    }
  }
  // internal::DyadicMultiply == DyadicMultiply (two moduli, n = 8)
  const uint64_t moduli[2] = {769, 12289};
  std::vector<uint64_t> x(2 * 2 * N), y(2 * 2 * N), r1(3 * 2 * N), r2(3 * 2 * N);
  for (size_t i = 0; i < x.size(); ++i) {
    x[i] = (7 * i + 3) % 769;
    y[i] = (11 * i + 5) % 769;
  }
  intel::hexl::DyadicMultiply(r1.data(), x.data(), y.data(), N, moduli, 2);
  intel::hexl::internal::DyadicMultiply(r2.data(), x.data(), y.data(), N, moduli, 2);
  if (r1 != r2 || r1[0] != (x[0] * y[0]) % 769) {
    printf("internal::DyadicMultiply mismatch\n");
    return 1;
  }
  // internal::KeySwitch carries the reference's contract (key-switch-internal.cpp:31-34)
  bool threw = false;
  try {
    intel::hexl::internal::KeySwitch(r1.data(), x.data(), N, 1, 2, 2, 2, moduli, nullptr, nullptr,
                                     x.data());
  } catch (const std::exception&) {
    threw = true;
  }
  if (!threw) {
    printf("internal::KeySwitch accepted root_of_unity_powers_ptr\n");
    return 1;
  }
  HEXL_VLOG(3, "consumer done");
  printf("consumer OK\n");
  return 0;
}
