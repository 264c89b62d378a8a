// Minimal consumer of the installed package: the reference's example shape
// (example/example.cpp:110-144 ExampleNTT) through find_package(HEXL).
#include <cstdio>
#include <vector>

#include "hexl/hexl.hpp"

int main() {
  const uint64_t N = 8, modulus = 769;
  std::vector<uint64_t> arg{1, 2, 3, 4, 5, 6, 7, 8};
  auto exp_out = arg;
  intel::hexl::NTT ntt(N, modulus);
  ntt.ComputeForward(arg.data(), arg.data(), 1, 1);
  ntt.ComputeInverse(arg.data(), arg.data(), 1, 1);
  if (arg != exp_out) {
    printf("round trip mismatch\n");
    return 1;
  }
  printf("consumer OK\n");
  return 0;
}
