// test_shim_debug.cpp -- the debug contract of the drop-in API: compiled with -DHEXL_DEBUG and
// linked against libhexl_debug.so (the shim built with HEXL_DEBUG, like the reference's
// hexl_debug library), out-of-range ELEMENTS throw -- the reference's TEST(NTT, bad_input)
// (test/test-ntt.cpp:20-94) statement by statement, the null / bad-input blocks of
// test/test-eltwise-*.cpp (#ifdef HEXL_DEBUG), and the same for device memory.  Needs a GPU.
#include <cstdio>
#include <vector>

#include "hexl/hexl.hpp"

using namespace intel::hexl;

static int g_fail = 0;
#define EXPECT_ANY_THROW(stmt)                                              \
  do {                                                                      \
    bool threw = false;                                                     \
    try {                                                                   \
      stmt;                                                                 \
    } catch (...) {                                                         \
      threw = true;                                                         \
    }                                                                       \
    if (!threw) {                                                           \
      std::printf("FAIL %s:%d: no throw: %s\n", __FILE__, __LINE__, #stmt); \
      ++g_fail;                                                             \
    }                                                                       \
  } while (0)
#define EXPECT_NO_THROW(stmt)                                                              \
  do {                                                                                     \
    try {                                                                                  \
      stmt;                                                                                \
    } catch (const std::exception& e) {                                                    \
      std::printf("FAIL %s:%d: threw (%s): %s\n", __FILE__, __LINE__, e.what(), #stmt);    \
      ++g_fail;                                                                            \
    }                                                                                      \
  } while (0)

typedef std::vector<uint64_t> V;

static void ntt_bad_input(bool mapped) {  // TEST(NTT, bad_input), test/test-ntt.cpp:21-93
  const uint64_t N = 8, modulus = 769;
  AlignedVector64<uint64_t> input, p_input, p_times_2_input, p_times_4_input;
  NTT ntt(N, modulus);
  auto make = [&](std::initializer_list<uint64_t> v) {
    AlignedVector64<uint64_t> r = mapped ? DeviceMappedVector(0) : AlignedVector64<uint64_t>();
    r.assign(v);
    return r;
  };
  auto fill = [&](uint64_t value) {
    AlignedVector64<uint64_t> r = mapped ? DeviceMappedVector(0) : AlignedVector64<uint64_t>();
    r.assign(N, value);
    return r;
  };
  auto init_inputs = [&]() {
    input = make({1, 2, 3, 4, 5, 6, 7, 8});
    p_input = fill(modulus);
    p_times_2_input = fill(2 * modulus);
    p_times_4_input = fill(4 * modulus);
  };
  // Forward transform: bad input
  init_inputs();
  EXPECT_ANY_THROW(ntt.ComputeForward(input.data(), nullptr, 1, 1));
  init_inputs();
  EXPECT_ANY_THROW(ntt.ComputeForward(nullptr, input.data(), 1, 1));
  init_inputs();
  EXPECT_NO_THROW(ntt.ComputeForward(input.data(), input.data(), 1, 1));
  init_inputs();
  EXPECT_NO_THROW(ntt.ComputeForward(p_input.data(), p_input.data(), 4, 4));
  init_inputs();
  EXPECT_ANY_THROW(ntt.ComputeForward(p_times_2_input.data(), p_times_2_input.data(), 2, 1));
  init_inputs();
  EXPECT_NO_THROW(ntt.ComputeForward(p_times_2_input.data(), p_times_2_input.data(), 4, 4));
  init_inputs();
  EXPECT_ANY_THROW(ntt.ComputeForward(p_times_4_input.data(), p_times_4_input.data(), 4, 4));
  init_inputs();
  // Bad mod factors
  EXPECT_NO_THROW(ntt.ComputeForward(input.data(), input.data(), 2, 1));
  init_inputs();
  EXPECT_ANY_THROW(ntt.ComputeForward(input.data(), input.data(), 123, 1));
  init_inputs();
  EXPECT_ANY_THROW(ntt.ComputeForward(input.data(), input.data(), 2, 123));
  init_inputs();
  // Inverse transform: bad input
  EXPECT_ANY_THROW(ntt.ComputeInverse(input.data(), nullptr, 1, 1));
  init_inputs();
  EXPECT_ANY_THROW(ntt.ComputeInverse(nullptr, input.data(), 1, 1));
  init_inputs();
  EXPECT_NO_THROW(ntt.ComputeInverse(input.data(), input.data(), 1, 1));
  init_inputs();
  EXPECT_ANY_THROW(ntt.ComputeInverse(p_input.data(), p_input.data(), 1, 1));
  init_inputs();
  EXPECT_NO_THROW(ntt.ComputeInverse(p_input.data(), p_input.data(), 2, 2));
  init_inputs();
  EXPECT_ANY_THROW(ntt.ComputeInverse(p_times_2_input.data(), p_times_2_input.data(), 2, 2));
  init_inputs();
  // Bad mod factors
  EXPECT_NO_THROW(ntt.ComputeInverse(input.data(), input.data(), 1, 1));
  init_inputs();
  EXPECT_ANY_THROW(ntt.ComputeInverse(input.data(), input.data(), 123, 1));
  init_inputs();
  EXPECT_ANY_THROW(ntt.ComputeInverse(input.data(), input.data(), 1, 123));
}

static void eltwise_bad_input() {
  // test/test-eltwise-mult-mod.cpp, -add-mod.cpp, -sub-mod.cpp, -fma-mod.cpp: #ifdef HEXL_DEBUG
  // blocks (null pointers, n == 0, modulus ranges, out-of-range elements)
  const uint64_t modulus = 769;
  V op1{1, 2, 3, 4, 5, 6, 7, 8}, op2{1, 2, 3, 4, 5, 6, 7, 8}, big{1, 2, 3, 4, 5, 6, 7, 769},
      result(8, 0);
  EXPECT_ANY_THROW(EltwiseMultMod(nullptr, op1.data(), op2.data(), 8, modulus, 1));
  EXPECT_ANY_THROW(EltwiseMultMod(result.data(), nullptr, op2.data(), 8, modulus, 1));
  EXPECT_ANY_THROW(EltwiseMultMod(result.data(), op1.data(), nullptr, 8, modulus, 1));
  EXPECT_ANY_THROW(EltwiseMultMod(result.data(), op1.data(), op2.data(), 0, modulus, 1));
  EXPECT_ANY_THROW(EltwiseMultMod(result.data(), op1.data(), op2.data(), 8, 1, 1));
  EXPECT_ANY_THROW(EltwiseMultMod(result.data(), op1.data(), op2.data(), 8, modulus, 3));
  EXPECT_NO_THROW(EltwiseMultMod(result.data(), op1.data(), op2.data(), 8, modulus, 1));
  EXPECT_ANY_THROW(EltwiseMultMod(result.data(), big.data(), op2.data(), 8, modulus, 1));
  EXPECT_ANY_THROW(EltwiseMultMod(result.data(), op1.data(), big.data(), 8, modulus, 1));
  EXPECT_NO_THROW(EltwiseMultMod(result.data(), big.data(), op2.data(), 8, modulus, 2));

  EXPECT_ANY_THROW(EltwiseAddMod(result.data(), big.data(), op2.data(), 8, modulus));
  EXPECT_ANY_THROW(EltwiseAddMod(result.data(), op1.data(), big.data(), 8, modulus));
  EXPECT_ANY_THROW(EltwiseAddMod(result.data(), big.data(), uint64_t{3}, 8, modulus));
  EXPECT_ANY_THROW(EltwiseAddMod(result.data(), op1.data(), uint64_t{769}, 8, modulus));
  EXPECT_NO_THROW(EltwiseAddMod(result.data(), op1.data(), op2.data(), 8, modulus));
  EXPECT_ANY_THROW(EltwiseSubMod(result.data(), big.data(), op2.data(), 8, modulus));
  EXPECT_ANY_THROW(EltwiseSubMod(result.data(), op1.data(), big.data(), 8, modulus));
  EXPECT_ANY_THROW(EltwiseSubMod(result.data(), big.data(), uint64_t{3}, 8, modulus));
  EXPECT_NO_THROW(EltwiseSubMod(result.data(), op1.data(), op2.data(), 8, modulus));

  EXPECT_ANY_THROW(EltwiseFMAMod(result.data(), big.data(), 2, nullptr, 8, modulus, 1));
  EXPECT_ANY_THROW(EltwiseFMAMod(result.data(), op1.data(), 2, big.data(), 8, modulus, 1));
  EXPECT_ANY_THROW(EltwiseFMAMod(result.data(), op1.data(), 769, nullptr, 8, modulus, 1));
  EXPECT_NO_THROW(EltwiseFMAMod(result.data(), big.data(), 2, op2.data(), 8, modulus, 2));
  EXPECT_ANY_THROW(EltwiseReduceMod(result.data(), big.data(), 8, modulus, 1, 1));
  // (in == out with distinct buffers is a plain copy, eltwise-reduce-mod.cpp:94-99)
  EXPECT_NO_THROW(EltwiseReduceMod(result.data(), big.data(), 8, modulus, 2, 2));
  V two_q(8, 2 * modulus);
  EXPECT_ANY_THROW(EltwiseReduceMod(result.data(), two_q.data(), 8, modulus, 2, 1));
  EXPECT_NO_THROW(EltwiseReduceMod(result.data(), two_q.data(), 8, modulus, 4, 1));
  EXPECT_NO_THROW(EltwiseReduceMod(result.data(), two_q.data(), 8, modulus, modulus, 1));
}

int main() {
  ntt_bad_input(false);  // ordinary host buffers, as in the reference's test
  ntt_bad_input(true);   // device-mapped memory: the check runs as a kernel
  eltwise_bad_input();
  if (g_fail) {
    std::printf("%d checks failed\n", g_fail);
    return 1;
  }
  std::printf("all debug-contract checks passed\n");
  return 0;
}
