// test_shim.cpp -- exercises the drop-in C++ API (namespace intel::hexl from
// include/hexl/hexl.hpp, libhexl.so) exactly as a HEXL caller would: host
// std::vector buffers, one polynomial per call.  The vectors are the
// reference's own known-answer tests (test/test-ntt.cpp:96-115, :357-404;
// test/test-eltwise-*.cpp; test/test-number-theory.cpp), so each block reads like
// the reference test it mirrors.  Needs a GPU: every compute call runs the HIP
// kernels through the C-ABI (there is no CPU fallback).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "hexl/hexl.hpp"

using namespace intel::hexl;

static int g_fail = 0;
#define EXPECT(cond)                                                  \
  do {                                                                \
    if (!(cond)) {                                                    \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
      ++g_fail;                                                       \
    }                                                                 \
  } while (0)
#define EXPECT_THROW(stmt)                                            \
  do {                                                                \
    bool threw = false;                                               \
    try {                                                             \
      stmt;                                                           \
    } catch (const std::exception&) {                                 \
      threw = true;                                                   \
    }                                                                 \
    if (!threw) {                                                     \
      std::printf("FAIL %s:%d: no throw: %s\n", __FILE__, __LINE__, #stmt); \
      ++g_fail;                                                       \
    }                                                                 \
  } while (0)

typedef std::vector<uint64_t> V;

struct NttCase {
  uint64_t n, q;
  V in, out;
};

static void test_ntt_api() {  // TEST_P(DegreeModulusInputOutput, API)
  const uint64_t M60 = 0xffffffffffc0001ULL;
  std::vector<NttCase> cases = {
      {2, 281474976710897ULL, {0, 0}, {0, 0}},
      {2, M60, {0, 0}, {0, 0}},
      {2, 281474976710897ULL, {1, 0}, {1, 1}},
      {2, 281474976710897ULL, {1, 1}, {19842761023586ULL, 261632215687313ULL}},
      {2, M60, {1, 1}, {288794978602139553ULL, 864126526004445282ULL}},
      {4, 113, {94, 109, 11, 18}, {82, 2, 81, 98}},
      {4, 281474976710897ULL, {281474976710765ULL, 49, 281474976710643ULL, 275},
       {12006376116355ULL, 216492038983166ULL, 272441922811203ULL, 62009615510542ULL}},
      {4, 113, {59, 50, 98, 50}, {1, 2, 3, 4}},
      {4, 73, {2, 1, 1, 1}, {17, 41, 36, 60}},
      {4, 16417, {31, 21, 15, 34}, {1611, 14407, 14082, 2858}},
      {4, 4194353, {4127, 9647, 1987, 5410}, {1478161, 3359347, 222964, 3344742}},
      {8, 4194353, {1, 0, 0, 0, 0, 0, 0, 0}, {1, 1, 1, 1, 1, 1, 1, 1}},
      {8, 4194353, {1, 1, 0, 0, 0, 0, 0, 0},
       {132171, 4062184, 2675172, 1519183, 462763, 3731592, 1824324, 2370031}},
      {32, 769,
       {401, 203, 221, 352, 487, 151, 405, 356, 343, 424, 635, 757, 457, 280, 624, 353,
        496, 353, 624, 280, 457, 757, 635, 424, 343, 356, 405, 151, 487, 352, 221, 203},
       {1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16,
        17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32}}};
  for (auto& c : cases) {
    NTT ntt(c.n, c.q);
    V x = c.in;
    ntt.ComputeForward(x.data(), x.data(), 1, 1);  // in place
    EXPECT(x == c.out);
    x = c.in;
    ntt.ComputeForward(x.data(), x.data(), 2, 4);  // lazy: compare mod q
    for (auto& e : x) {
      EXPECT(e < 4 * c.q);
      e %= c.q;
    }
    EXPECT(x == c.out);
    V out(c.n, 99), back(c.n, 99);
    ntt.ComputeForward(out.data(), c.in.data(), 1, 1);  // out of place round trip
    EXPECT(out == c.out);
    ntt.ComputeInverse(back.data(), out.data(), 1, 1);
    EXPECT(back == c.in);
    ntt.ComputeInverse(back.data(), out.data(), 1, 2);  // lazy inverse
    for (auto& e : back) {
      EXPECT(e < 2 * c.q);
      e %= c.q;
    }
    EXPECT(back == c.in);
    EXPECT(ntt.GetDegree() == c.n && ntt.GetModulus() == c.q);
    EXPECT(ntt.GetMinimalRootOfUnity() == MinimalPrimitiveRoot(2 * c.n, c.q));
  }
}

static void test_ntt_powers_and_roots() {  // TEST(NTT, Powers), root_of_unity
  const uint64_t modulus = 0xffffffffffc0001ULL;
  {
    NTT ntt(2, modulus);
    EXPECT(ntt.GetRootOfUnityPower(0) == 1ULL);
    EXPECT(ntt.GetRootOfUnityPower(1) == 288794978602139552ULL);
  }
  {
    NTT ntt(4, modulus);
    EXPECT(ntt.GetRootOfUnityPower(2) == 178930308976060547ULL);
    EXPECT(ntt.GetRootOfUnityPower(3) == 748001537669050592ULL);
  }
  uint64_t N = 8, q = 769;
  V a{1, 2, 3, 4, 5, 6, 7, 8}, b = a;
  NTT ntt1(N, q), ntt2(N, q, MinimalPrimitiveRoot(2 * N, q));
  ntt1.ComputeForward(a.data(), a.data(), 1, 1);
  ntt2.ComputeForward(b.data(), b.data(), 1, 1);
  EXPECT(a == b);
  EXPECT(ntt1.GetInvRootOfUnityPower(0) == ntt1.GetInvRootOfUnityPowers()[0]);
  // table identities: R * IR pairs, precon = floor(W 2^64 / q)
  const auto& R = ntt1.GetRootOfUnityPowers();
  const auto& P = ntt1.GetPrecon64RootOfUnityPowers();
  for (size_t i = 0; i < N; ++i) EXPECT(P[i] == MultiplyFactor(R[i], 64, q).BarrettFactor());
  EXPECT(ntt1.GetAVX512RootOfUnityPowers().size() == N + N / 4 + 3 * (N / 8));
  NTT copy = ntt1;  // copyable, shares state
  V c{1, 2, 3, 4, 5, 6, 7, 8};
  copy.ComputeForward(c.data(), c.data(), 1, 1);
  EXPECT(c == a);
  NTT empty;
  EXPECT_THROW(empty.GetDegree());
  EXPECT_THROW(NTT(8, 770));
  EXPECT(NTT::CheckArguments(8, 769) && !NTT::CheckArguments(8, 771));
}

struct CountingAllocator {  // test/test-ntt.cpp:117-200: a custom allocator is used
  static size_t allocations;
  void* allocate(size_t n) {
    ++allocations;
    return std::malloc(n);
  }
  void deallocate(void* p, size_t) { std::free(p); }
};
size_t CountingAllocator::allocations = 0;

static void test_ntt_allocator() {
  CountingAllocator a;
  {
    NTT ntt(1024, 0xffffee001ULL, std::move(a));
    EXPECT(CountingAllocator::allocations >= 4);
    V x(1024, 1), y(1024);
    ntt.ComputeForward(y.data(), x.data(), 1, 1);
    ntt.ComputeInverse(y.data(), y.data(), 1, 1);
    EXPECT(x == y);
  }
}

static void test_ntt_threads_and_sizes() {
  // one shared NTT used from several threads (README "thread-safe"), N = 2^16
  const uint64_t N = 65536;
  const uint64_t q = GeneratePrimes(1, 54, true, N)[0];
  EXPECT(q == 18014398510661633ULL);
  NTT ntt(N, q);
  std::vector<std::thread> ts;
  std::vector<int> ok(4, 0);
  for (int t = 0; t < 4; ++t)
    ts.emplace_back([&, t] {
      V x(N), y(N);
      uint64_t s = 12345 + t;
      for (auto& e : x) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        e = (s >> 8) % q;
      }
      for (int it = 0; it < 3; ++it) {
        ntt.ComputeForward(y.data(), x.data(), 1, 1);
        ntt.ComputeInverse(y.data(), y.data(), 1, 1);
      }
      ok[t] = (x == y);
    });
  for (auto& t : ts) t.join();
  for (int v : ok) EXPECT(v == 1);
}

static void test_device_mapped_memory() {
  // Extension: buffers in pinned, device-mapped host memory -- same calls, same results, the
  // kernels run straight on them (include/hexl/util/device-mapped-allocator.hpp).
  for (uint64_t N : {1024ull, 4096ull, 16384ull, 65536ull}) {
    const uint64_t q = GeneratePrimes(1, 54, true, N)[0];
    NTT ntt(N, q);
    AlignedVector64<uint64_t> x = DeviceMappedVector(N), y = DeviceMappedVector(N);
    V hx(N), hy(N);
    uint64_t s = 99 + N;
    for (uint64_t i = 0; i < N; ++i) {
      s = s * 6364136223846793005ULL + 1442695040888963407ULL;
      x[i] = hx[i] = (s >> 8) % q;
    }
    ntt.ComputeForward(y.data(), x.data(), 1, 1);    // mapped -> mapped
    ntt.ComputeForward(hy.data(), hx.data(), 1, 1);  // ordinary host memory (staged)
    EXPECT(std::equal(hy.begin(), hy.end(), y.begin()));
    ntt.ComputeForward(hy.data(), x.data(), 4, 4);   // mapped operand, ordinary result
    for (uint64_t i = 0; i < N; ++i)
      if (hy[i] >= 4 * q || hy[i] % q != y[i]) {
        EXPECT(!"lazy forward from mapped memory");
        break;
      }
    ntt.ComputeInverse(y.data(), y.data(), 1, 1);  // in place on mapped memory
    EXPECT(std::equal(x.begin(), x.end(), y.begin()));
    EltwiseMultMod(y.data(), x.data(), x.data(), N, q, 1);
    EltwiseMultMod(hy.data(), hx.data(), hx.data(), N, q, 1);
    EXPECT(std::equal(hy.begin(), hy.end(), y.begin()));
    EltwiseFMAMod(y.data(), x.data(), 3, nullptr, N, q, 1);
    EltwiseFMAMod(hy.data(), hx.data(), 3, nullptr, N, q, 1);
    EXPECT(std::equal(hy.begin(), hy.end(), y.begin()));
  }
  {  // an existing allocation registered once (a caller's memory pool)
    const uint64_t N = 8192, q = GeneratePrimes(1, 50, true, N)[0];
    NTT ntt(N, q);
    V pool(3 * N), ref(N);
    for (uint64_t i = 0; i < N; ++i) pool[i] = ref[i] = (i * 2654435761ULL) % q;
    RegisterHostMemory(pool.data(), pool.size() * sizeof(uint64_t));
    ntt.ComputeForward(pool.data() + N, pool.data(), 1, 1);
    ntt.ComputeInverse(pool.data() + 2 * N, pool.data() + N, 1, 1);
    UnregisterHostMemory(pool.data());
    EXPECT(std::equal(ref.begin(), ref.end(), pool.begin() + 2 * N));
    V fwd(N);
    ntt.ComputeForward(fwd.data(), ref.data(), 1, 1);
    EXPECT(std::equal(fwd.begin(), fwd.end(), pool.begin() + N));
  }
}

static void test_ntt_map_extension() {
  // Extension: polynomials of several moduli in one call, SEAL's interleaved layout
  // [component][modulus][N] (key-switch-internal.cpp:60-90), host buffers.
  const uint64_t N = 4096, K = 3, comps = 4;
  std::vector<uint64_t> primes = GeneratePrimes(K, 54, true, N);
  std::vector<NTT> ntts;
  for (uint64_t q : primes) ntts.emplace_back(N, q);
  std::vector<const NTT*> ptrs;
  for (auto& t : ntts) ptrs.push_back(&t);
  V x(comps * K * N), want(comps * K * N), got(comps * K * N);
  uint64_t s = 7;
  for (uint64_t i = 0; i < comps * K; ++i)
    for (uint64_t j = 0; j < N; ++j) {
      s = s * 6364136223846793005ULL + 1442695040888963407ULL;
      x[i * N + j] = (s >> 8) % primes[i % K];
    }
  for (uint64_t i = 0; i < comps * K; ++i)
    ntts[i % K].ComputeForward(want.data() + i * N, x.data() + i * N, 1, 1);
  const uint8_t tab[3] = {0, 1, 2};
  NTT::ComputeForwardMap(ptrs.data(), K, tab, K, 1, got.data(), x.data(), comps * K, 1, 1);
  EXPECT(got == want);
  std::vector<uint32_t> idx(comps * K);
  for (uint64_t i = 0; i < comps * K; ++i) idx[i] = (uint32_t)(i % K);
  NTT::ComputeInverseIndexed(ptrs.data(), K, idx.data(), got.data(), got.data(), comps * K, 1, 1);
  EXPECT(got == x);
  EXPECT_THROW(NTT::ComputeForwardMap(ptrs.data(), K, nullptr, K, 1, got.data(), x.data(), 1, 1, 1));
  // the same on device buffers (DeviceMalloc / Copy: no HIP toolchain needed): one launch
  // sequence over the interleaved layout, then a batch call and an element-wise op in place
  const size_t bytes = x.size() * sizeof(uint64_t);
  uint64_t* d = static_cast<uint64_t*>(DeviceMalloc(bytes));
  Copy(d, x.data(), bytes);
  NTT::ComputeForwardMap(ptrs.data(), K, tab, K, 1, d, d, comps * K, 1, 1);
  V back(x.size());
  Copy(back.data(), d, bytes);
  EXPECT(back == want);
  NTT::ComputeInverseIndexed(ptrs.data(), K, idx.data(), d, d, comps * K, 1, 1);
  Copy(back.data(), d, bytes);
  EXPECT(back == x);
  EltwiseAddMod(d, d, d, N, primes[0]);  // first polynomial doubled mod its prime, on the device
  Copy(back.data(), d, N * sizeof(uint64_t));
  for (uint64_t j = 0; j < N; ++j)
    if (back[j] != (2 * x[j]) % primes[0]) {
      EXPECT(!"EltwiseAddMod on device memory");
      break;
    }
  DeviceFree(d);
}

static void test_eltwise() {
  {  // TEST(EltwiseMultMod, 4 / 6 / 8big3), in place and out of place
    V a{2, 4, 3, 2}, b{2, 1, 2, 0}, r(4);
    EltwiseMultMod(r.data(), a.data(), b.data(), 4, 769, 1);
    EXPECT((r == V{4, 4, 6, 0}));
    EltwiseMultMod(a.data(), a.data(), b.data(), 4, 769, 1);
    EXPECT((a == V{4, 4, 6, 0}));
    V c{1078888294739028ULL, 1, 1, 1, 1, 1, 1, 1}, d{1114802337613200ULL, 1, 1, 1, 1, 1, 1, 1};
    V e(8);
    EltwiseMultMod(e.data(), c.data(), d.data(), 8, 1125891450734593ULL, 1);
    EXPECT((e == V{13344071208410ULL, 1, 1, 1, 1, 1, 1, 1}));
  }
  {  // TEST(EltwiseFMAMod, small / native_null / mult_input_mod_factor)
    V a{1, 2, 3, 4, 5, 6, 7, 8}, c{9, 10, 11, 12, 13, 14, 15, 16};
    EltwiseFMAMod(a.data(), a.data(), 1, c.data(), 8, 769, 1);
    EXPECT((a == V{10, 12, 14, 16, 18, 20, 22, 24}));
    V b{1, 2, 3, 4, 5, 6, 7, 8, 9};
    EltwiseFMAMod(b.data(), b.data(), 1, nullptr, 9, 769, 1);
    EXPECT((b == V{1, 2, 3, 4, 5, 6, 7, 8, 9}));
    for (uint64_t f = 1; f <= 8; f *= 2) {
      V x(17), add(17);
      for (int i = 0; i < 17; ++i) {
        x[i] = (f - 1) * 101 + i + 1;
        add[i] = 17 + i;
      }
      EltwiseFMAMod(x.data(), x.data(), 72, add.data(), 17, 101, f);
      EXPECT((x == V{89, 61, 33, 5, 78, 50, 22, 95, 67, 39, 11, 84, 56, 28, 0, 73, 45}));
    }
  }
  {  // TEST(EltwiseReduceMod, 2_2 / 4_1 / 0_1 / 2_1 / 4_2)
    V op{0, 450, 735, 900, 1350, 1459}, r(6);
    EltwiseReduceMod(r.data(), op.data(), 6, 750, 2, 2);
    EXPECT(r == op);
    EltwiseReduceMod(r.data(), op.data(), 6, 730, 2, 1);
    EXPECT((r == V{0, 450, 5, 170, 620, 729}));
    V op4{2, 4, 1600, 2500}, r4(4);
    EltwiseReduceMod(r4.data(), op4.data(), 4, 750, 4, 1);
    EXPECT((r4 == V{2, 4, 100, 250}));
    EltwiseReduceMod(r4.data(), op4.data(), 4, 750, 750, 1);
    EXPECT((r4 == V{2, 4, 100, 250}));
    V op42{1, 730, 1000, 1460, 2100, 2919};
    EltwiseReduceMod(r.data(), op42.data(), 6, 730, 4, 2);
    EXPECT((r == V{1, 730, 1000, 0, 640, 1459}));
  }
  {  // TEST(EltwiseAddMod / EltwiseSubMod, vector_vector / vector_scalar)
    V a{1, 2, 3, 4, 5, 6, 7, 8}, b{1, 3, 5, 7, 9, 4, 4, 6}, r(8);
    EltwiseAddMod(r.data(), a.data(), b.data(), 8, 10);
    EXPECT((r == V{2, 5, 8, 1, 4, 0, 1, 4}));
    EltwiseAddMod(r.data(), a.data(), 3, 8, 10);
    EXPECT((r == V{4, 5, 6, 7, 8, 9, 0, 1}));
    EltwiseSubMod(r.data(), a.data(), b.data(), 8, 10);
    EXPECT((r == V{0, 9, 8, 7, 6, 2, 3, 2}));
    EltwiseSubMod(r.data(), a.data(), 3, 8, 10);
    EXPECT((r == V{8, 9, 0, 1, 2, 3, 4, 5}));
  }
  {  // ExampleEltwiseCmpAdd / ExampleEltwiseCmpSubMod (example/example.cpp:56-85) and
     // TEST_P(EltwiseCmpAddTest / EltwiseCmpSubModTest, Native) NE rows, in place
    V op1{1, 2, 3, 4, 5, 6, 7, 8};
    EltwiseCmpAdd(op1.data(), op1.data(), op1.size(), CMPINT::NLE, 3, 5);
    EXPECT((op1 == V{1, 2, 3, 9, 10, 11, 12, 13}));
    V op2{1, 2, 3, 4, 5, 6, 7};
    EltwiseCmpSubMod(op2.data(), op2.data(), op2.size(), 10, CMPINT::NLE, 4, 5);
    EXPECT((op2 == V{1, 2, 3, 4, 0, 1, 2}));
    V op3{1, 2, 3, 4, 5, 6, 7}, r(7);
    EltwiseCmpAdd(r.data(), op3.data(), 7, CMPINT::NE, 4, 5);
    EXPECT((r == V{6, 7, 8, 4, 10, 11, 12}));
    EltwiseCmpSubMod(r.data(), op3.data(), 7, 10, CMPINT::NE, 4, 5);
    EXPECT((r == V{6, 7, 8, 4, 0, 1, 2}));
    EXPECT(Not(CMPINT::LT) == CMPINT::NLT && Not(CMPINT::TRUE) == CMPINT::FALSE);
  }
  {  // TEST(DyadicMultiply, small_two_mod / small_one_mod_inplace)
    V moduli{10, 20};
    V x{1, 2, 3, 11, 12, 13, 4, 5, 6, 14, 15, 16}, y{2, 4, 6, 12, 14, 16, 8, 1, 3, 18, 11, 13};
    V out(18, 0);
    DyadicMultiply(out.data(), x.data(), y.data(), 3, moduli.data(), 2);
    EXPECT((out == V{2, 8, 8, 12, 8, 8, 6, 2, 5, 6, 2, 5, 2, 5, 8, 12, 5, 8}));
    V one{10}, a{1, 2, 3, 4, 5, 6, 0, 0, 0}, b{2, 4, 6, 8, 1, 3};
    DyadicMultiply(a.data(), a.data(), b.data(), 3, one.data(), 1);
    EXPECT((a == V{2, 8, 8, 6, 2, 5, 2, 5, 8}));
  }
  V z(4);
  EXPECT_THROW(EltwiseCmpAdd(z.data(), z.data(), 4, CMPINT::EQ, 1, 0));
  EXPECT_THROW(EltwiseCmpSubMod(z.data(), z.data(), 4, 10, CMPINT::EQ, 1, 0));
  EXPECT_THROW(EltwiseMultMod(z.data(), z.data(), z.data(), 4, 769, 3));
  EXPECT_THROW(EltwiseFMAMod(z.data(), z.data(), 1, nullptr, 4, 1ULL << 61, 1));
}

static void test_number_theory() {  // test/test-number-theory.cpp
  EXPECT(MultiplyMod(7, 7, 10) == 9);
  EXPECT(MultiplyMod(1152921504605798400ULL, 1152921504605798401ULL, 2305843009211596801ULL) ==
         576460752302899200ULL);
  MultiplyFactor mf(1152921504605798401ULL, 64, 2305843009211596801ULL);
  EXPECT(MultiplyMod(1152921504605798401ULL, 1152921504605798401ULL, mf.BarrettFactor(),
                     2305843009211596801ULL) == 1729382256908697601ULL);
  EXPECT(MultiplyModLazy<64>(2305843009211596800ULL, 2305843009211596800ULL,
                             2305843009211596801ULL) == 2305843009211596802ULL);
  EXPECT(PowMod(2424242424ULL, 16, 131313131313ULL) == 39418477653ULL);
  EXPECT(IsPrimitiveRoot(960907033ULL, 8, 1234565441ULL));
  EXPECT(!IsPrimitiveRoot(1180581915ULL, 32, 1234565441ULL));
  EXPECT(MinimalPrimitiveRoot(8, 1234565441ULL) == 249725733ULL);
  EXPECT(InverseMod(5, 19) == 4 && InverseMod(3, 2) == 1);
  EXPECT(ReverseBits(0xFFFF0000FFFF0000ULL, 64) == 0x0000FFFF0000FFFFULL);
  EXPECT(IsPrime(0xffffee001ULL) && !IsPrime(72307ULL * 59399ULL));
  EXPECT(AddUIntMod(5, 6, 10) == 1 && SubUIntMod(3, 7, 10) == 6);
  EXPECT(DivideUInt128UInt64Lo(4294908658ULL, 0xffffffffffffffffULL, 0xffffffffffffffffULL) ==
         4294908659ULL);
  EXPECT(MSB(2305843009213689601ULL) == 60 && MSB(1) == 0);
  EXPECT(Log2(1025) == 10 && IsPowerOfFour(4096) && !IsPowerOfFour(2048));
  EXPECT(MontgomeryReduce<64>(136630700ULL, 6847304339915631516ULL, 67280421310725ULL, 46,
                              70368744177663ULL, 62463730494515ULL) == 1546598034044ULL);
  EXPECT(MontgomeryReduce<52>(559639348720ULL, 1832906312477596ULL, 67280421310725ULL, 46,
                              70368744177663ULL, 62463730494515ULL) == 1546598034044ULL);
  EXPECT(HenselLemma2adicRoot(3, 5) == 3 && HenselLemma2adicRoot(46, 67280421310725ULL) ==
                                                62463730494515ULL);
  uint64_t two = 2 * 769, r = ReduceMod<4>(3 * 769 + 5, 769, &two);
  EXPECT(r == 5);
  EXPECT(BarrettReduce64<1>(12345678, 769, MultiplyFactor(1, 64, 769).BarrettFactor()) ==
         12345678 % 769);
  auto primes = GeneratePrimes(10, 50, false, 4096);
  EXPECT(primes.size() == 10);
  for (auto p : primes) EXPECT(p % 8192 == 1 && IsPrime(p));
  EXPECT(Not(CMPINT::LT) == CMPINT::NLT && Not(CMPINT::TRUE) == CMPINT::FALSE);
}

// TEST(KeySwitch, small) through host buffers; data from tests/golden/hexl_kat.json
// (written next to this file by tests/test_cpp_shim.py as key_switch_kat.txt:
// n D K R C, moduli, modswitch factors, keys, input, target, expected).
static void test_key_switch(const char* path) {
  FILE* f = fopen(path, "r");
  if (!f) {
    printf("key_switch KAT file missing, skipped\n");
    return;
  }
  unsigned long long n, D, K, R, C;
  EXPECT(fscanf(f, "%llu %llu %llu %llu %llu", &n, &D, &K, &R, &C) == 5);
  auto read = [&](size_t count) {
    V v(count);
    for (auto& x : v) {
      unsigned long long t;
      EXPECT(fscanf(f, "%llu", &t) == 1);
      x = t;
    }
    return v;
  };
  V moduli = read(K), msf = read(D);
  std::vector<V> keys;
  for (unsigned long long j = 0; j < D; ++j) keys.push_back(read(C * K * n));
  V input = read(C * D * n), target = read(D * n), expected = read(C * D * n);
  fclose(f);
  std::vector<const uint64_t*> kp;
  for (auto& k : keys) kp.push_back(k.data());
  KeySwitch(input.data(), target.data(), n, D, K, R, C, moduli.data(), kp.data(), msf.data());
  EXPECT(input == expected);
  EXPECT_THROW(KeySwitch(input.data(), target.data(), n, D, K, R, C, moduli.data(), kp.data(),
                         msf.data(), moduli.data()));
}

int main(int argc, char** argv) {
  if (argc > 1) test_key_switch(argv[1]);
  test_number_theory();
  test_ntt_api();
  test_ntt_powers_and_roots();
  test_ntt_allocator();
  test_ntt_threads_and_sizes();
  test_device_mapped_memory();
  test_ntt_map_extension();
  test_eltwise();
  if (g_fail) {
    std::printf("%d checks failed\n", g_fail);
    return 1;
  }
  std::printf("all C++ shim checks passed\n");
  return 0;
}
