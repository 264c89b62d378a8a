// test_shim_debug.cpp -- the debug contract of the drop-in API as a TABLE: compiled with
// -DHEXL_DEBUG and linked against libhexl_debug.so (the shim built with HEXL_DEBUG, like the
// reference's hexl_debug library), a call throws exactly when it breaks the contract the
// reference checks in debug builds:
//   NTT::ComputeForward   result, operand non-null; input_mod_factor in {1, 2, 4};
//                         output_mod_factor in {1, 4}; every element < input_mod_factor * q
//                         (hexl/ntt/ntt-internal.cpp:193-198)
//   NTT::ComputeInverse   the same with input_mod_factor in {1, 2}, output_mod_factor in {1, 2}
//                         (hexl/ntt/ntt-internal.cpp:257-261)
//   Eltwise*              non-null operands, n != 0, modulus and mod-factor ranges, every element
//                         (and scalar) below its bound (hexl/eltwise/eltwise-*-mod.cpp, the
//                         HEXL_CHECK / HEXL_CHECK_BOUNDS lines at the top of each entry point)
// The NTT half enumerates {direction} x {operand fill} x {input factor} x {output factor} x
// {which pointer is null} x {kind of memory} and derives the expectation from the rule above
// (the reference pins the same rule with a hand-picked list, test/test-ntt.cpp:20-94); the
// Eltwise half is one row per way of breaking a call.  Needs a GPU.
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "hexl/hexl.hpp"

using namespace intel::hexl;

namespace {

int g_fail = 0, g_rows = 0;

void expect(bool want_throw, const std::string& what, const std::function<void()>& call) {
  bool threw = false;
  std::string msg;
  try {
    call();
  } catch (const std::exception& e) {
    threw = true;
    msg = e.what();
  } catch (...) {
    threw = true;
  }
  ++g_rows;
  if (threw != want_throw) {
    std::printf("FAIL %s: %s%s%s\n", what.c_str(), want_throw ? "did not throw" : "threw",
                msg.empty() ? "" : ": ", msg.c_str());
    ++g_fail;
  }
}

const uint64_t kN = 8, kQ = 769;

// ---------------------------------------------------------------- NTT
struct Fill {
  const char* name;
  uint64_t value;  // every element of the operand (0: the ramp 1..N, all below q)
};
const Fill kFills[] = {{"ramp", 0}, {"q-1", kQ - 1}, {"q", kQ}, {"2q-1", 2 * kQ - 1}, {"2q", 2 * kQ},
                       {"4q-1", 4 * kQ - 1}, {"4q", 4 * kQ}};
const uint64_t kFactors[] = {1, 2, 4, 3, 123};  // legal and illegal mod factors

bool legal_factors(bool forward, uint64_t in_mf, uint64_t out_mf) {
  if (forward) return (in_mf == 1 || in_mf == 2 || in_mf == 4) && (out_mf == 1 || out_mf == 4);
  return (in_mf == 1 || in_mf == 2) && (out_mf == 1 || out_mf == 2);
}

void ntt_contract(bool mapped) {
  NTT ntt(kN, kQ);
  for (int forward = 1; forward >= 0; --forward)
    for (const Fill& fill : kFills)
      for (uint64_t in_mf : kFactors)
        for (uint64_t out_mf : kFactors)
          for (int null_which = 0; null_which < 3; ++null_which) {  // 0 none, 1 result, 2 operand
            // (null pointers with one fill and one factor pair are enough)
            if (null_which && (fill.value != 0 || in_mf != 1 || out_mf != 1)) continue;
            AlignedVector64<uint64_t> data = mapped ? DeviceMappedVector(0) : AlignedVector64<uint64_t>();
            for (uint64_t i = 0; i < kN; ++i) data.push_back(fill.value ? fill.value : i + 1);
            const uint64_t largest = fill.value ? fill.value : kN;
            const bool breaks = null_which != 0 || !legal_factors(forward, in_mf, out_mf) ||
                                largest >= in_mf * kQ;
            uint64_t* result = null_which == 1 ? nullptr : data.data();
            const uint64_t* operand = null_which == 2 ? nullptr : data.data();
            const std::string what = std::string(mapped ? "mapped " : "host ") +
                                     (forward ? "ComputeForward" : "ComputeInverse") + " fill=" + fill.name +
                                     " in_mf=" + std::to_string(in_mf) + " out_mf=" + std::to_string(out_mf) +
                                     (null_which == 1 ? " result=null" : null_which == 2 ? " operand=null" : "");
            expect(breaks, what, [&] {
              if (forward)
                ntt.ComputeForward(result, operand, in_mf, out_mf);
              else
                ntt.ComputeInverse(result, operand, in_mf, out_mf);
            });
          }
}

// ---------------------------------------------------------------- Eltwise
typedef std::vector<uint64_t> V;

struct EltRow {
  const char* what;
  bool throws;
  std::function<void()> call;
};

void eltwise_contract() {
  static V ok1{1, 2, 3, 4, 5, 6, 7, 8}, ok2{8, 7, 6, 5, 4, 3, 2, 1};
  static V has_q{1, 2, 3, 4, 5, 6, 7, kQ};  // one element == q: legal only with a factor >= 2
  static V all_2q(8, 2 * kQ), out(8, 0);
  uint64_t* r = out.data();
  const uint64_t *a = ok1.data(), *b = ok2.data(), *big = has_q.data(), *two_q = all_2q.data();
  const EltRow rows[] = {
      // EltwiseMultMod (eltwise-mult-mod.cpp:18-33)
      {"MultMod ok", false, [=] { EltwiseMultMod(r, a, b, 8, kQ, 1); }},
      {"MultMod result null", true, [=] { EltwiseMultMod(nullptr, a, b, 8, kQ, 1); }},
      {"MultMod operand1 null", true, [=] { EltwiseMultMod(r, nullptr, b, 8, kQ, 1); }},
      {"MultMod operand2 null", true, [=] { EltwiseMultMod(r, a, nullptr, 8, kQ, 1); }},
      {"MultMod n == 0", true, [=] { EltwiseMultMod(r, a, b, 0, kQ, 1); }},
      {"MultMod modulus 1", true, [=] { EltwiseMultMod(r, a, b, 8, 1, 1); }},
      {"MultMod factor 3", true, [=] { EltwiseMultMod(r, a, b, 8, kQ, 3); }},
      {"MultMod operand1 >= q", true, [=] { EltwiseMultMod(r, big, b, 8, kQ, 1); }},
      {"MultMod operand2 >= q", true, [=] { EltwiseMultMod(r, a, big, 8, kQ, 1); }},
      {"MultMod operand1 < 2q, factor 2", false, [=] { EltwiseMultMod(r, big, b, 8, kQ, 2); }},
      {"MultMod operand 2q, factor 2", true, [=] { EltwiseMultMod(r, two_q, b, 8, kQ, 2); }},
      {"MultMod operand 2q, factor 4", false, [=] { EltwiseMultMod(r, two_q, b, 8, kQ, 4); }},
      // EltwiseAddMod / SubMod, vector and scalar forms (eltwise-add-mod.cpp:16-33, :65-80)
      {"AddMod ok", false, [=] { EltwiseAddMod(r, a, b, 8, kQ); }},
      {"AddMod operand1 >= q", true, [=] { EltwiseAddMod(r, big, b, 8, kQ); }},
      {"AddMod operand2 >= q", true, [=] { EltwiseAddMod(r, a, big, 8, kQ); }},
      {"AddMod scalar ok", false, [=] { EltwiseAddMod(r, a, uint64_t{3}, 8, kQ); }},
      {"AddMod scalar, operand1 >= q", true, [=] { EltwiseAddMod(r, big, uint64_t{3}, 8, kQ); }},
      {"AddMod scalar >= q", true, [=] { EltwiseAddMod(r, a, kQ, 8, kQ); }},
      {"SubMod ok", false, [=] { EltwiseSubMod(r, a, b, 8, kQ); }},
      {"SubMod operand1 >= q", true, [=] { EltwiseSubMod(r, big, b, 8, kQ); }},
      {"SubMod operand2 >= q", true, [=] { EltwiseSubMod(r, a, big, 8, kQ); }},
      {"SubMod scalar, operand1 >= q", true, [=] { EltwiseSubMod(r, big, uint64_t{3}, 8, kQ); }},
      {"SubMod scalar >= q", true, [=] { EltwiseSubMod(r, a, kQ, 8, kQ); }},
      // EltwiseFMAMod (eltwise-fma-mod.cpp:17-31)
      {"FMAMod ok, no addend", false, [=] { EltwiseFMAMod(r, a, 2, nullptr, 8, kQ, 1); }},
      {"FMAMod arg1 >= q", true, [=] { EltwiseFMAMod(r, big, 2, nullptr, 8, kQ, 1); }},
      {"FMAMod arg3 >= q", true, [=] { EltwiseFMAMod(r, a, 2, big, 8, kQ, 1); }},
      {"FMAMod arg2 >= q", true, [=] { EltwiseFMAMod(r, a, kQ, nullptr, 8, kQ, 1); }},
      {"FMAMod arg1 < 2q, factor 2", false, [=] { EltwiseFMAMod(r, big, 2, b, 8, kQ, 2); }},
      {"FMAMod factor 3", true, [=] { EltwiseFMAMod(r, a, 2, b, 8, kQ, 3); }},
      // EltwiseReduceMod (eltwise-reduce-mod.cpp:16-31; in == out with distinct buffers is a
      // plain copy, :94-99)
      {"ReduceMod 1 -> 1, element == q", true, [=] { EltwiseReduceMod(r, big, 8, kQ, 1, 1); }},
      {"ReduceMod 2 -> 2 copy", false, [=] { EltwiseReduceMod(r, big, 8, kQ, 2, 2); }},
      {"ReduceMod 2 -> 1, elements 2q", true, [=] { EltwiseReduceMod(r, two_q, 8, kQ, 2, 1); }},
      {"ReduceMod 4 -> 1, elements 2q", false, [=] { EltwiseReduceMod(r, two_q, 8, kQ, 4, 1); }},
      {"ReduceMod q -> 1, any words", false, [=] { EltwiseReduceMod(r, two_q, 8, kQ, kQ, 1); }},
      {"ReduceMod factor 3", true, [=] { EltwiseReduceMod(r, a, 8, kQ, 3, 1); }},
      {"ReduceMod output factor 4", true, [=] { EltwiseReduceMod(r, a, 8, kQ, 4, 4); }},
  };
  for (const EltRow& row : rows) expect(row.throws, row.what, row.call);
}

}  // namespace

int main() {
  ntt_contract(false);  // ordinary host buffers: the bound check is a host loop
  ntt_contract(true);   // device-mapped memory: the check runs as a kernel
  eltwise_contract();
  if (g_fail) {
    std::printf("%d of %d contract rows failed\n", g_fail, g_rows);
    return 1;
  }
  std::printf("all debug-contract checks passed (%d rows)\n", g_rows);
  return 0;
}
