// hexl/util/util.hpp -- CMPINT, the comparison selector of EltwiseCmpAdd /
// EltwiseCmpSubMod (hexl/include/hexl/util/util.hpp:16-51 in the reference;
// same enumerator values, they are part of the ABI of those two functions).
#pragma once
#include <cstddef>

namespace intel {
namespace hexl {

#undef TRUE
#undef FALSE

enum class CMPINT {
  EQ = 0,     ///< a == b
  LT = 1,     ///< a <  b
  LE = 2,     ///< a <= b
  FALSE = 3,  ///< never
  NE = 4,     ///< a != b
  NLT = 5,    ///< a >= b
  NLE = 6,    ///< a >  b
  TRUE = 7    ///< always
};

/// Logical negation of a comparison: the encoding is built so that flipping
/// bit 2 negates.
inline CMPINT Not(CMPINT cmp) { return static_cast<CMPINT>(static_cast<int>(cmp) ^ 4); }

}  // namespace hexl
}  // namespace intel
