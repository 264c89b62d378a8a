// hexl/util/check.hpp -- argument checks.  As in the reference
// (hexl/include/hexl/util/check.hpp:19-41) they throw std::runtime_error when
// HEXL_DEBUG is defined and expand to nothing otherwise.  Independently of this
// macro the GPU entry points always validate their scalar arguments (the check
// is O(1), not O(n)) and report violations by exception.
#pragma once
#include <stdint.h>

#include <sstream>
#include <stdexcept>

#include "hexl/util/types.hpp"

#ifdef HEXL_DEBUG
#define HEXL_CHECK(cond, expr)                                               \
  if (!(cond)) {                                                             \
    std::ostringstream hexl_check_msg;                                       \
    hexl_check_msg << expr << " in function: " << __FUNCTION__ << " in file: " \
                   << __FILE__ << ":" << __LINE__;                           \
    throw std::runtime_error(hexl_check_msg.str());                          \
  }
#define HEXL_CHECK_BOUNDS(arg, n, bound, expr)                               \
  for (size_t hexl_check_idx = 0; hexl_check_idx < n; ++hexl_check_idx) {    \
    HEXL_CHECK((arg)[hexl_check_idx] < bound, expr);                         \
  }
#else
#define HEXL_CHECK(cond, expr) \
  {}
#define HEXL_CHECK_BOUNDS(...) \
  {}
#endif
