// hexl/logging/logging.hpp -- HEXL_VLOG / START_EASYLOGGINGPP for callers that name them.
// Drop-in for hexl/include/hexl/logging/logging.hpp:13-43.  Release builds: both macros are the
// empty block the reference defines.  HEXL_DEBUG builds: the reference routes HEXL_VLOG through
// easylogging++ (a third-party dependency outside this build's scope); here the same statement
// streams to std::cerr when the verbosity -- set by START_EASYLOGGINGPP's "--v=N" argument, as
// with easylogging++, or by intel::hexl::logging::SetVerbosity -- is at least N.
#pragma once

#include <algorithm>
#include <vector>

#include "hexl/util/defines.hpp"

#ifdef HEXL_DEBUG

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>

namespace intel {
namespace hexl {
namespace logging {

inline std::atomic<int>& VerbosityRef() {
  static std::atomic<int> level{0};
  return level;
}
inline void SetVerbosity(int level) { VerbosityRef().store(level); }
inline int Verbosity() { return VerbosityRef().load(std::memory_order_relaxed); }
/// Reads "--v=N" / "-v" (easylogging++'s verbose flags) from a command line.
inline void ParseArgs(int argc, char** argv) {
  for (int i = 1; i < argc; ++i) {
    if (std::strncmp(argv[i], "--v=", 4) == 0) SetVerbosity(std::atoi(argv[i] + 4));
    if (std::strcmp(argv[i], "-v") == 0 || std::strcmp(argv[i], "--verbose") == 0) SetVerbosity(9);
  }
}

}  // namespace logging
}  // namespace hexl
}  // namespace intel

#define HEXL_VLOG(N, rest)                              \
  do {                                                  \
    if (::intel::hexl::logging::Verbosity() >= (N)) {   \
      std::ostringstream hexl_vlog_line_;               \
      hexl_vlog_line_ << rest << '\n';                  \
      std::cerr << hexl_vlog_line_.str();               \
    }                                                   \
  } while (0);

#define START_EASYLOGGINGPP(X, Y) ::intel::hexl::logging::ParseArgs((X), (Y))

#else

#define HEXL_VLOG(N, rest) \
  {}

#define START_EASYLOGGINGPP(X, Y) \
  {}

#endif
