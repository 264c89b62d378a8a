// hexl/util/defines.hpp -- build configuration of the MI355X-native HEXL.
// (The reference generates this file with CMake; here it is static: the host
// side is always built with a GNU-compatible compiler, and the compute path is
// the HIP library behind include/hexl_amd.h.)
#pragma once

#if defined(__clang__)
#define HEXL_USE_CLANG
#else
#define HEXL_USE_GNU
#endif

#define HEXL_AMD_GPU 1

// Silences unused-variable warnings.
#define HEXL_UNUSED(x) (void)(x)
