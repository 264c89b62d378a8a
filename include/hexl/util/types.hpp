// hexl/util/types.hpp -- 128-bit integer aliases used by the scalar helpers.
#pragma once
#include <stdint.h>

#include "hexl/util/defines.hpp"

__extension__ typedef __int128 int128_t;
__extension__ typedef unsigned __int128 uint128_t;
