// hexl/hexl.hpp -- umbrella header of the MI355X-native HEXL hot path
// (reference: hexl/include/hexl/hexl.hpp:6-26): everything the reference's umbrella exposes except
// its FFT-like and LR mat-vec experiments (outside this build's scope).
// hexl/experimental/seal/ntt-cache.hpp (GetNTT) and locks.hpp are installed too; like the
// reference's umbrella this header does not pull them in.
#pragma once

#include "hexl/eltwise/eltwise-add-mod.hpp"
#include "hexl/eltwise/eltwise-cmp-add.hpp"
#include "hexl/eltwise/eltwise-cmp-sub-mod.hpp"
#include "hexl/eltwise/eltwise-fma-mod.hpp"
#include "hexl/eltwise/eltwise-mult-mod.hpp"
#include "hexl/eltwise/eltwise-reduce-mod.hpp"
#include "hexl/eltwise/eltwise-sub-mod.hpp"
#include "hexl/experimental/seal/dyadic-multiply-internal.hpp"
#include "hexl/experimental/seal/dyadic-multiply.hpp"
#include "hexl/experimental/seal/key-switch-internal.hpp"
#include "hexl/experimental/seal/key-switch.hpp"
#include "hexl/logging/logging.hpp"
#include "hexl/ntt/ntt.hpp"
#include "hexl/number-theory/number-theory.hpp"
#include "hexl/util/aligned-allocator.hpp"
#include "hexl/util/allocator.hpp"
#include "hexl/util/check.hpp"
#include "hexl/util/compiler.hpp"
#include "hexl/util/defines.hpp"
#include "hexl/util/device-mapped-allocator.hpp"
#include "hexl/util/types.hpp"
#include "hexl/util/util.hpp"
