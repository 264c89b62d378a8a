// hexl/experimental/seal/key-switch-internal.hpp -- intel::hexl::internal::KeySwitch.
// Drop-in for hexl/include/hexl/experimental/seal/key-switch-internal.hpp:12-17; same contract as
// the public KeySwitch (key-switch.hpp) -- both reach hexl_amd_key_switch / _host.
#pragma once

#include <stdint.h>

namespace intel {
namespace hexl {
namespace internal {

void KeySwitch(uint64_t* result, const uint64_t* t_target_iter_ptr, uint64_t n,
               uint64_t decomp_modulus_size, uint64_t key_modulus_size,
               uint64_t rns_modulus_size, uint64_t key_component_count, const uint64_t* moduli,
               const uint64_t** k_switch_keys, const uint64_t* modswitch_factors,
               const uint64_t* root_of_unity_powers_ptr = nullptr);

}  // namespace internal
}  // namespace hexl
}  // namespace intel
