// hexl/experimental/seal/key-switch.hpp -- CKKS key switching on the GPU.
// Drop-in for hexl/include/hexl/experimental/seal/key-switch.hpp:40-46.
#pragma once
#include <stdint.h>

namespace intel {
namespace hexl {

/// Key switching of one target polynomial (CKKS): for every RNS index the target is
/// brought to the key modulus, multiplied with the switching keys and accumulated in
/// 128 bits; the special prime is then divided out (rounded) and the outcome is added
/// to `result` (key_component_count x decomp_modulus_size x n words).
/// t_target_iter_ptr: decomp_modulus_size x n words in NTT form;
/// k_switch_keys[j]: key_component_count x key_modulus_size x n words.
/// root_of_unity_powers_ptr must be nullptr (as in the reference).
void KeySwitch(uint64_t* result, const uint64_t* t_target_iter_ptr, uint64_t n,
               uint64_t decomp_modulus_size, uint64_t key_modulus_size,
               uint64_t rns_modulus_size, uint64_t key_component_count, const uint64_t* moduli,
               const uint64_t** k_switch_keys, const uint64_t* modswitch_factors,
               const uint64_t* root_of_unity_powers_ptr = nullptr);

/// Additive extension (not in the reference): `num_targets` ciphertexts with the same keys
/// and moduli in one call -- targets and results back to back in the layouts above, all
/// DEVICE memory.  One sequence of at most eleven launches whatever the sizes (hexl_amd_key_switch_batch).
void KeySwitchBatch(uint64_t* result, const uint64_t* t_target_iter_ptr, uint64_t num_targets,
                    uint64_t n, uint64_t decomp_modulus_size, uint64_t key_modulus_size,
                    uint64_t rns_modulus_size, uint64_t key_component_count,
                    const uint64_t* moduli, const uint64_t** k_switch_keys,
                    const uint64_t* modswitch_factors);

}  // namespace hexl
}  // namespace intel
