/*
 * hexl_oracle.h -- declarations for the CPU oracle (TEST INFRASTRUCTURE ONLY;
 * see hexl_oracle.c).  Not part of the product's include/ tree.
 */
#ifndef HEXL_ORACLE_H_
#define HEXL_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

uint64_t ho_mul_hi64(uint64_t x, uint64_t y);
uint64_t ho_reduce128(uint64_t hi, uint64_t lo, uint64_t modulus);
uint64_t ho_divide_u128_u64_lo(uint64_t x1, uint64_t x0, uint64_t y);
uint64_t ho_msb(uint64_t x);

uint64_t ho_multiply_factor(uint64_t operand, uint64_t bit_shift,
                            uint64_t modulus);
uint64_t ho_inverse_mod(uint64_t input, uint64_t modulus);
uint64_t ho_multiply_mod(uint64_t x, uint64_t y, uint64_t modulus);
uint64_t ho_multiply_mod_precon(uint64_t x, uint64_t y, uint64_t y_precon,
                                uint64_t modulus);
uint64_t ho_multiply_mod_lazy64(uint64_t x, uint64_t y_operand,
                                uint64_t y_barrett_factor, uint64_t modulus);
uint64_t ho_add_uint_mod(uint64_t x, uint64_t y, uint64_t modulus);
uint64_t ho_sub_uint_mod(uint64_t x, uint64_t y, uint64_t modulus);
uint64_t ho_pow_mod(uint64_t base, uint64_t exp, uint64_t modulus);
int ho_is_primitive_root(uint64_t root, uint64_t degree, uint64_t modulus);
uint64_t ho_minimal_primitive_root(uint64_t degree, uint64_t modulus);
uint64_t ho_reverse_bits(uint64_t x, uint64_t bit_width);
int ho_is_prime(uint64_t n);
size_t ho_generate_primes(uint64_t* out, size_t num_primes, size_t bit_size,
                          int prefer_small_primes, size_t ntt_size);

void ho_ntt_tables(uint64_t n, uint64_t q, uint64_t w, uint64_t* root_pows,
                   uint64_t* precon_root_pows, uint64_t* inv_root_pows,
                   uint64_t* precon_inv_root_pows);
void ho_ntt_forward_radix2(uint64_t* result, const uint64_t* operand,
                           uint64_t n, uint64_t q, const uint64_t* root_pows,
                           const uint64_t* precon_root_pows, uint64_t in_mf,
                           uint64_t out_mf);
void ho_ntt_inverse_radix2(uint64_t* result, const uint64_t* operand,
                           uint64_t n, uint64_t q,
                           const uint64_t* inv_root_pows,
                           const uint64_t* precon_inv_root_pows, uint64_t in_mf,
                           uint64_t out_mf);
/* hexl/ntt/ntt-radix-4.cpp:17-400, :402-700 -- the reference's radix-4 native transforms */
void ho_ntt_forward_radix4(uint64_t* result, const uint64_t* operand,
                           uint64_t n, uint64_t q, const uint64_t* root_pows,
                           const uint64_t* precon_root_pows, uint64_t in_mf,
                           uint64_t out_mf);
void ho_ntt_inverse_radix4(uint64_t* result, const uint64_t* operand,
                           uint64_t n, uint64_t q,
                           const uint64_t* inv_root_pows,
                           const uint64_t* precon_inv_root_pows, uint64_t in_mf,
                           uint64_t out_mf);
/* hexl_oracle_avx512.c: 8-lane variants of the two transforms above (cpu_baseline
 * of bench.py); call only when ho_has_avx512() != 0. */
int ho_has_avx512(void);
void ho_ntt_forward_radix2_avx512(uint64_t* result, const uint64_t* operand,
                                  uint64_t n, uint64_t q,
                                  const uint64_t* root_pows,
                                  const uint64_t* precon_root_pows,
                                  uint64_t in_mf, uint64_t out_mf);
void ho_ntt_inverse_radix2_avx512(uint64_t* result, const uint64_t* operand,
                                  uint64_t n, uint64_t q,
                                  const uint64_t* inv_root_pows,
                                  const uint64_t* precon_inv_root_pows,
                                  uint64_t in_mf, uint64_t out_mf);
void ho_ntt_forward_reference(uint64_t* operand, uint64_t n, uint64_t q,
                              const uint64_t* root_pows);
void ho_ntt_inverse_reference(uint64_t* operand, uint64_t n, uint64_t q,
                              const uint64_t* inv_root_pows);

void ho_eltwise_add_mod(uint64_t* result, const uint64_t* a, const uint64_t* b,
                        uint64_t n, uint64_t q);
void ho_eltwise_add_mod_scalar(uint64_t* result, const uint64_t* a, uint64_t b,
                               uint64_t n, uint64_t q);
void ho_eltwise_sub_mod(uint64_t* result, const uint64_t* a, const uint64_t* b,
                        uint64_t n, uint64_t q);
void ho_eltwise_sub_mod_scalar(uint64_t* result, const uint64_t* a, uint64_t b,
                               uint64_t n, uint64_t q);
void ho_eltwise_mult_mod(uint64_t* result, const uint64_t* a,
                         const uint64_t* b, uint64_t n, uint64_t q,
                         uint64_t in_mf);
void ho_eltwise_fma_mod(uint64_t* result, const uint64_t* arg1, uint64_t arg2,
                        const uint64_t* arg3, uint64_t n, uint64_t q,
                        uint64_t in_mf);
void ho_eltwise_cmp_add(uint64_t* result, const uint64_t* operand1, uint64_t n,
                        int cmp, uint64_t bound, uint64_t diff);
void ho_eltwise_cmp_sub_mod(uint64_t* result, const uint64_t* operand1,
                            uint64_t n, uint64_t modulus, int cmp,
                            uint64_t bound, uint64_t diff);
void ho_dyadic_multiply(uint64_t* result, const uint64_t* operand1,
                        const uint64_t* operand2, uint64_t n,
                        const uint64_t* moduli, uint64_t num_moduli);
void ho_eltwise_reduce_mod(uint64_t* result, const uint64_t* operand,
                           uint64_t n, uint64_t q, uint64_t in_mf,
                           uint64_t out_mf);

typedef struct ho_ntt {
  uint64_t n, q, w;
  uint64_t* root_pows;            /* bit-reversed powers of w               */
  uint64_t* precon_root_pows;     /* floor(W * 2^64 / q)                    */
  uint64_t* inv_root_pows;        /* stage-ordered inverse powers           */
  uint64_t* precon_inv_root_pows;
} ho_ntt;

ho_ntt* ho_ntt_create(uint64_t n, uint64_t q, uint64_t root /* 0 = minimal */);
void ho_ntt_destroy(ho_ntt* p);
void ho_ntt_forward(const ho_ntt* p, uint64_t* result, const uint64_t* operand,
                    uint64_t in_mf, uint64_t out_mf);
void ho_ntt_inverse(const ho_ntt* p, uint64_t* result, const uint64_t* operand,
                    uint64_t in_mf, uint64_t out_mf);
void ho_ntt_forward_batch(const ho_ntt* p, uint64_t* result,
                          const uint64_t* operand, uint64_t batch,
                          uint64_t in_mf, uint64_t out_mf);
void ho_ntt_inverse_batch(const ho_ntt* p, uint64_t* result,
                          const uint64_t* operand, uint64_t batch,
                          uint64_t in_mf, uint64_t out_mf);

void ho_key_switch(uint64_t* result, const uint64_t* t_target_iter, uint64_t n,
                   uint64_t decomp_modulus_size, uint64_t key_modulus_size,
                   uint64_t rns_modulus_size, uint64_t key_component_count,
                   const uint64_t* moduli, const uint64_t* const* k_switch_keys,
                   const uint64_t* modswitch_factors);
void ho_ntt_forward_batch_avx512(const ho_ntt* p, uint64_t* result,
                                 const uint64_t* operand, uint64_t batch,
                                 uint64_t in_mf, uint64_t out_mf);
void ho_ntt_inverse_batch_avx512(const ho_ntt* p, uint64_t* result,
                                 const uint64_t* operand, uint64_t batch,
                                 uint64_t in_mf, uint64_t out_mf);
void ho_fill_splitmix(uint64_t* out, uint64_t n, uint64_t seed,
                      uint64_t bound);

#ifdef __cplusplus
}
#endif
#endif  /* HEXL_ORACLE_H_ */
