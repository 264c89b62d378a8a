/*
 * hexl_amd.h -- C-ABI of the MI355X-native NTT / Eltwise hot path.
 *
 * This is the drop-in boundary: plain `extern "C"` entry points, raw pointers
 * and sizes, no C++ or torch types.  Every entry point names the reference
 * interface it replaces (paths relative to the intel/hexl tree).  The C++ shim
 * in include/hexl/ (namespace intel::hexl) is written on top of exactly these
 * functions; INTEGRATION.md shows the binding a HEXL maintainer would add.
 *
 * Conventions
 *   - A polynomial is N contiguous uint64_t; a batch is `batch` polynomials
 *     back to back (stride N).  Forward input / inverse output are in natural
 *     coefficient order, forward output / inverse input in bit-reversed order
 *     (hexl/include/hexl/ntt/ntt.hpp:92,102).
 *   - `result` may alias `operand` (in-place), as in the reference.
 *   - Functions without a `_host` suffix take DEVICE pointers and enqueue on
 *     `stream` (a hipStream_t passed as void*; NULL = the default stream) and
 *     return without synchronising.  `_host` variants take host pointers,
 *     stage through device memory, and return after the result is back on the
 *     host (this is what the single-call intel::hexl API needs).
 *   - Every function returns HEXL_AMD_OK (0) or an error code; the message is
 *     available from hexl_amd_last_error() (thread-local).  There is no CPU
 *     fallback: without a usable gfx950 device every compute entry point fails
 *     with HEXL_AMD_ERR_NO_DEVICE / HEXL_AMD_ERR_HIP.
 *   - Thread-safety: plans are immutable after creation and may be used from
 *     many threads concurrently (hexl README.md:264-265 "thread-safe").
 */
#ifndef HEXL_AMD_H_
#define HEXL_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HEXL_AMD_VERSION_MAJOR 0
#define HEXL_AMD_VERSION_MINOR 1

enum {
  HEXL_AMD_OK = 0,
  HEXL_AMD_ERR_INVALID_ARG = 1, /* argument outside the reference's contract */
  HEXL_AMD_ERR_HIP = 2,         /* a HIP runtime call failed                  */
  HEXL_AMD_ERR_NO_DEVICE = 3,   /* no gfx950 device visible                   */
  HEXL_AMD_ERR_ALLOC = 4
};

/* Message describing the last error on the calling thread ("" if none). */
const char* hexl_amd_last_error(void);

/* Number of visible HIP devices (0 and HEXL_AMD_ERR_NO_DEVICE if none). */
int hexl_amd_device_count(int* count);

/* 1 if `p` points to device (or managed) memory of a visible HIP device, 0 for
 * ordinary host memory.  Used by the intel::hexl shim to route a call either
 * straight to the kernels or through the *_host staging entry points. */
int hexl_amd_pointer_is_device(const void* p);

/* Host memory the kernels can address.  The *_host entry points (what the intel::hexl shim
 * calls for host pointers) stage ordinary host memory through a device buffer: H2D copy,
 * kernels, D2H copy.  Memory that is pinned AND mapped into the device's address space needs
 * none of that: a one-kernel transform (degree <= 2^13, 2^14 from 96 polynomials) or an
 * element-wise kernel runs straight on it over the link -- one launch, one synchronisation --
 * and a multi-pass transform reads its operand there.
 *   hexl_amd_host_alloc / _free        such memory from the runtime (hipHostMalloc, mapped)
 *   hexl_amd_host_register / _unregister  an existing allocation made such (hipHostRegister,
 *                                      mapped): one call over a caller's memory pool.  _unregister
 *                                      first waits for every device this process has used through
 *                                      the library, and the current one (nothing may still be using
 *                                      the mapping; devices the caller drives entirely by itself are
 *                                      the caller's to synchronise -- the call creates no context on
 *                                      a device the process never touched).  REGISTER MAPPINGS YOU OWN -- a page-aligned pool
 *                                      from mmap / an aligned allocator that you keep, or unmap
 *                                      after unregistering -- NOT sub-allocations of the malloc heap:
 *                                      on the HIP runtime torch 2.10 bundles (ROCm 7.0.2) a process
 *                                      that registers and unregisters heap arrays and later lets the
 *                                      runtime copy pageable memory from / to re-used heap pages
 *                                      (hipMemcpyAsync, torch.Tensor.to) dies of a GPU memory access
 *                                      fault; reproduced without this library in seconds by
 *                                      experiments/rocm_fault/register_then_pageable_copy_soak.py --register raw
 *                                      (EXPERIMENTS.md section 10).  The library's own copies of
 *                                      pageable memory go through pinned slots and are not affected
 *   hexl_amd_pointer_kind              0 ordinary host, 1 device / managed, 2 mapped host
 * include/hexl/util/device-mapped-allocator.hpp wraps the first pair as an
 * intel::hexl::AllocatorBase (allocator.hpp:12-51) for AlignedVector64 data buffers. */
int hexl_amd_host_alloc(void** p, uint64_t bytes);
int hexl_amd_host_free(void* p);
int hexl_amd_host_register(void* p, uint64_t bytes);
int hexl_amd_host_unregister(void* p);
int hexl_amd_pointer_kind(const void* p);

/* Device buffers for callers without a HIP toolchain of their own (a HEXL maintainer's GPU
 * branch, a cgo / JNI / ctypes host): the device-pointer entry points above are the throughput
 * path, and these four calls are all it takes to use them.
 *   hexl_amd_device_alloc   `bytes` of device memory on `device` (-1: the calling thread's current
 *                           device)
 *   hexl_amd_device_free
 *   hexl_amd_copy           dst <- src, `bytes`; each side may be host, mapped or device memory
 *                           (the direction is detected); enqueued on `stream` (NULL: the default
 *                           stream) and complete on return when `blocking` is non-zero.  Ordinary
 *                           (pageable) host memory goes through the thread's pinned slots: src
 *                           may be reused on return, and a copy INTO such memory is complete on
 *                           return whatever `blocking` says
 *   hexl_amd_synchronize    waits for everything enqueued on `stream` (NULL: the whole device).  On a
 *                           stream from hexl_amd_stream_create it polls a sequence number that a
 *                           one-thread kernel behind the work publishes in mapped host memory (tuning
 *                           key "host_poll"): 3-4 us earlier than hipStreamSynchronize returns -- one
 *                           N = 4096 transform on device memory, call + wait, 14.6 -> 10.6 us */
int hexl_amd_device_alloc(void** p, uint64_t bytes, int device);
int hexl_amd_device_free(void* p);
int hexl_amd_copy(void* dst, const void* src, uint64_t bytes, void* stream, int blocking);
int hexl_amd_synchronize(void* stream);

/* Devices and streams for callers that drive several GPUs from one process ("one
 * host thread + one stream per GPU"; the per-modulus loop of
 * hexl/experimental/seal/key-switch-internal.cpp:51-90 cut across devices).  Nothing here
 * requires the caller to track a "current device":
 *   - a plan runs on the device it was created for (hexl_amd_ntt_create(..., device)), whatever
 *     the calling thread's current device is; its `stream` argument must be NULL or a stream
 *     of that device;
 *   - every other stream-taking entry point (element-wise ops, DyadicMultiply, KeySwitch,
 *     hexl_amd_fill_splitmix, hexl_amd_copy) runs on the device that OWNS `stream`; with
 *     stream == NULL on the calling thread's current device;
 *   - hexl_amd_device_alloc takes the device explicitly.
 * hexl_amd_stream_create: a non-blocking stream on `device` (-1: the current device).
 * hexl_amd_stream_destroy: waits for the stream, frees the scratch the composite entry points
 * keyed by it, destroys it.  hexl_amd_set_device / _get_device: the calling thread's current
 * device, for callers that prefer the HIP model (one hipSetDevice per worker thread).
 * tests/cpp/multi_device.cpp is the reference caller: one std::thread + stream + plans per
 * device, BASELINE configs[3] sharded by the flat (prime, polynomial) index. */
int hexl_amd_set_device(int device);
int hexl_amd_get_device(int* device);
int hexl_amd_stream_create(void** stream, int device);
int hexl_amd_stream_destroy(void* stream);

/* Debug contract: the reference's debug builds (HEXL_DEBUG) check every ELEMENT of an
 * operand against its bound and throw (HEXL_CHECK_BOUNDS, hexl/include/hexl/util/check.hpp:32-35;
 * hexl/ntt/ntt-internal.cpp:198, :261; hexl/eltwise/eltwise-mult-mod.cpp:31-33 and the other
 * eltwise entry points; exercised by test/test-ntt.cpp:20-94).  *violations = number of words
 * of data[0, n) that are >= bound; data may be host, mapped or device memory (device data:
 * one reduction kernel, synchronous).  The debug flavour of the shim (libhexl_debug.so,
 * compiled with -DHEXL_DEBUG) calls it before every operation and throws. */
int hexl_amd_check_bounds(const uint64_t* data, uint64_t n, uint64_t bound,
                          uint64_t* violations);

/* ---------------------------------------------------------------------------
 * NTT plan == the state of one intel::hexl::NTT object
 * (hexl/include/hexl/ntt/ntt.hpp:22-293; ctors hexl/ntt/ntt-internal.cpp:24-52;
 * tables hexl/ntt/ntt-internal.cpp:54-169), resident on one device.
 * ------------------------------------------------------------------------- */
typedef struct hexl_amd_ntt hexl_amd_ntt;

/* Replaces NTT::NTT(degree, q[, root_of_unity]) (ntt-internal.cpp:24-52).
 * degree: power of two in [2, 2^20]; modulus: prime, == 1 mod 2*degree,
 * < 2^62 (NTT::CheckArguments, ntt-internal.cpp:171-186 -- enforced here in
 * all builds); root_of_unity: a primitive 2N-th root, or 0 for
 * MinimalPrimitiveRoot(2N, q).  device: HIP device ordinal, or -1 for the
 * calling thread's current device.  Builds the twiddle tables on the host,
 * uploads them once. */
int hexl_amd_ntt_create(hexl_amd_ntt** plan, uint64_t degree, uint64_t modulus,
                        uint64_t root_of_unity, int device);
int hexl_amd_ntt_destroy(hexl_amd_ntt* plan);

/* Getters (ntt.hpp:113-119). */
uint64_t hexl_amd_ntt_degree(const hexl_amd_ntt* plan);
uint64_t hexl_amd_ntt_modulus(const hexl_amd_ntt* plan);
uint64_t hexl_amd_ntt_root_of_unity(const hexl_amd_ntt* plan);
int hexl_amd_ntt_device(const hexl_amd_ntt* plan);

/* Host copies of the reference-layout tables (ntt.hpp:122-194).  `which`:
 *   0 GetRootOfUnityPowers          1 GetPrecon32RootOfUnityPowers
 *   2 GetPrecon64RootOfUnityPowers  3 GetInvRootOfUnityPowers
 *   4 GetPrecon32InvRootOfUnityPowers 5 GetPrecon52InvRootOfUnityPowers
 *   6 GetPrecon64InvRootOfUnityPowers
 * Returns a pointer to N uint64_t owned by the plan (NULL on bad `which`). */
const uint64_t* hexl_amd_ntt_table(const hexl_amd_ntt* plan, int which);

/* Replaces NTT::ComputeForward (ntt.hpp:99-100; ntt-internal.cpp:188-250 and
 * the kernels it dispatches to: ntt-radix-2.cpp:17-261, fwd-ntt-avx512.cpp)
 * for `batch` independent polynomials.  input_mod_factor in {1,2,4}: operand
 * in [0, f*q); output_mod_factor in {1,4}: result in [0, f*q) (canonical for
 * 1, any representative for 4). */
int hexl_amd_ntt_forward(const hexl_amd_ntt* plan, uint64_t* result,
                         const uint64_t* operand, uint64_t batch,
                         uint64_t input_mod_factor, uint64_t output_mod_factor,
                         void* stream);

/* Replaces NTT::ComputeInverse (ntt.hpp:109-110; ntt-internal.cpp:252-310;
 * ntt-radix-2.cpp:330-519, inv-ntt-avx512.cpp).  input_mod_factor in {1,2},
 * output_mod_factor in {1,2}. */
int hexl_amd_ntt_inverse(const hexl_amd_ntt* plan, uint64_t* result,
                         const uint64_t* operand, uint64_t batch,
                         uint64_t input_mod_factor, uint64_t output_mod_factor,
                         void* stream);

/* RNS form: `num_plans` plans (one per prime, same degree, same device);
 * polynomials [k*batch_per_plan, (k+1)*batch_per_plan) belong to plans[k].
 * This is the per-modulus loop of the in-tree callers
 * (hexl/experimental/seal/key-switch-internal.cpp:52-90) as one call. */
int hexl_amd_ntt_forward_rns(const hexl_amd_ntt* const* plans,
                             uint64_t num_plans, uint64_t* result,
                             const uint64_t* operand, uint64_t batch_per_plan,
                             uint64_t input_mod_factor,
                             uint64_t output_mod_factor, void* stream);
int hexl_amd_ntt_inverse_rns(const hexl_amd_ntt* const* plans,
                             uint64_t num_plans, uint64_t* result,
                             const uint64_t* operand, uint64_t batch_per_plan,
                             uint64_t input_mod_factor,
                             uint64_t output_mod_factor, void* stream);

/* Multi-prime form with a per-polynomial prime map ("plan*[] + per-poly prime
 * index"): `polys` polynomials back to back, polynomial i transformed with
 *     plans[ plan_of_slot[ (i / inner) % period ] ].
 * One call covers the layouts RNS callers use:
 *   - prime-major blocks (the _rns form): inner = polynomials per prime, period = primes,
 *     plan_of_slot = 0, 1, 2, ...;
 *   - SEAL's ciphertext layout [ciphertext][component][modulus][N], where modulus j of
 *     every component sits at polynomial index = ... * k + j
 *     (hexl/experimental/seal/key-switch-internal.cpp:60-90 indexes
 *     t_target[j * coeff_count] per modulus j inside one component): inner = 1,
 *     period = k, plan_of_slot = 0 .. k-1;
 *   - any periodic pattern (a modulus chain with dropped levels, key-switching moduli
 *     interleaved with the special prime, ...).
 * All plans: same degree, same device.  plan_of_slot is host memory, `period` entries,
 * each < num_plans.  For degrees 2^12 .. 2^17 (num_plans <= 40, period <= 1024) the whole
 * batch is ONE launch sequence whatever the number of moduli -- the workgroup's polynomial
 * selects its plan on the device; moduli of different arithmetic policies get one sequence
 * per policy.  Other shapes fall back to one call per run of polynomials that share a plan
 * (same results; the launch count then grows with the number of runs). */
int hexl_amd_ntt_forward_map(const hexl_amd_ntt* const* plans, uint64_t num_plans,
                             const uint8_t* plan_of_slot, uint64_t period, uint64_t inner,
                             uint64_t* result, const uint64_t* operand, uint64_t polys,
                             uint64_t input_mod_factor, uint64_t output_mod_factor,
                             void* stream);
int hexl_amd_ntt_inverse_map(const hexl_amd_ntt* const* plans, uint64_t num_plans,
                             const uint8_t* plan_of_slot, uint64_t period, uint64_t inner,
                             uint64_t* result, const uint64_t* operand, uint64_t polys,
                             uint64_t input_mod_factor, uint64_t output_mod_factor,
                             void* stream);

/* The same with an explicit prime index per polynomial: prime_index is host memory, `polys`
 * entries, each < num_plans.  The (inner, period) structure is recovered from the array
 * (shortest period over runs of equal length); an array without one is served run by run. */
int hexl_amd_ntt_forward_indexed(const hexl_amd_ntt* const* plans, uint64_t num_plans,
                                 const uint32_t* prime_index, uint64_t* result,
                                 const uint64_t* operand, uint64_t polys,
                                 uint64_t input_mod_factor, uint64_t output_mod_factor,
                                 void* stream);
int hexl_amd_ntt_inverse_indexed(const hexl_amd_ntt* const* plans, uint64_t num_plans,
                                 const uint32_t* prime_index, uint64_t* result,
                                 const uint64_t* operand, uint64_t polys,
                                 uint64_t input_mod_factor, uint64_t output_mod_factor,
                                 void* stream);

/* Host-pointer forms used by the intel::hexl::NTT shim: H2D, transform, D2H,
 * synchronous. */
int hexl_amd_ntt_forward_host(const hexl_amd_ntt* plan, uint64_t* result,
                              const uint64_t* operand, uint64_t batch,
                              uint64_t input_mod_factor,
                              uint64_t output_mod_factor);
int hexl_amd_ntt_inverse_host(const hexl_amd_ntt* plan, uint64_t* result,
                              const uint64_t* operand, uint64_t batch,
                              uint64_t input_mod_factor,
                              uint64_t output_mod_factor);

/* ---------------------------------------------------------------------------
 * Element-wise modular arithmetic on device vectors of n uint64_t.
 * ------------------------------------------------------------------------- */

/* EltwiseAddMod vector-vector / vector-scalar
 * (hexl/include/hexl/eltwise/eltwise-add-mod.hpp:22-37;
 * hexl/eltwise/eltwise-add-mod.cpp:16-113).  Inputs < modulus < 2^63. */
int hexl_amd_eltwise_add_mod(uint64_t* result, const uint64_t* operand1,
                             const uint64_t* operand2, uint64_t n,
                             uint64_t modulus, void* stream);
int hexl_amd_eltwise_add_mod_scalar(uint64_t* result, const uint64_t* operand1,
                                    uint64_t operand2, uint64_t n,
                                    uint64_t modulus, void* stream);

/* EltwiseSubMod (eltwise-sub-mod.hpp:22-37; eltwise-sub-mod.cpp:15-110). */
int hexl_amd_eltwise_sub_mod(uint64_t* result, const uint64_t* operand1,
                             const uint64_t* operand2, uint64_t n,
                             uint64_t modulus, void* stream);
int hexl_amd_eltwise_sub_mod_scalar(uint64_t* result, const uint64_t* operand1,
                                    uint64_t operand2, uint64_t n,
                                    uint64_t modulus, void* stream);

/* EltwiseMultMod (eltwise-mult-mod.hpp:23-25; eltwise-mult-mod.cpp:18-83;
 * eltwise-mult-mod-internal.hpp:34-100).  input_mod_factor in {1,2,4},
 * input_mod_factor * modulus < 2^63, modulus < 2^62. */
int hexl_amd_eltwise_mult_mod(uint64_t* result, const uint64_t* operand1,
                              const uint64_t* operand2, uint64_t n,
                              uint64_t modulus, uint64_t input_mod_factor,
                              void* stream);

/* EltwiseFMAMod: result = (arg1 * arg2 + arg3) mod q, arg3 may be NULL
 * (eltwise-fma-mod.hpp:22-24; eltwise-fma-mod.cpp:17-101;
 * eltwise-fma-mod-internal.hpp:12-39).  input_mod_factor in {1,2,4,8},
 * modulus < 2^61. */
int hexl_amd_eltwise_fma_mod(uint64_t* result, const uint64_t* arg1,
                             uint64_t arg2, const uint64_t* arg3, uint64_t n,
                             uint64_t modulus, uint64_t input_mod_factor,
                             void* stream);

/* EltwiseReduceMod (eltwise-reduce-mod.hpp:24-26; eltwise-reduce-mod.cpp:16-123).
 * input_mod_factor in {modulus, 2, 4}; output_mod_factor in {1, 2}. */
int hexl_amd_eltwise_reduce_mod(uint64_t* result, const uint64_t* operand,
                                uint64_t n, uint64_t modulus,
                                uint64_t input_mod_factor,
                                uint64_t output_mod_factor, void* stream);

/* Fused EltwiseFMAMod + EltwiseReduceMod for the BASELINE config-5 shape:
 * result = ((arg1 mod q) * arg2 + (arg3 mod q)) mod q where arg1/arg3 are
 * arbitrary 64-bit words when input_mod_factor == modulus (single-word Barrett
 * first, eltwise-reduce-mod.cpp:32-55), else as hexl_amd_eltwise_fma_mod. */
int hexl_amd_eltwise_reduce_fma_mod(uint64_t* result, const uint64_t* arg1,
                                    uint64_t arg2, const uint64_t* arg3,
                                    uint64_t n, uint64_t modulus,
                                    uint64_t input_mod_factor, void* stream);

/* EltwiseCmpAdd: result[i] = cmp(operand1[i], bound) ? operand1[i] + diff :
 * operand1[i], plain 64-bit addition
 * (hexl/include/hexl/eltwise/eltwise-cmp-add.hpp:24-25;
 * hexl/eltwise/eltwise-cmp-add.cpp:16-106).  `cmp` is the reference's CMPINT
 * value (hexl/include/hexl/util/util.hpp:16-25): 0 EQ, 1 LT, 2 LE, 3 FALSE,
 * 4 NE, 5 NLT, 6 NLE, 7 TRUE.  n != 0, diff != 0. */
int hexl_amd_eltwise_cmp_add(uint64_t* result, const uint64_t* operand1,
                             uint64_t n, int cmp, uint64_t bound,
                             uint64_t diff, void* stream);

/* EltwiseCmpSubMod: result[i] = cmp(operand1[i], bound) ?
 * (operand1[i] mod modulus - diff) mod modulus : operand1[i] mod modulus
 * (hexl/include/hexl/eltwise/eltwise-cmp-sub-mod.hpp:26-28;
 * hexl/eltwise/eltwise-cmp-sub-mod.cpp:18-66).  Any modulus > 1 (not
 * necessarily prime), arbitrary 64-bit operands, 0 < diff < modulus. */
int hexl_amd_eltwise_cmp_sub_mod(uint64_t* result, const uint64_t* operand1,
                                 uint64_t n, uint64_t modulus, int cmp,
                                 uint64_t bound, uint64_t diff, void* stream);

/* Host-pointer forms for the intel::hexl::Eltwise* shim (synchronous).
 * `op`: 0 add, 1 add_scalar, 2 sub, 3 sub_scalar, 4 mult, 5 fma, 6 reduce.
 * Unused operands are NULL / 0. */
int hexl_amd_eltwise_host(int op, uint64_t* result, const uint64_t* operand1,
                          const uint64_t* operand2, uint64_t scalar,
                          uint64_t n, uint64_t modulus,
                          uint64_t input_mod_factor,
                          uint64_t output_mod_factor);

/* DyadicMultiply (hexl/include/hexl/experimental/seal/dyadic-multiply.hpp:26-28;
 * hexl/experimental/seal/dyadic-multiply-internal.cpp:17-74): ciphertext
 * product (x0, x1) * (y0, y1) -> (x0 y0, x0 y1 + x1 y0, x1 y1) in RNS form.
 * operand1 / operand2: 2 polynomials of n * num_moduli words (device), result:
 * 3 polynomials (may alias an operand); `moduli` is a HOST array of num_moduli
 * words, each 1 < q < 2^62, operands < q.  One fused kernel. */
int hexl_amd_dyadic_multiply(uint64_t* result, const uint64_t* operand1,
                             const uint64_t* operand2, uint64_t n,
                             const uint64_t* moduli, uint64_t num_moduli,
                             void* stream);
/* num_pairs ciphertext pairs with the same moduli in one launch: operands hold num_pairs
 * x 2 polynomials, result num_pairs x 3 (the loop a caller runs around DyadicMultiply).
 * num_pairs <= 65535. */
int hexl_amd_dyadic_multiply_batch(uint64_t* result, const uint64_t* operand1,
                                   const uint64_t* operand2, uint64_t num_pairs, uint64_t n,
                                   const uint64_t* moduli, uint64_t num_moduli,
                                   void* stream);
/* Same as hexl_amd_dyadic_multiply with host buffers (synchronous). */
int hexl_amd_dyadic_multiply_host(uint64_t* result, const uint64_t* operand1,
                                  const uint64_t* operand2, uint64_t n,
                                  const uint64_t* moduli, uint64_t num_moduli);

/* KeySwitch (hexl/include/hexl/experimental/seal/key-switch.hpp:40-46;
 * hexl/experimental/seal/key-switch-internal.cpp:25-201, CKKS path).  All data
 * pointers are device memory: result (key_component_count x decomp_modulus_size
 * x n words, accumulated into), t_target_iter_ptr (decomp_modulus_size x n,
 * NTT form), k_switch_keys[j] (key_component_count x key_modulus_size x n).
 * `moduli`, `modswitch_factors` and the array `k_switch_keys` itself are HOST
 * arrays.  Plans for (n, moduli[i]) are created on first use and cached, as the
 * reference's GetNTT does (ntt-cache.hpp:27-53).  Requires rns_modulus_size ==
 * decomp_modulus_size + 1 <= key_modulus_size, decomp_modulus_size <= 32,
 * NTT-friendly primes below 2^61.
 * Scratch: the intermediates live in a device buffer keyed by (device, stream).
 * Calls on one stream reuse it in stream order -- from several host threads too: a
 * call's launches are enqueued under a per-stream lock; calls on different streams
 * get different buffers and may overlap on the device.
 * Replay: a call on a stream of its own (not NULL) that comes back with the same result,
 * target and key buffers, sizes, moduli and factors as an earlier one is captured into a HIP
 * graph at its second sight and replayed from the third (tuning key "ks_graph"; up to eight
 * buffer sets per stream, least recently used evicted) -- the DATA in the buffers is read at
 * execution time, only the addresses are fixed.  Callers that walk over different buffers run
 * launch by launch, as do calls made while the caller itself is capturing the stream.  What the
 * replay saves is host time (n = 16384, seven decomposition moduli: 5-7 us per call for the enqueue
 * instead of 27-30); on the device both forms are the same chain of dependent kernels (56-57 us per
 * call launch by launch, 59-61 replayed, in a stream of calls; 71-75 either way when each is waited for). */
int hexl_amd_key_switch(uint64_t* result, const uint64_t* t_target_iter_ptr,
                        uint64_t n, uint64_t decomp_modulus_size,
                        uint64_t key_modulus_size, uint64_t rns_modulus_size,
                        uint64_t key_component_count, const uint64_t* moduli,
                        const uint64_t* const* k_switch_keys,
                        const uint64_t* modswitch_factors, void* stream);
/* Many ciphertexts with the same keys and moduli in one call -- the loop SEAL runs around
 * KeySwitch (one call per ciphertext, key-switch-internal.cpp:25-201 each time) as ONE
 * sequence of at most nine launches: t_target_iter_ptr holds num_targets targets back to back
 * (each decomp_modulus_size x n), result num_targets results back to back (each
 * key_component_count x decomp_modulus_size x n, accumulated into).  Every per-modulus
 * transform of every target runs in one multi-plan NTT launch.  num_targets *
 * key_component_count <= 65535. */
int hexl_amd_key_switch_batch(uint64_t* result, const uint64_t* t_target_iter_ptr,
                              uint64_t num_targets, uint64_t n,
                              uint64_t decomp_modulus_size, uint64_t key_modulus_size,
                              uint64_t rns_modulus_size, uint64_t key_component_count,
                              const uint64_t* moduli, const uint64_t* const* k_switch_keys,
                              const uint64_t* modswitch_factors, void* stream);
/* Same as hexl_amd_key_switch with host result / target buffers (synchronous).  Each key block
 * k_switch_keys[j] may be host memory (copied to the device for the call) or device memory
 * (used where it lies): the keys are the bulk of a call's bytes and long-lived, so a caller that
 * uploads them once (hexl_amd_device_alloc + hexl_amd_copy) pays only for result and target. */
int hexl_amd_key_switch_host(uint64_t* result, const uint64_t* t_target_iter_ptr,
                             uint64_t n, uint64_t decomp_modulus_size,
                             uint64_t key_modulus_size, uint64_t rns_modulus_size,
                             uint64_t key_component_count, const uint64_t* moduli,
                             const uint64_t* const* k_switch_keys,
                             const uint64_t* modswitch_factors);

/* Host-pointer forms of the two comparison ops (synchronous).  `modulus` == 0
 * selects EltwiseCmpAdd, anything else EltwiseCmpSubMod. */
int hexl_amd_eltwise_cmp_host(uint64_t* result, const uint64_t* operand1,
                              uint64_t n, uint64_t modulus, int cmp,
                              uint64_t bound, uint64_t diff);

/* ---------------------------------------------------------------------------
 * Host-side scalar number theory the plan builder uses; exported because it
 * is part of the reference's installed API
 * (hexl/include/hexl/number-theory/number-theory.hpp:19-339,
 *  hexl/number-theory/number-theory.cpp:13-261).
 * ------------------------------------------------------------------------- */
uint64_t hexl_amd_multiply_factor(uint64_t operand, uint64_t bit_shift,
                                  uint64_t modulus);
uint64_t hexl_amd_inverse_mod(uint64_t x, uint64_t modulus);
uint64_t hexl_amd_multiply_mod(uint64_t x, uint64_t y, uint64_t modulus);
uint64_t hexl_amd_pow_mod(uint64_t base, uint64_t exp, uint64_t modulus);
int hexl_amd_is_primitive_root(uint64_t root, uint64_t degree,
                               uint64_t modulus);
uint64_t hexl_amd_generate_primitive_root(uint64_t degree, uint64_t modulus);
uint64_t hexl_amd_minimal_primitive_root(uint64_t degree, uint64_t modulus);
uint64_t hexl_amd_reverse_bits(uint64_t x, uint64_t bit_width);
int hexl_amd_is_prime(uint64_t n);
/* Writes up to num_primes primes to out; returns how many were found. */
size_t hexl_amd_generate_primes(uint64_t* out, size_t num_primes,
                                size_t bit_size, int prefer_small_primes,
                                size_t ntt_size);
/* NTT::CheckArguments (ntt-internal.cpp:171-186): 1 if (degree, modulus) is a
 * legal negacyclic NTT parameter set. */
int hexl_amd_ntt_check_arguments(uint64_t degree, uint64_t modulus);

/* ---------------------------------------------------------------------------
 * Bench / test support (device-side synthetic input):
 * poly b of the batch gets coefficients splitmix64(seed0 + b) mod bound.
 * ------------------------------------------------------------------------- */
int hexl_amd_fill_splitmix(uint64_t* data, uint64_t n, uint64_t batch,
                           uint64_t seed0, uint64_t bound, void* stream);

/* Per-kernel timing for bench.py's roofline line: between start and stop every
 * kernel this thread launches through the entry points above is bracketed by a
 * pair of hipEvents recorded on its launch stream.  stop() synchronises the
 * events; get(i) returns the kernel family name and its duration in ms. */
int hexl_amd_profile_start(int max_records);
int hexl_amd_profile_stop(int* num_records);
int hexl_amd_profile_get(int i, const char** name, float* ms);

/* Tuning knobs (process-wide, thread-safe: atomics).  The library reads NO environment
 * variable; every knob has a compiled-in default and changes only through this call.  Results
 * never depend on them, only speed.  Unknown keys / out-of-range values return
 * HEXL_AMD_ERR_INVALID_ARG.  Keys:
 *   "fp64"             1 (default) = plans for 2^30 <= q < 2^50 use the Fp64 arithmetic policy
 *                      (exact integers in doubles), 0 = the integer Lazy policy, 2 = Fp64 also
 *                      below 2^30; read when a plan is created
 *   "fp64_long"        1 (default) = plans for 2^30 <= q < 2^47 use the long-run member of the Fp64
 *                      family (no reduction inside a forward pass, inverse runs of 6 stages), 0 =
 *                      the Fp64 policy of 2^47 <= q < 2^50 for them too; read when a plan is created
 *   "lazy_family"      1 (default) = plans for 2^56 <= q < 2^58 / 2^58 <= q < 2^59 use the bounded
 *                      members of the Lazy arithmetic family (doubled values kept below 32q / 16q:
 *                      a sign-tested subtraction on the stages the host marks instead of one per
 *                      butterfly), 0 = Harvey60 from 2^56 on; read when a plan is created
 *   "h60"              1 (default) = plans for moduli from 2^59 (2^56 with "lazy_family" 0) to
 *                      2^60 + 2^28 use the Harvey60 arithmetic policy (Harvey ranges on doubled
 *                      values, 19/20-instruction butterflies), 0 = the Strict policy; read when a
 *                      plan is created
 *   "tile13"           which degrees above 4096 run as ONE kernel on an LDS tile holding the
 *                      whole polynomial (one HBM round trip instead of two): 2 (default) =
 *                      N = 8192 (64 KiB tile) and N = 16384 (128 KiB tile, batches >= 96),
 *                      1 = N = 8192 only, 0 = neither
 *   "bigtile"          1 (default) = N = 2^18, 2^19 as five strided stages + a 13- / 14-stage
 *                      tile pass (two HBM round trips), 0 = three passes (3 + 3 + 12, 4 + 3 + 12)
 *   "walk14"           the N = 16384 one-kernel plan with more polynomials than the device has
 *                      compute units: 1 (default) = the inverse transform (every arithmetic
 *                      policy) and the forward transform of the Fp64 policies run as ONE
 *                      persistent workgroup per compute unit that walks the polynomials, the next
 *                      polynomial's loads issued ahead of this one's stores (round 6: -9 ... -16 %
 *                      on the inverse, -5 ... -7 % on the Fp64 forward); 0 = one workgroup per
 *                      polynomial everywhere; 2 = the walk everywhere (A/B)
 *   "host_bounce_kb"   largest host-pointer call (KiB of operand) that runs on the per-thread
 *                      pinned, device-mapped bounce buffer instead of staged copies (default 512;
 *                      0 = never)
 *   "host_direct_copy" 0 (default) = the *_host entry points and hexl_amd_copy move ordinary (pageable)
 *                      caller memory through the calling thread's pinned slots (four, 1 MiB each
 *                      until a copy of 4 MiB or more is seen, 4 MiB from then on), host copies
 *                      overlapped with the DMA of the other slots (the runtime is never handed
 *                      pageable caller memory: from about 1 MiB it pins the caller's pages for the
 *                      copy, the path the GPU memory access faults of rounds 4 and 5 sat in,
 *                      EXPERIMENTS.md section 10); *_host transforms of several polynomials and
 *                      8 MiB or more run as a chunk pipeline (copy in | kernels | copy out, both
 *                      directions of the link at once); 1 = such buffers go to hipMemcpyAsync whole
 *   "host_copy_threads" threads that share a host-side copy of 1 MiB or more between caller memory
 *                      and a pinned slot, the calling thread included (default 6; 1 = the calling
 *                      thread alone: one memcpy moves 12-25 GB/s, the link 54 GB/s each way).  The
 *                      helpers are created at the first such copy and sleep otherwise
 *   "host_poll"        1 (default) = a host-pointer call learns that its work is done from a sequence
 *                      number which a one-thread kernel behind the work stores into device-mapped host
 *                      memory and the calling thread polls (for at most a millisecond, then it waits
 *                      in the runtime); 0 = hipStreamSynchronize, whose completion path costs 3-4 us
 *                      more per call (N = 4096: 17.2 -> 13.8 us on mapped memory).  hexl_amd_synchronize
 *                      on a stream from hexl_amd_stream_create waits the same way
 *   "ks_graph"         1 (default) = hexl_amd_key_switch / _host / _batch calls of at most four
 *                      targets whose buffers, keys and moduli were seen before on the same stream
 *                      are replayed from a HIP graph captured at their second sight (one graph
 *                      launch instead of up to nine kernel launches; captured on a stream of the
 *                      library's own, never on the caller's); 0 = always launch by launch
 *   "ks_fuse"          1 (default) = the rounding and finish stages of KeySwitch ride on the load /
 *                      store of the forward transform between them (two launches and two
 *                      intermediate buffers less; degrees from 4096), 0 = stage by stage
 *   "ks_mac_onestep"   1 (default) = the multiply-accumulate of KeySwitch reduces every 128-bit sum
 *                      below 2^(bits(q) + 61) in one generalised Barrett step (decided per coefficient;
 *                      the same canonical residue), 0 = always the two-word form of BarrettReduce128
 *   (the round-2 keys "host_pipeline_min_mb" / "host_chunk_mb" went with the chunked two-stream
 *   host pipeline they switched on: measured equal to the plain sequence, removed in round 5) */
int hexl_amd_set_tuning(const char* key, uint64_t value);
/* Process-wide event counters (monotonic; for tests and the bench line, which must be able to
 * tell a replayed KeySwitch from a launch-by-launch one): "ks_graph_captures" (sequences
 * captured and instantiated), "ks_graph_replays" (calls served by a graph launch), "ks_eager"
 * (calls enqueued launch by launch); "host_polls" (host-pointer calls whose end the polled completion
 * flag reported), "host_poll_timeouts" (those that polled for a millisecond and then waited in the
 * runtime).  Unknown keys return HEXL_AMD_ERR_INVALID_ARG. */
int hexl_amd_get_counter(const char* key, uint64_t* value);

/* Device scratch of the composite entry points (KeySwitch, the experimental one-launch
 * transform) is cached per (device, stream) and grows on demand.  _release_stream_workspaces
 * frees what is keyed by `stream` on the device that OWNS the stream (NULL: the calling
 * thread's current device), whatever the caller's current device is -- hexl_amd_stream_destroy
 * does it for streams it created; call it before destroying a stream of your own that ran such
 * calls; waits for the device.  _release_workspaces frees all of it, buffer by buffer under its
 * stream's sequence lock -- a buffer another thread is enqueueing against at that moment is left
 * alone and the call returns HEXL_AMD_ERR_INVALID_ARG (call again later).  The per-thread
 * streams of the host-pointer entry points release theirs when the thread ends. */
int hexl_amd_release_stream_workspaces(void* stream);
int hexl_amd_release_workspaces(void);

#ifdef __cplusplus
}
#endif
#endif /* HEXL_AMD_H_ */
