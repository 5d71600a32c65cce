/*
 * libalgebra/libalgebra.h — the slice of the libalgebra surface that storm.h callers use,
 * provided by libstorm_hip.so so that `#include "storm.h"` keeps working unchanged.
 *
 * The reference pulls this header from an external submodule (storm.h:33; .gitmodules:1-3,
 * github.com/mklarqvist/libalgebra, absent from the reference checkout). Only the names the
 * reference's own sources touch are provided (call sites: SURVEY.md §8c). In the reference
 * these select a CPU SIMD kernel at run time; here the all-pairs entry points run on the
 * MI355X and a STORM_compute_func value is only an *identity token*: the device path accepts
 * NULL or any leaf exported below (they all denote sum popcount(a & b)) and rejects foreign
 * function pointers (it cannot run caller code on the GPU).
 *
 * STORM_HAVE_SSE42 / AVX2 / AVX512 are defined on x86-64 hosts and the direct SIMD leaves behind them
 * (STORM_intersect_count_sse4 / _avx2 / _avx512; the reference harness compiles its direct-to-SIMD rows
 * under those macros and calls them when STORM_get_cpuid() reports the ISA, benchmark.cpp:949-1045) are
 * exported as HOST functions (stormbitmaps_amd/csrc/storm_leaves.c): one-pair leaves for a caller's own loops
 * — tools/storm_benchmark.cpp times them as CPU rows beside the GPU rows. The all-pairs entry points never
 * run them.
 */
#ifndef STORM_LIBALGEBRA_COMPAT_H_
#define STORM_LIBALGEBRA_COMPAT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what these headers declare is its whole export list */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#if defined(__cplusplus)
#define STORM_RESTRICT __restrict__
#else
#define STORM_RESTRICT restrict
#endif
#define STORM_ALIGN(n) __attribute__((aligned(n)))

#define STORM_HAVE_CPUID 1
#if defined(__x86_64__)
#define STORM_HAVE_SSE42 1
#define STORM_HAVE_AVX2 1
#define STORM_HAVE_AVX512 1
#endif
#define STORM_CPUID_runtime_bit_SSE42 (1 << 0)
#define STORM_CPUID_runtime_bit_AVX2 (1 << 1)
#define STORM_CPUID_runtime_bit_AVX512BW (1 << 2)
#define STORM_CPUID_runtime_bit_GFX950 (1 << 16) /* extension: an MI355X is visible */

/* sum_{k<n} popcount(b1[k] & b2[k]) — the shape fixed by benchmark.cpp:237 and storm.c:144 */
typedef uint64_t (*STORM_compute_func)(const uint64_t* b1, const uint64_t* b2, const size_t n);

/* single-pair host leaf (used by the one-pair helpers of storm.h, never by the all-pairs
 * entry points) and the token returned by STORM_get_intersect_count_func */
uint64_t STORM_intersect_count_scalar(const uint64_t* STORM_RESTRICT b1,
                                      const uint64_t* STORM_RESTRICT b2, const size_t n);
/* benchmark.cpp:1040 flwrapper<&STORM_intersect_count_scalar_list> */
uint64_t STORM_intersect_count_scalar_list(const uint64_t* STORM_RESTRICT b1,
                                           const uint64_t* STORM_RESTRICT b2,
                                           const uint32_t* STORM_RESTRICT l1,
                                           const uint32_t* STORM_RESTRICT l2, const size_t n1,
                                           const size_t n2);
#if defined(__x86_64__)
/* benchmark.cpp:961, :1013, :1031 — call only when STORM_get_cpuid() has the matching STORM_CPUID_runtime_bit_* */
uint64_t STORM_intersect_count_sse4(const uint64_t* STORM_RESTRICT b1, const uint64_t* STORM_RESTRICT b2, const size_t n);
uint64_t STORM_intersect_count_avx2(const uint64_t* STORM_RESTRICT b1, const uint64_t* STORM_RESTRICT b2, const size_t n);
uint64_t STORM_intersect_count_avx512(const uint64_t* STORM_RESTRICT b1, const uint64_t* STORM_RESTRICT b2, const size_t n);
#endif
STORM_compute_func STORM_get_intersect_count_func(const size_t n_bitmaps_vector);
uint32_t STORM_get_alignment(void);
void* STORM_aligned_malloc(size_t alignment, size_t size); /* alignment first: storm.c:452 */
void STORM_aligned_free(void* memblock);
int STORM_get_cpuid(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* STORM_LIBALGEBRA_COMPAT_H_ */
