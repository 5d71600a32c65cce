/* storm_leaves.c — the direct SIMD leaves of the libalgebra surface (STORM_intersect_count_sse4 / _avx2 /
 * _avx512), host side of libstorm_hip.so.
 *
 * The reference's harness binds these names under STORM_HAVE_SSE42 / AVX2 / AVX512 and times its own blocked
 * loop over them on ONE host thread (benchmark.cpp:949-1045, fwrapper_blocked :256-316); they come from the
 * un-vendored libalgebra (storm.h:33). Each computes the ONE-PAIR leaf sum_{k<n} popcount(b1[k] & b2[k])
 * (shape: benchmark.cpp:237, storm.c:144). Written from scratch:
 *   sse4   : popcnt r64 on the AND of the words, four independent chains;
 *   avx2   : 256-bit AND, per-byte popcount by two 16-entry nibble look-ups (vpshufb), byte sums widened by
 *            vpsadbw every 8 vectors at most (a byte holds at most 8 per vector);
 *   avx512 : vpopcntq on the 512-bit AND where the CPU has AVX512-VPOPCNTDQ, else the nibble look-up on
 *            512-bit vectors (AVX512BW).
 * They are host functions for callers' own loops (tools/storm_benchmark.cpp prints them as CPU rows beside
 * the GPU rows). The all-pairs entry points of storm.h never call them: there a STORM_compute_func is an
 * identity token and the work runs on the MI355X (no CPU fallback). Each is compiled for its own ISA by a
 * function attribute and must only be called when STORM_get_cpuid() reports the ISA — the reference's rule.
 */
#include <stddef.h>
#include <stdint.h>

#include "storm.h"

#if defined(__x86_64__)
#include <immintrin.h>

/* The reference's rule — call a leaf only when STORM_get_cpuid() reports its ISA — is documented, not enforced by the
 * STORM_HAVE_* macros (libalgebra.h defines them for every x86-64 compile). So each exported leaf checks the running CPU
 * once and falls back to the portable loop where the ISA is missing, instead of faulting. */
static uint64_t count_portable(const uint64_t* STORM_RESTRICT b1, const uint64_t* STORM_RESTRICT b2, const size_t n) {
    uint64_t c = 0;
    for (size_t k = 0; k < n; ++k) c += (uint64_t)__builtin_popcountll(b1[k] & b2[k]);
    return c;
}
/* 0 = portable loop, 1 = sse4.2 + popcnt, 2 = avx2, 3 = avx512bw, 4 = avx512bw + vpopcntdq (benign race: every thread
 * computes the same value) */
static int cpu_level(void) {
    static int level = -1;
    int v = __atomic_load_n(&level, __ATOMIC_RELAXED);
    if (v < 0) {
        __builtin_cpu_init();
        v = 0;
        if (__builtin_cpu_supports("sse4.2") && __builtin_cpu_supports("popcnt")) v = 1;
        if (v == 1 && __builtin_cpu_supports("avx2")) v = 2;
        if (v == 2 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw")) v = 3;
        if (v == 3 && __builtin_cpu_supports("avx512vpopcntdq")) v = 4;
        __atomic_store_n(&level, v, __ATOMIC_RELAXED);
    }
    return v;
}

__attribute__((target("sse4.2,popcnt")))
static uint64_t count_sse4(const uint64_t* STORM_RESTRICT b1, const uint64_t* STORM_RESTRICT b2, const size_t n) {
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    size_t k = 0;
    for (; k + 4 <= n; k += 4) {
        c0 += (uint64_t)_mm_popcnt_u64(b1[k] & b2[k]);
        c1 += (uint64_t)_mm_popcnt_u64(b1[k + 1] & b2[k + 1]);
        c2 += (uint64_t)_mm_popcnt_u64(b1[k + 2] & b2[k + 2]);
        c3 += (uint64_t)_mm_popcnt_u64(b1[k + 3] & b2[k + 3]);
    }
    for (; k < n; ++k) c0 += (uint64_t)_mm_popcnt_u64(b1[k] & b2[k]);
    return c0 + c1 + c2 + c3;
}

__attribute__((target("avx2,popcnt")))
static uint64_t count_avx2(const uint64_t* STORM_RESTRICT b1, const uint64_t* STORM_RESTRICT b2, const size_t n) {
    const __m256i lut = _mm256_setr_epi8(0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4,
                                         0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4);
    const __m256i low = _mm256_set1_epi8(0x0f);
    __m256i total = _mm256_setzero_si256();
    size_t k = 0;
    while (k + 4 <= n) {
        __m256i bytes = _mm256_setzero_si256();
        /* at most 8 vectors per round: a byte lane gains at most 8 per vector, 64 in all */
        for (int r = 0; r < 8 && k + 4 <= n; ++r, k += 4) {
            const __m256i v = _mm256_and_si256(_mm256_loadu_si256((const __m256i*)(b1 + k)),
                                               _mm256_loadu_si256((const __m256i*)(b2 + k)));
            const __m256i lo = _mm256_shuffle_epi8(lut, _mm256_and_si256(v, low));
            const __m256i hi = _mm256_shuffle_epi8(lut, _mm256_and_si256(_mm256_srli_epi16(v, 4), low));
            bytes = _mm256_add_epi8(bytes, _mm256_add_epi8(lo, hi));
        }
        total = _mm256_add_epi64(total, _mm256_sad_epu8(bytes, _mm256_setzero_si256()));
    }
    uint64_t lanes[4];
    _mm256_storeu_si256((__m256i*)lanes, total);
    uint64_t count = lanes[0] + lanes[1] + lanes[2] + lanes[3];
    for (; k < n; ++k) count += (uint64_t)_mm_popcnt_u64(b1[k] & b2[k]);
    return count;
}

__attribute__((target("avx512f,avx512bw,avx512vpopcntdq,popcnt")))
static uint64_t count_avx512_vpopcnt(const uint64_t* STORM_RESTRICT b1, const uint64_t* STORM_RESTRICT b2,
                                     const size_t n) {
    __m512i t0 = _mm512_setzero_si512(), t1 = _mm512_setzero_si512();
    size_t k = 0;
    for (; k + 16 <= n; k += 16) {
        t0 = _mm512_add_epi64(t0, _mm512_popcnt_epi64(_mm512_and_si512(_mm512_loadu_si512(b1 + k),
                                                                       _mm512_loadu_si512(b2 + k))));
        t1 = _mm512_add_epi64(t1, _mm512_popcnt_epi64(_mm512_and_si512(_mm512_loadu_si512(b1 + k + 8),
                                                                       _mm512_loadu_si512(b2 + k + 8))));
    }
    for (; k + 8 <= n; k += 8)
        t0 = _mm512_add_epi64(t0, _mm512_popcnt_epi64(_mm512_and_si512(_mm512_loadu_si512(b1 + k),
                                                                       _mm512_loadu_si512(b2 + k))));
    uint64_t count = (uint64_t)_mm512_reduce_add_epi64(_mm512_add_epi64(t0, t1));
    for (; k < n; ++k) count += (uint64_t)_mm_popcnt_u64(b1[k] & b2[k]);
    return count;
}

__attribute__((target("avx512f,avx512bw,popcnt")))
static uint64_t count_avx512_bw(const uint64_t* STORM_RESTRICT b1, const uint64_t* STORM_RESTRICT b2,
                                const size_t n) {
    const __m512i lut = _mm512_broadcast_i32x4(_mm_setr_epi8(0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4));
    const __m512i low = _mm512_set1_epi8(0x0f);
    __m512i total = _mm512_setzero_si512();
    size_t k = 0;
    while (k + 8 <= n) {
        __m512i bytes = _mm512_setzero_si512();
        for (int r = 0; r < 8 && k + 8 <= n; ++r, k += 8) {
            const __m512i v = _mm512_and_si512(_mm512_loadu_si512(b1 + k), _mm512_loadu_si512(b2 + k));
            const __m512i lo = _mm512_shuffle_epi8(lut, _mm512_and_si512(v, low));
            const __m512i hi = _mm512_shuffle_epi8(lut, _mm512_and_si512(_mm512_srli_epi16(v, 4), low));
            bytes = _mm512_add_epi8(bytes, _mm512_add_epi8(lo, hi));
        }
        total = _mm512_add_epi64(total, _mm512_sad_epu8(bytes, _mm512_setzero_si512()));
    }
    uint64_t count = (uint64_t)_mm512_reduce_add_epi64(total);
    for (; k < n; ++k) count += (uint64_t)_mm_popcnt_u64(b1[k] & b2[k]);
    return count;
}

uint64_t STORM_intersect_count_sse4(const uint64_t* STORM_RESTRICT b1, const uint64_t* STORM_RESTRICT b2,
                                    const size_t n) {
    return cpu_level() >= 1 ? count_sse4(b1, b2, n) : count_portable(b1, b2, n);
}

uint64_t STORM_intersect_count_avx2(const uint64_t* STORM_RESTRICT b1, const uint64_t* STORM_RESTRICT b2,
                                    const size_t n) {
    return cpu_level() >= 2 ? count_avx2(b1, b2, n) : STORM_intersect_count_sse4(b1, b2, n);
}

uint64_t STORM_intersect_count_avx512(const uint64_t* STORM_RESTRICT b1, const uint64_t* STORM_RESTRICT b2,
                                      const size_t n) {
    const int v = cpu_level();
    return v >= 4 ? count_avx512_vpopcnt(b1, b2, n) : v == 3 ? count_avx512_bw(b1, b2, n) : STORM_intersect_count_avx2(b1, b2, n);
}

#endif /* __x86_64__ */
