/*
 * storm_synth.h — deterministic synthetic inputs for benchmarks and parity tests.
 *
 * Mirrors what the reference harness does per row (benchmark.cpp:762-772, :563-572): draw
 * `draws` values uniformly on [0, M) WITH replacement, keep each distinct value once, sort.
 * The reference seeds std::mt19937 from std::random_device (benchmark.cpp:756-757), so its
 * inputs are not reproducible; here the stream is counter-based splitmix64 so that the host
 * generator, the numpy restatement in tests/ and the device fill kernel produce identical
 * bits everywhere:
 *
 *     state(n)  = seed + n * 0x9E3779B97F4A7C15          (n = row * draws + i + 1)
 *     z         = state; z = (z ^ z>>30) * 0xBF58476D1CE4E5B9;
 *                        z = (z ^ z>>27) * 0x94D049BB133111EB;  z ^= z>>31
 *     position  = (z * M) >> 64                           (multiply-high range reduction)
 */
#ifndef STORM_SYNTH_H_
#define STORM_SYNTH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what these headers declare is its whole export list */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define STORM_SYNTH_GOLDEN 0x9E3779B97F4A7C15ULL

static inline uint64_t storm_synth_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* the i-th draw of `row` */
static inline uint32_t storm_synth_draw(uint64_t seed, uint64_t n_bits, uint64_t row,
                                        uint32_t draws, uint32_t i) {
    const uint64_t n = row * (uint64_t)draws + i + 1;
    const uint64_t z = storm_synth_mix(seed + n * STORM_SYNTH_GOLDEN);
    return (uint32_t)(((unsigned __int128)z * n_bits) >> 64);
}

/* OR the row's draws into `row_words` (n_words = ceil(n_bits/64) words, caller-zeroed) */
void storm_synth_fill_row(uint64_t* row_words, uint64_t n_bits, uint64_t row, uint32_t draws,
                          uint64_t seed);
/* dense matrix: rows [row0, row0+n_rows) into vals (row stride = stride_words), zeroing first */
void storm_synth_fill_dense(uint64_t* vals, uint64_t stride_words, uint64_t n_bits,
                            uint64_t row0, uint64_t n_rows, uint32_t draws, uint64_t seed);
/* sorted distinct positions of one row; `scratch` = ceil(n_bits/64) words; returns count */
uint32_t storm_synth_positions(uint32_t* out, uint64_t* scratch, uint64_t n_bits, uint64_t row,
                               uint32_t draws, uint64_t seed);

/* Feed rows [row0, row0+n_rows) of the synthetic matrix to the storm.h containers exactly as
 * the reference harness does (STORM_add / STORM_contig_add per row with the sorted distinct
 * positions, benchmark.cpp:794-795). Returns the number of rows added, or -1. */
struct STORM_s;
struct STORM_contiguous_s;
int64_t storm_synth_fill_storm(struct STORM_s* h, uint64_t n_bits, uint64_t row0, uint64_t n_rows,
                               uint32_t draws, uint64_t seed);
int64_t storm_synth_fill_contig(struct STORM_contiguous_s* h, uint64_t n_bits, uint64_t row0,
                                uint64_t n_rows, uint32_t draws, uint64_t seed);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* STORM_SYNTH_H_ */
