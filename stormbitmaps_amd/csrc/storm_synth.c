/* storm_synth.c — host side of the deterministic synthetic-input generator (storm_synth.h).
 * Follows the per-row recipe of the reference harness (benchmark.cpp:762-772): draw with
 * replacement, keep distinct values, sorted. */
#include "storm_synth.h"

#include <stdlib.h>
#include <string.h>

#include "storm.h"

void storm_synth_fill_row(uint64_t* row_words, uint64_t n_bits, uint64_t row, uint32_t draws,
                          uint64_t seed) {
    for (uint32_t i = 0; i < draws; ++i) {
        const uint32_t v = storm_synth_draw(seed, n_bits, row, draws, i);
        row_words[v >> 6] |= 1ULL << (v & 63u);
    }
}

void storm_synth_fill_dense(uint64_t* vals, uint64_t stride_words, uint64_t n_bits,
                            uint64_t row0, uint64_t n_rows, uint32_t draws, uint64_t seed) {
    const uint64_t n_words = (n_bits + 63) / 64;
    for (uint64_t r = 0; r < n_rows; ++r) {
        uint64_t* dst = vals + r * stride_words;
        memset(dst, 0, n_words * sizeof(uint64_t));
        storm_synth_fill_row(dst, n_bits, row0 + r, draws, seed);
    }
}

uint32_t storm_synth_positions(uint32_t* out, uint64_t* scratch, uint64_t n_bits, uint64_t row,
                               uint32_t draws, uint64_t seed) {
    const uint64_t n_words = (n_bits + 63) / 64;
    memset(scratch, 0, n_words * sizeof(uint64_t));
    storm_synth_fill_row(scratch, n_bits, row, draws, seed);
    uint32_t n = 0;
    for (uint64_t k = 0; k < n_words; ++k) {
        uint64_t x = scratch[k];
        while (x) {
            out[n++] = (uint32_t)(k * 64 + (uint64_t)__builtin_ctzll(x));
            x &= x - 1;
        }
    }
    return n;
}

static int64_t fill_container(void* h, int sparse, uint64_t n_bits, uint64_t row0, uint64_t n_rows,
                              uint32_t draws, uint64_t seed) {
    if (!h || n_bits == 0) return -1;
    const uint64_t n_words = (n_bits + 63) / 64;
    uint64_t* scratch = (uint64_t*)malloc(n_words * sizeof(uint64_t));
    uint32_t* pos = (uint32_t*)malloc(((size_t)draws + 1) * sizeof(uint32_t));
    int64_t added = -1;
    if (scratch && pos) {
        added = 0;
        for (uint64_t r = 0; r < n_rows; ++r) {
            const uint32_t n = storm_synth_positions(pos, scratch, n_bits, row0 + r, draws, seed);
            const int rc = sparse ? STORM_add((STORM_t*)h, pos, n)
                                  : STORM_contig_add((STORM_contiguous_t*)h, pos, n);
            if (rc < 0) { added = -1; break; }
            ++added;
        }
    }
    free(scratch);
    free(pos);
    return added;
}

int64_t storm_synth_fill_storm(struct STORM_s* h, uint64_t n_bits, uint64_t row0, uint64_t n_rows,
                               uint32_t draws, uint64_t seed) {
    return fill_container(h, 1, n_bits, row0, n_rows, draws, seed);
}

int64_t storm_synth_fill_contig(struct STORM_contiguous_s* h, uint64_t n_bits, uint64_t row0,
                                uint64_t n_rows, uint32_t draws, uint64_t seed) {
    return fill_container(h, 0, n_bits, row0, n_rows, draws, seed);
}
