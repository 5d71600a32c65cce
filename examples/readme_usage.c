/* readme_usage.c — the usage pattern the reference documents (README.md:89-124: build both
 * containers row by row from sorted positions, ask each for the all-pairs intersect total),
 * written against include/storm.h and linked with libstorm_hip.so instead of storm.c. The
 * program checks itself: a plain host loop over bit rows gives the expected total. Duplicate
 * positions in a row (the README draws with rand() % width) count once.
 *
 *   readme_usage [rows] [width] [draws_per_row]      exit code 0 = all totals agree
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "storm.h"

static int by_value(const void* a, const void* b) {
    const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return (x > y) - (x < y);
}

int main(int argc, char** argv) {
    const uint32_t n_rows = argc > 1 ? (uint32_t)strtoul(argv[1], NULL, 10) : 3000;
    const uint32_t width = argc > 2 ? (uint32_t)strtoul(argv[2], NULL, 10) : 1000;
    const uint32_t draws = argc > 3 ? (uint32_t)strtoul(argv[3], NULL, 10) : 128;
    if (n_rows == 0 || width == 0 || draws == 0) return 2;

    STORM_t* sparse = STORM_new();
    STORM_contiguous_t* dense = STORM_contig_new(width);
    const size_t words = (width + 63) / 64;
    uint64_t* bits = (uint64_t*)calloc((size_t)n_rows * words, sizeof(uint64_t));
    uint32_t* row = (uint32_t*)malloc(draws * sizeof(uint32_t));
    if (!sparse || !dense || !bits || !row) return 2;

    srand(12345);
    for (uint32_t i = 0; i < n_rows; ++i) {
        for (uint32_t j = 0; j < draws; ++j) {
            row[j] = (uint32_t)rand() % width;
            bits[i * words + row[j] / 64] |= 1ULL << (row[j] % 64);
        }
        qsort(row, draws, sizeof(uint32_t), by_value); /* both containers want sorted input */
        if (STORM_add(sparse, row, draws) < 0 || STORM_contig_add(dense, row, draws) < 0) return 2;
    }

    uint64_t expected = 0;
    for (uint32_t i = 0; i < n_rows; ++i)
        for (uint32_t j = i + 1; j < n_rows; ++j)
            for (size_t k = 0; k < words; ++k)
                expected += (uint64_t)__builtin_popcountll(bits[i * words + k] & bits[j * words + k]);

    const uint64_t from_sparse = STORM_pairw_intersect_cardinality(sparse);
    const uint64_t from_dense = STORM_contig_pairw_intersect_cardinality(dense);
    const uint64_t from_blocked = STORM_contig_pairw_intersect_cardinality_blocked(dense, 0);
    printf("contig=%" PRIu64 " contig_blocked=%" PRIu64 " storm=%" PRIu64 " expected=%" PRIu64 "\n",
           from_dense, from_blocked, from_sparse, expected);
    if (from_dense == (uint64_t)-1 || from_sparse == (uint64_t)-1)
        fprintf(stderr, "device error: %s\n", STORM_hip_error());

    free(row);
    free(bits);
    STORM_free(sparse);
    STORM_contig_free(dense);
    return (from_sparse == expected && from_dense == expected && from_blocked == expected) ? 0 : 1;
}
