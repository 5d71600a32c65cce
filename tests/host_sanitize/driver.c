/* driver.c — TEST ONLY: runs the storm.h host side (containers, growth, marshalling, one-pair
 * helpers) under ASan/UBSan against device_stub.c. The stub's "total" is the number of set bits
 * that reached the device, so every all-pairs call doubles as an integrity check of what the
 * containers stored: it must equal the distinct positions fed in. Exit code 0 = all good. */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "storm.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd(void) {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static int cmp_u32(const void* a, const void* b) {
    const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return (x > y) - (x < y);
}
#define CHECK(cond)                                                             \
    do {                                                                        \
        if (!(cond)) {                                                          \
            fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                           \
        }                                                                       \
    } while (0)

/* one scenario: n_rows rows over [0, width), `draws` draws each (duplicates kept, sorted) */
static int scenario(uint32_t n_rows, uint32_t width, uint32_t draws, int with_empty_rows) {
    STORM_contiguous_t* dense = STORM_contig_new(width);
    STORM_t* sparse = STORM_new();
    CHECK(dense && sparse);
    uint32_t* row = (uint32_t*)malloc((draws + 1) * sizeof(uint32_t));
    const size_t words = (width + 63) / 64;
    uint64_t* bits = (uint64_t*)calloc(words, sizeof(uint64_t));
    uint64_t distinct_dense = 0, distinct_sparse = 0;
    for (uint32_t i = 0; i < n_rows; ++i) {
        const uint32_t n = (with_empty_rows && i % 7 == 3) ? 0 : draws;
        memset(bits, 0, words * sizeof(uint64_t));
        uint64_t distinct = 0;
        for (uint32_t j = 0; j < n; ++j) {
            row[j] = (uint32_t)(rnd() % width);
            const uint64_t bit = 1ULL << (row[j] % 64);
            if (!(bits[row[j] / 64] & bit)) ++distinct;
            bits[row[j] / 64] |= bit;
        }
        qsort(row, n, sizeof(uint32_t), cmp_u32);
        const int rc_d = STORM_contig_add(dense, row, n);
        const int rc_s = STORM_add(sparse, row, n);
        CHECK(rc_d == (int)n);              /* storm.c:1136 returns n_values, 0 for empty */
        CHECK(rc_s == 1);                   /* storm.c:866 */
        if (n) distinct_dense += distinct;  /* an empty row is not appended to the dense container */
        distinct_sparse += distinct;
        if (i % 97 == 0) { /* all-pairs in the middle of construction: mirror rebuilt afterwards */
            CHECK(STORM_contig_pairw_intersect_cardinality(dense) == (dense->n_data < 2 ? 0 : distinct_dense));
        }
    }
    /* the stub returns the set bits that reached the device */
    if (dense->n_data >= 2) {
        CHECK(STORM_contig_pairw_intersect_cardinality(dense) == distinct_dense);
        CHECK(STORM_contig_pairw_intersect_cardinality_blocked(dense, 17) == distinct_dense);
        CHECK(STORM_contig_pairw_intersect_cardinality_list(dense) == distinct_dense);
        CHECK(STORM_contig_pairw_intersect_cardinality_blocked_list(dense, 3) == distinct_dense);
        CHECK(STORM_wrapper_diag(dense->n_data, dense->data, dense->n_bitmaps_vector, NULL) == distinct_dense);
        uint32_t* out = (uint32_t*)malloc((size_t)dense->n_data * dense->n_data * sizeof(uint32_t));
        CHECK(out && STORM_contig_pairw_matrix(dense, 0, out, dense->n_data, dense->n_data) == 0);
        CHECK(STORM_contig_n_rows(dense) == dense->n_data);
        CHECK(dense->n_data < 2 || STORM_contig_pairw_matrix(dense, 0, out, dense->n_data - 1, dense->n_data) == -4);
        free(out);
    }
    if (sparse->n_conts >= 2) {
        CHECK(STORM_pairw_intersect_cardinality(sparse) == distinct_sparse);
        CHECK(STORM_pairw_intersect_cardinality_blocked(sparse, 0) == distinct_sparse);
        const uint64_t n = STORM_n_rows(sparse);
        CHECK(n == sparse->n_conts);
        uint32_t* out = (uint32_t*)malloc((size_t)n * n * sizeof(uint32_t));
        CHECK(out && STORM_pairw_matrix(sparse, 0, out, n, n) == 0);   /* dense replica built beside the arena */
        CHECK(STORM_pairw_matrix(sparse, 0, out, n - 1, n) == -4 && STORM_pairw_matrix(sparse, 0, NULL, n, n) == -2);
        CHECK(STORM_pairw_intersect_cardinality(sparse) == distinct_sparse);
        free(out);
    }
    CHECK(STORM_pairw_matrix(NULL, 0, NULL, 0, 0) == -1);
    CHECK(STORM_serialized_size(sparse) >= 8);
    { /* serialized form: exact size, round trip, truncations rejected, nothing leaked */
        const uint64_t n = STORM_serialized_size(sparse);
        uint8_t* buf = (uint8_t*)malloc(n + 2);
        CHECK(buf && STORM_serialize(sparse, buf, n) == n);
        CHECK(STORM_serialize(sparse, buf, n - 1) == 0);
        STORM_t* back = STORM_deserialize(buf, n);
        CHECK(back && STORM_serialized_size(back) == n && back->n_conts == sparse->n_conts);
        uint8_t* again = (uint8_t*)malloc(n + 2);
        CHECK(again && STORM_serialize(back, again, n) == n && memcmp(buf, again, n) == 0);
        if (sparse->n_conts >= 2) CHECK(STORM_pairw_intersect_cardinality(back) == distinct_sparse);
        STORM_free(back);
        for (uint64_t cut = 0; cut < n; cut += (n / 37) + 1) CHECK(STORM_deserialize(buf, cut) == NULL);
        if (sparse->n_conts >= 2) CHECK(STORM_serialized_pairw_intersect_cardinality(buf, n) != (uint64_t)-1);
        free(again);
        free(buf);
    }
    /* one-pair helpers on the first two rows of the sparse container vs a naive count */
    if (sparse->n_conts >= 2) {
        const uint64_t got = STORM_bitmap_cont_intersect_cardinality(&sparse->conts[0], &sparse->conts[1]);
        uint64_t want = 0;
        for (size_t k = 0; k < dense->n_bitmaps_vector && dense->n_data >= 2 && !with_empty_rows; ++k)
            want += (uint64_t)__builtin_popcountll(dense->data[k] & dense->data[dense->n_bitmaps_vector + k]);
        if (!with_empty_rows && dense->n_data >= 2) CHECK(got == want);
    }
    /* clear keeps the handles usable */
    CHECK(STORM_contig_clear(dense) == 1 && dense->n_data == 0);
    CHECK(STORM_clear(sparse) == 1);
    row[0] = 1 % width;
    CHECK(STORM_contig_add(dense, row, 1) == 1 && STORM_add(sparse, row, 1) == 1);
    CHECK(STORM_contig_pairw_intersect_cardinality(dense) == 0); /* one row: no pairs */
    free(bits);
    free(row);
    STORM_free(sparse);
    STORM_contig_free(dense);
    return 0;
}

/* rows of every kind in one container: below the list cutoff (positions from `scalar`), between the cutoff and
 * W words (positions kept from the add: STORM_contig_add's device-side construction), denser (words); all-pairs
 * calls in between (partial uploads), rows edited in place + STORM_contig_hip_invalidate (everything as words) */
static int scenario_mixed(uint32_t n_rows, uint32_t width) {
    STORM_contiguous_t* dense = STORM_contig_new(width);
    CHECK(dense);
    const uint32_t W = (width + 63) / 64, cutoff = dense->scalar_cutoff;
    const uint32_t max_draws = 4 * W + 8;
    uint32_t* row = (uint32_t*)malloc((max_draws + 1) * sizeof(uint32_t));
    const size_t words = W;
    uint64_t* bits = (uint64_t*)calloc(words, sizeof(uint64_t));
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_rows; ++i) {
        uint32_t n;
        switch (rnd() % 4) {
            case 0: n = 1 + (uint32_t)(rnd() % (cutoff ? cutoff : 1)); break;  /* list kept in `scalar`   */
            case 1: n = cutoff + (uint32_t)(rnd() % (W > cutoff ? W - cutoff : 1)); break; /* kept positions */
            case 2: n = W + (uint32_t)(rnd() % (3 * W)); break;                   /* words                    */
            default: n = (uint32_t)(rnd() % 2) ? W : cutoff; break;              /* the edges                */
        }
        if (n == 0) n = 1;
        memset(bits, 0, words * sizeof(uint64_t));
        for (uint32_t j = 0; j < n; ++j) {
            row[j] = (uint32_t)(rnd() % width);
            const uint64_t bit = 1ULL << (row[j] % 64);
            if (!(bits[row[j] / 64] & bit)) ++total;
            bits[row[j] / 64] |= bit;
        }
        qsort(row, n, sizeof(uint32_t), cmp_u32);
        CHECK(STORM_contig_add(dense, row, n) == (int)n);
        if (i % 61 == 7) CHECK(STORM_contig_pairw_intersect_cardinality(dense) == (dense->n_data < 2 ? 0 : total));
        if (i == n_rows / 2) { /* a caller sets a bit behind the library's back */
            uint64_t* w = dense->data; /* row 0, word 0 */
            if (!(w[0] & 1ULL)) ++total;
            w[0] |= 1ULL;
            CHECK(STORM_contig_hip_invalidate(dense) == 0);
            CHECK(STORM_contig_pairw_intersect_cardinality(dense) == (dense->n_data < 2 ? 0 : total));
        }
    }
    CHECK(STORM_contig_pairw_intersect_cardinality(dense) == total);
    CHECK(STORM_contig_clear(dense) == 1 && dense->n_data == 0);
    row[0] = 0;
    row[1] = width - 1;
    CHECK(STORM_contig_add(dense, row, 2) == 2 && STORM_contig_add(dense, row, 1) == 1);
    CHECK(STORM_contig_pairw_intersect_cardinality(dense) == 3);
    free(bits);
    free(row);
    STORM_contig_free(dense);
    return 0;
}

int main(void) {
    /* error conventions (storm.c:878, :1032-1034, :1150) */
    CHECK(STORM_contig_pairw_intersect_cardinality(NULL) == (uint64_t)-1);
    CHECK(STORM_pairw_intersect_cardinality(NULL) == (uint64_t)-1);
    CHECK(STORM_contig_add(NULL, NULL, 0) == -1);
    {
        STORM_contiguous_t* d = STORM_contig_new(100);
        CHECK(STORM_contig_add(d, NULL, 3) == -2);
        STORM_contig_free(d);
    }
    /* growth paths: > 512 rows (row regrow), > 16384 stored positions (list regrow), sparse rows
     * below the list cutoff, dense blocks >= 4096 values, multi-block rows, width not % 64 */
    if (scenario(1300, 4096, 40, 0)) return 1;      /* list path + both regrows            */
    if (scenario(700, 65536, 13, 1)) return 1;      /* empty rows, tiny lists              */
    if (scenario(40, 200000, 30000, 0)) return 1;   /* bitmap-kind blocks, 4 blocks / row  */
    if (scenario(300, 131071, 4200, 0)) return 1;   /* per-block counts around 4096/2      */
    if (scenario(3, 70, 5, 0)) return 1;
    if (scenario(600, 1000, 128, 0)) return 1;      /* the README's shape (duplicates)     */
    if (scenario(700, 65536, 300, 0)) return 1;     /* rows between the list cutoff and W words: positions kept */
    if (scenario(4500, 65536, 9, 1)) return 1;      /* >= 4096 rows: the arena fingerprint runs on its helper threads */
    if (scenario_mixed(900, 65536)) return 1;
    if (scenario_mixed(300, 20000)) return 1;
    if (scenario_mixed(200, 1000)) return 1;        /* cutoff 5, W 16 */
    puts("host sanitize: ok");
    return 0;
}
