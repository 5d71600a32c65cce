/* alloc_inject.c — TEST ONLY. Runs a storm.h scenario (both containers, growth steps, all-pairs
 * calls through the device stub, serialize / deserialize) once to count its allocations, then once
 * per allocation with exactly that one failing. Whatever the product returns on a failed
 * allocation is fine as long as it is one of its documented failure values and the process
 * neither crashes nor leaks (AddressSanitizer + LeakSanitizer watch). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "storm.h"

static long g_count = 0, g_fail_at = -1;
static int tick(void) { return ++g_count == g_fail_at; }
void* inject_malloc(size_t n) { return tick() ? NULL : malloc(n); }
void* inject_calloc(size_t a, size_t b) { return tick() ? NULL : calloc(a, b); }
void* inject_realloc(void* p, size_t n) { return tick() ? NULL : realloc(p, n); }
int inject_posix_memalign(void** out, size_t align, size_t n) {
    if (tick()) { *out = NULL; return 12; /* ENOMEM */ }
    return posix_memalign(out, align, n);
}

static uint32_t lcg(uint32_t* s) { return *s = *s * 1664525u + 1013904223u; }

/* returns 0 when every call ended in success or in a documented failure value */
static int scenario(void) {
    uint32_t seed = 7;
    enum { M = 200000, ROWS = 530 };
    uint32_t* vals = (uint32_t*)(malloc)(9000 * sizeof(uint32_t)); /* the scenario's own memory never fails */
    if (!vals) return 1;
    int bad = 0;
    STORM_contiguous_t* c = STORM_contig_new(M);
    STORM_t* s = STORM_new();
    for (int r = 0; c && s && r < ROWS; ++r) {
        const uint32_t n = (r % 7 == 0) ? 8000 : 1 + lcg(&seed) % 60; /* bitmap- and list-kind blocks */
        uint32_t v = lcg(&seed) % 16;
        uint32_t k = 0;
        for (; k < n && v < M; ++k) { vals[k] = v; v += 1 + lcg(&seed) % (M / n); }
        const int rc = STORM_contig_add(c, vals, k);
        if (rc != (int)k && rc != -3) bad = 1;
        const int rs = STORM_add(s, vals, k);
        if (rs != 1 && rs != -3) bad = 1;
    }
    if (c) {
        (void)STORM_contig_pairw_intersect_cardinality_blocked(c, 31); /* total or (uint64_t)-1 */
        uint32_t* out = (uint32_t*)(malloc)((size_t)STORM_contig_n_rows(c) * STORM_contig_n_rows(c) * 4 + 4);
        if (out) {
            const int rc = STORM_contig_pairw_matrix(c, 0, out, STORM_contig_n_rows(c), STORM_contig_n_rows(c));
            if (rc != 0 && rc != -3) bad = 1;
            (free)(out);
        }
    }
    if (s) {
        (void)STORM_pairw_intersect_cardinality(s);
        uint32_t* tri = (uint32_t*)(malloc)((size_t)STORM_n_rows(s) * STORM_n_rows(s) * 4 + 4);
        if (tri) {
            const int rc = STORM_pairw_matrix(s, 0, tri, STORM_n_rows(s), STORM_n_rows(s));
            if (rc != 0 && rc != -3) bad = 1;
            (free)(tri);
        }
        const uint64_t n = STORM_serialized_size(s);
        uint8_t* buf = (uint8_t*)(malloc)(n + 2);
        if (buf) {
            if (STORM_serialize(s, buf, n) == n) {
                STORM_t* back = STORM_deserialize(buf, n); /* NULL when an allocation failed */
                if (back) {
                    if (STORM_serialized_size(back) != n) bad = 1;
                    STORM_free(back);
                }
                (void)STORM_serialized_pairw_intersect_cardinality(buf, n);
            } else {
                bad = 1; /* serialize allocates nothing: it must succeed */
            }
            (free)(buf);
        }
    }
    STORM_contig_free(c);
    STORM_free(s);
    (free)(vals);
    return bad;
}

int main(void) {
    g_fail_at = -1;
    g_count = 0;
    if (scenario()) { printf("alloc inject: scenario fails without injection\n"); return 1; }
    const long total = g_count;
    long failures = 0;
    /* every allocation up to 400, then a stride through the rest (the growth paths repeat) */
    for (long k = 1; k <= total; k += (k < 400 ? 1 : 1 + total / 300)) {
        g_count = 0;
        g_fail_at = k;
        if (scenario()) { printf("alloc inject: undocumented result with allocation %ld of %ld failing\n", k, total); return 1; }
        ++failures;
    }
    printf("alloc inject: ok (%ld allocations in the scenario, %ld of them failed one at a time)\n", total, failures);
    return 0;
}
