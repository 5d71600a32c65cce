/* bench_contig_add.c — what a caller pays to get a STORM_contiguous_t onto the device:
 * N rows of `draws` synthetic positions over M bits through STORM_contig_add (storm.c:1031-1137), then the
 * first and the second STORM_contig_pairw_intersect_cardinality. With STORM_HIP_ADD_POSITIONS=1 (default) rows
 * of at most W = M / 64 distinct positions reach the device as positions (4 B each, set_bits_kernel), with =0
 * as their W words. A tiny container is run first so that the device context exists (rows are then streamed
 * in batches of 256 during the adds, as in a process that has used the device before).
 *   gcc -O2 -Iinclude -o tools/bench_contig_add tools/bench_contig_add.c -Lstormbitmaps_amd -lstorm_hip -Wl,-rpath,'$ORIGIN/../stormbitmaps_amd'
 *   tools/bench_contig_add <M> <N> <draws> [reps] */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "storm.h"
#include "storm_synth.h"

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <M> <N> <draws> [reps]\n", argv[0]);
        return 2;
    }
    const uint64_t M = strtoull(argv[1], NULL, 10), N = strtoull(argv[2], NULL, 10);
    const uint32_t draws = (uint32_t)strtoul(argv[3], NULL, 10);
    const int reps = argc > 4 ? atoi(argv[4]) : 3;
    { /* device context + clock */
        STORM_contiguous_t* w = STORM_contig_new(4096);
        storm_synth_fill_contig(w, 4096, 0, 300, 400, 1);
        for (int i = 0; i < 20; ++i) (void)STORM_contig_pairw_intersect_cardinality(w);
        STORM_contig_free(w);
    }
    /* the rows, generated once outside the timed part */
    uint32_t* pos = (uint32_t*)malloc((size_t)N * (draws + 1) * sizeof(uint32_t));
    uint64_t* off = (uint64_t*)malloc((size_t)(N + 1) * sizeof(uint64_t));
    uint64_t* scratch = (uint64_t*)calloc((size_t)(M + 63) / 64, sizeof(uint64_t));
    if (!pos || !off || !scratch) return 1;
    off[0] = 0;
    for (uint64_t i = 0; i < N; ++i) off[i + 1] = off[i] + storm_synth_positions(pos + off[i], scratch, M, i, draws, 42);
    double best_add = 1e30, best_first = 1e30, best_second = 1e30;
    uint64_t total = 0;
    for (int r = 0; r < reps; ++r) {
        STORM_contiguous_t* c = STORM_contig_new(M);
        const double t0 = now_ms();
        for (uint64_t i = 0; i < N; ++i) {
            const uint32_t n = (uint32_t)(off[i + 1] - off[i]);
            if (STORM_contig_add(c, pos + off[i], n) != (int)n) return 1;
        }
        const double t1 = now_ms();
        total = STORM_contig_pairw_intersect_cardinality(c);
        const double t2 = now_ms();
        if (STORM_contig_pairw_intersect_cardinality(c) != total) return 1;
        const double t3 = now_ms();
        if (t1 - t0 < best_add) best_add = t1 - t0;
        if (t2 - t1 < best_first) best_first = t2 - t1;
        if (t3 - t2 < best_second) best_second = t3 - t2;
        STORM_contig_free(c);
    }
    const char* e = getenv("STORM_HIP_ADD_POSITIONS");
    const char* s = getenv("STORM_HIP_STREAM_ROWS");
    printf("{\"M\": %llu, \"N\": %llu, \"draws\": %u, \"add_positions\": \"%s\", \"stream_rows\": \"%s\", "
           "\"adds_ms\": %.3f, \"first_call_ms\": %.3f, \"adds_plus_first_ms\": %.3f, \"second_call_ms\": %.3f, \"total\": %llu}\n",
           (unsigned long long)M, (unsigned long long)N, draws, e ? e : "default", s ? s : "default", best_add,
           best_first, best_add + best_first, best_second, (unsigned long long)total);
    free(pos);
    free(off);
    free(scratch);
    return 0;
}
