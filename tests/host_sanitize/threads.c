/* threads.c — TEST ONLY: caller threads, each with handles of its own, through the host side on the device stub,
 * under ThreadSanitizer (tests/test_host_sanitize.py): the per-slot device locks, the arena fingerprint's helper
 * threads (>= 4096 rows), the lazily read environment switches. Two phases:
 *   1. three threads over the default configuration (one device slot: they take turns behind its lock);
 *   2. three device slots configured, three threads, each narrowed to ITS slot (STORM_hip_set_thread_devices) — they
 *      run side by side — plus a fourth thread that keeps calling a raw-buffer wrapper (which locks all slots). */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include "storm.h"
#include "storm_synth.h"
static int g_phase = 1;
static void* worker(void* p) {
    const uint64_t seed = (uint64_t)(uintptr_t)p;
    if (g_phase == 2 && STORM_hip_set_thread_devices((int)seed - 1, 1) != 0) { fprintf(stderr, "VIEW\n"); exit(1); }
    STORM_t* s = STORM_new();
    storm_synth_fill_storm(s, 65536, 0, 4500, 9, seed);
    STORM_contiguous_t* c = STORM_contig_new(4096);
    storm_synth_fill_contig(c, 4096, 0, 600, 900, seed);
    uint64_t a = 0, b = 0;
    for (int i = 0; i < 20; ++i) {
        const uint64_t x = STORM_pairw_intersect_cardinality(s), y = STORM_contig_pairw_intersect_cardinality(c);
        if (i && (x != a || y != b)) { fprintf(stderr, "MISMATCH\n"); exit(1); }
        a = x; b = y;
        if (g_phase == 2 && i == 10) {   /* widen to all slots and back: the handles follow the view */
            if (STORM_hip_set_thread_devices(0, 0) != 0) exit(1);
            const uint64_t all = STORM_contig_pairw_intersect_cardinality(c);
            if (all != y) { fprintf(stderr, "MISMATCH (all slots)\n"); exit(1); }
            if (STORM_hip_set_thread_devices((int)seed - 1, 1) != 0) exit(1);
        }
    }
    STORM_free(s);
    STORM_contig_free(c);
    return NULL;
}
static void* wrapper_caller(void* p) {
    (void)p;
    enum { N = 300, W = 16 };
    uint64_t* vals = (uint64_t*)calloc((size_t)N * W, sizeof(uint64_t));
    if (!vals) exit(1);
    storm_synth_fill_dense(vals, W, W * 64, 0, N, 200, 7);
    uint64_t first = 0;
    for (int i = 0; i < 20; ++i) {
        const uint64_t t = STORM_wrapper_diag(N, vals, W, NULL);
        if (i && t != first) { fprintf(stderr, "MISMATCH (wrapper)\n"); exit(1); }
        first = t;
    }
    free(vals);
    return NULL;
}
int main(void) {
    pthread_t t[4];
    for (int i = 0; i < 3; ++i) pthread_create(&t[i], NULL, worker, (void*)(uintptr_t)(i + 1));
    for (int i = 0; i < 3; ++i) pthread_join(t[i], NULL);
    g_phase = 2;
    const int ids[3] = {0, 0, 0};   /* three slots (on the stub every slot is its own device) */
    if (STORM_hip_set_devices(3, ids) != 0) return 1;
    if (STORM_hip_set_thread_devices(2, 2) != -1) return 1;   /* outside the configuration */
    for (int i = 0; i < 3; ++i) pthread_create(&t[i], NULL, worker, (void*)(uintptr_t)(i + 1));
    pthread_create(&t[3], NULL, wrapper_caller, NULL);
    for (int i = 0; i < 4; ++i) pthread_join(t[i], NULL);
    /* a view made for three slots outlives a reconfiguration to one: the next call falls back to all (= the one) slot */
    if (STORM_hip_set_thread_devices(2, 1) != 0) return 1;
    const int one[1] = {0};
    if (STORM_hip_set_devices(1, one) != 0) return 1;
    {
        STORM_contiguous_t* c = STORM_contig_new(4096);
        storm_synth_fill_contig(c, 4096, 0, 300, 700, 5);
        const uint64_t a = STORM_contig_pairw_intersect_cardinality(c), b = STORM_contig_pairw_intersect_cardinality(c);
        if (a != b || a == (uint64_t)-1) { fprintf(stderr, "STALE VIEW\n"); return 1; }
        STORM_contig_free(c);
    }
    puts("mt ok");
    return 0;
}
