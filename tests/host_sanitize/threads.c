/* threads.c — TEST ONLY: three caller threads, each with handles of its own, through the host side on the device stub,
 * under ThreadSanitizer (tests/test_host_sanitize.py): the process-wide device lock, the arena fingerprint's helper
 * threads (>= 4096 rows) and the lazily read environment switches. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include "storm.h"
#include "storm_synth.h"
static void* worker(void* p) {
    const uint64_t seed = (uint64_t)(uintptr_t)p;
    STORM_t* s = STORM_new();
    storm_synth_fill_storm(s, 65536, 0, 4500, 9, seed);
    STORM_contiguous_t* c = STORM_contig_new(4096);
    storm_synth_fill_contig(c, 4096, 0, 600, 900, seed);
    uint64_t a = 0, b = 0;
    for (int i = 0; i < 20; ++i) {
        const uint64_t x = STORM_pairw_intersect_cardinality(s), y = STORM_contig_pairw_intersect_cardinality(c);
        if (i && (x != a || y != b)) { fprintf(stderr, "MISMATCH\n"); exit(1); }
        a = x; b = y;
    }
    STORM_free(s);
    STORM_contig_free(c);
    return NULL;
}
int main(void) {
    pthread_t t[3];
    for (int i = 0; i < 3; ++i) pthread_create(&t[i], NULL, worker, (void*)(uintptr_t)(i + 1));
    for (int i = 0; i < 3; ++i) pthread_join(t[i], NULL);
    puts("mt ok");
    return 0;
}
