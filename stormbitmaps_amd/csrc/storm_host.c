/*
 * storm_host.c — host side (C) of the storm.h API served by libstorm_hip.so.
 *
 * Containers are built on the host exactly as a caller of the reference expects (same public
 * struct members, same growth steps, same return codes; reference lines cited per function),
 * and every ALL-PAIRS entry point hands the data to the MI355X through the C-ABI shim of
 * storm_hip.h. Nothing in this file computes an all-pairs total on the CPU: if the device
 * path fails the entry points return (uint64_t)-1 and STORM_hip_error() says why.
 *
 * The only CPU arithmetic here is in the ONE-PAIR helpers of the reference API
 * (STORM_bitmap_intersect_cardinality & co., STORM_intersect_*): they answer a question
 * about two host-resident blocks and are never called by the all-pairs functions.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "storm.h"
#include <pthread.h>

#include "storm_hip.h"

#define BLOCK_BITS ((uint32_t)STORM_DEFAULT_BLOCK_SIZE)
#define BLOCK_WORDS (BLOCK_BITS / 64u)
#define MAX_DEVICES 16
#define ALL_PAIRS_FAILED ((uint64_t)-1)

/* ------------------------------------------------------------------------------------------
 * device runtime state (process-wide, like the reference: no locking, single caller thread)
 * ---------------------------------------------------------------------------------------- */
static int g_n_devices = 0; /* 0 = not configured yet */
static uint32_t g_config_generation = 1; /* bumped by STORM_hip_set_devices */
static int g_device_ids[MAX_DEVICES];
static storm_hip_ctx_t* g_ctx[MAX_DEVICES];
static uint32_t g_shard_rank = 0, g_shard_count = 1;
/* A caller thread may narrow what ITS all-pairs calls use to a run of the configured device slots
 * (STORM_hip_set_thread_devices): handles driven from that thread keep their device mirrors there, and only those
 * slots are locked — threads on distinct slots (GPUs) overlap. 0 slots = all of them (the default). */
static __thread int tl_view_first = 0, tl_view_count = 0;
static pthread_mutex_t g_config_mu = PTHREAD_MUTEX_INITIALIZER; /* guards the device configuration itself */
static void configure_from_env(void);
#define V0 (tl_view_count ? tl_view_first : 0)
#define VN (tl_view_count ? tl_view_count : g_n_devices)
#define V1 (V0 + VN)
/* what a handle's device state was built for: the device configuration AND the caller's slots */
#define VIEW_GENERATION (g_config_generation * 4096u + (uint32_t)V0 * 64u + (uint32_t)VN)
static __thread char g_host_error[256] = ""; /* per calling thread, like the device library's last error */

const char* STORM_hip_error(void) {
    return g_host_error[0] ? g_host_error : storm_hip_last_error();
}

static void host_error(const char* msg) {
    snprintf(g_host_error, sizeof(g_host_error), "%s", msg);
    fprintf(stderr, "[storm_hip] %s\n", msg);
}

static void device_error(const char* where) {
    g_host_error[0] = '\0';
    fprintf(stderr, "[storm_hip] %s: %s\n", where, storm_hip_last_error());
}

static void wrapper_states_release(void);
static storm_hip_comm_t* g_comm = NULL; /* multi-process runs: the partials of the ranks are all-reduced */

int STORM_hip_comm_unique_id(uint8_t id[128]) {
    if (storm_hip_comm_unique_id(id) != STORM_HIP_OK) {
        g_host_error[0] = '\0';
        fprintf(stderr, "[storm_hip] STORM_hip_comm_unique_id: %s\n", storm_hip_last_error());
        return -1;
    }
    return 0;
}

/* (not to be called while another thread is inside an all-pairs call; a thread's view that no longer fits the new
 * configuration falls back to all slots at that thread's next call) */
int STORM_hip_set_devices(int n_devices, const int* device_ids) {
    if (n_devices < 1 || n_devices > MAX_DEVICES || !device_ids) return -1;
    wrapper_states_release(); /* their matrices live on the contexts that go away here */
    for (int d = 0; d < MAX_DEVICES; ++d) {
        if (g_ctx[d]) storm_hip_ctx_destroy(g_ctx[d]);
        g_ctx[d] = NULL;
    }
    pthread_mutex_lock(&g_config_mu);
    for (int d = 0; d < n_devices; ++d) g_device_ids[d] = device_ids[d];
    __atomic_store_n(&g_n_devices, n_devices, __ATOMIC_RELEASE);
    ++g_config_generation; /* device mirrors cached in handles are rebuilt on their next use */
    pthread_mutex_unlock(&g_config_mu);
    if (tl_view_first + tl_view_count > n_devices) tl_view_first = tl_view_count = 0;
    return 0;
}

/* What the last all-pairs call of this process ran, over its configured devices (storm_hip_last_pass_report per
 * context: kernel mask OR-ed, work summed). -1 NULL argument. */
int STORM_hip_last_pass(uint64_t out[4]) {
    if (!out) return -1;
    memset(out, 0, 4 * sizeof(uint64_t));
    for (int d = V0; d < V1; ++d) {
        uint64_t r[4];
        if (!g_ctx[d] || storm_hip_last_pass_report(g_ctx[d], r) != STORM_HIP_OK) continue;
        out[0] |= r[0];
        out[1] += r[1];
        out[2] += r[2];
        if (r[3]) out[3] = r[3];
    }
    return 0;
}

/* The calling thread's all-pairs calls use the device slots [first_slot, first_slot + n_slots) of the configured
 * devices from now on (n_slots = 0: all of them again). -1 if the run is not inside the configuration. */
int STORM_hip_set_thread_devices(int first_slot, int n_slots) {
    configure_from_env();
    if (n_slots == 0) {
        tl_view_first = tl_view_count = 0;
        return 0;
    }
    if (first_slot < 0 || n_slots < 1 || first_slot + n_slots > g_n_devices) return -1;
    if (g_comm) {   /* the communicator's stream and buffers live on slot 0: a thread that does not hold slot 0's lock must not
                       drive them: one process per GPU has one slot anyway */
        host_error("STORM_hip_set_thread_devices: refused while a communicator is attached (STORM_hip_comm_init)");
        return -1;
    }
    tl_view_first = first_slot;
    tl_view_count = n_slots;
    return 0;
}

int STORM_hip_set_shard(uint32_t shard_rank, uint32_t shard_count) {
    if (shard_count == 0 || shard_rank >= shard_count) return -1;
    g_shard_rank = shard_rank;
    g_shard_count = shard_count;
    return 0;
}

/* STORM_HIP_DEVICES = "all" | "0,2,3";  STORM_HIP_SHARD = "rank/count" */
static void configure_from_env_locked(void);
static void configure_from_env(void) {
    const int n = __atomic_load_n(&g_n_devices, __ATOMIC_ACQUIRE);
    if (n != 0) {
        /* a view made for an earlier, larger configuration (STORM_hip_set_devices since): back to all slots */
        if (tl_view_count && tl_view_first + tl_view_count > n) tl_view_first = tl_view_count = 0;
        return;
    }
    pthread_mutex_lock(&g_config_mu);
    configure_from_env_locked();
    pthread_mutex_unlock(&g_config_mu);
}
static void configure_from_env_locked(void) {
    if (g_n_devices != 0) return;
    int n_devices = 0;
    const char* devs = getenv("STORM_HIP_DEVICES");
    if (devs && !strcmp(devs, "all")) {
        int n = storm_hip_device_count();
        if (n > MAX_DEVICES) n = MAX_DEVICES;
        for (int d = 0; d < n; ++d) g_device_ids[d] = d;
        n_devices = n > 0 ? n : 1;
    } else if (devs && devs[0]) {
        int n = 0;
        const char* p = devs;
        while (*p && n < MAX_DEVICES) {
            g_device_ids[n++] = (int)strtol(p, (char**)&p, 10);
            if (*p == ',') ++p;
        }
        n_devices = n > 0 ? n : 1;
    } else {
        g_device_ids[0] = 0;
        n_devices = 1;
    }
    const char* shard = getenv("STORM_HIP_SHARD");
    unsigned r = 0, c = 1;
    if (shard && sscanf(shard, "%u/%u", &r, &c) == 2 && c > 0 && r < c) {
        g_shard_rank = r;
        g_shard_count = c;
    }
    __atomic_store_n(&g_n_devices, n_devices, __ATOMIC_RELEASE); /* last: readers take the fast path from here on */
}

static __thread int g_quiet_ctx = 0; /* 1 while a best-effort caller (row streaming) asks for a context */

/* Extension (storm.h): context options (include/storm_hip.h: storm_hip_ctx_set_option) for the contexts behind the storm.h
 * handles — applied to the ones that exist and remembered for the ones still to be made. STORM_HIP_OPTIONS="key=value,..." in
 * the environment does the same at the first use. Tuning and A/B only: every option keeps the results exact. */
#define MAX_HOST_OPTIONS 16
static struct { char key[40]; int64_t value; } g_host_opt[MAX_HOST_OPTIONS];
static int g_n_host_opt = 0;
static pthread_once_t g_env_opts_once = PTHREAD_ONCE_INIT;
static void apply_host_options(storm_hip_ctx_t* ctx) {
    /* (every remembered option has passed storm_hip_option_check: a failure here would be a defect, not a typo) */
    for (int i = 0; i < g_n_host_opt; ++i)
        if (storm_hip_ctx_set_option(ctx, g_host_opt[i].key, g_host_opt[i].value) != STORM_HIP_OK)
            fprintf(stderr, "libstorm_hip: option %s=%lld was not applied: %s\n", g_host_opt[i].key,
                    (long long)g_host_opt[i].value, storm_hip_last_error());
}
static int remember_host_option(const char* key, int64_t value) {
    for (int i = 0; i < g_n_host_opt; ++i)
        if (!strcmp(g_host_opt[i].key, key)) {
            g_host_opt[i].value = value;
            return 0;
        }
    if (g_n_host_opt == MAX_HOST_OPTIONS || strlen(key) >= sizeof(g_host_opt[0].key)) return -1;
    snprintf(g_host_opt[g_n_host_opt].key, sizeof(g_host_opt[0].key), "%s", key);
    g_host_opt[g_n_host_opt++].value = value;
    return 0;
}
/* STORM_HIP_OPTIONS, once per process (pthread_once: two threads with disjoint device views may create their first
 * contexts at the same time), every entry validated before it is remembered; a typo is reported, not dropped silently */
static void read_env_options_once(void) {
    const char* e = getenv("STORM_HIP_OPTIONS");
    if (!e) return;
    char buf[512];
    snprintf(buf, sizeof(buf), "%s", e);
    char* save = NULL;
    for (char* tok = strtok_r(buf, ",", &save); tok; tok = strtok_r(NULL, ",", &save)) {
        char* eq = strchr(tok, '=');
        if (eq) *eq = '\0';
        char* end = NULL;
        const long long v = eq ? strtoll(eq + 1, &end, 10) : 0;
        if (!eq || end == eq + 1 || *end != '\0' || storm_hip_option_check(tok, v) != STORM_HIP_OK ||
            remember_host_option(tok, v) != 0)
            fprintf(stderr, "libstorm_hip: STORM_HIP_OPTIONS: entry \"%s%s%s\" ignored%s%s\n", tok, eq ? "=" : "", eq ? eq + 1 : "",
                    eq ? ": " : " (key=value expected)", eq ? storm_hip_last_error() : "");
    }
}
static void read_env_options(void) { pthread_once(&g_env_opts_once, read_env_options_once); }

static storm_hip_ctx_t* device_ctx(int slot) {
    configure_from_env();
    if (slot < 0 || slot >= g_n_devices) return NULL;
    if (!g_ctx[slot]) {
        if (storm_hip_ctx_create(g_device_ids[slot], NULL, &g_ctx[slot]) != STORM_HIP_OK) {
            if (!g_quiet_ctx) device_error("storm_hip_ctx_create");
            g_ctx[slot] = NULL;
        } else {
            /* the device mirrors behind the storm.h handles are written by this file only, so a
             * repeated all-pairs call on an unchanged handle may reuse the FP4 shadow */
            (void)storm_hip_ctx_set_option(g_ctx[slot], "keep_shadow", 1);
            read_env_options();
            apply_host_options(g_ctx[slot]);
        }
    }
    return g_ctx[slot];
}

/* ------------------------------------------------------------------------------------------
 * One worker thread per configured device beyond the first (created on the first multi-device call, parked
 * on a condition variable): a pass over G devices is G launches + G result reads, and issued one after the
 * other from the caller's thread they cost the host more than a 1/8 shard of the headline pass takes the GPUs
 * (a shard: ~118 us; eight launch + read-back rounds in a row: ~300 us). The caller's thread drives slot 0.
 * ---------------------------------------------------------------------------------------- */
typedef int (*slot_fn)(int slot, int phase, void* arg); /* phase 0: launch, 1: fetch the result */
static struct {
    pthread_t th[MAX_DEVICES];
    int started[MAX_DEVICES];
    pthread_mutex_t mu;
    pthread_cond_t go, done;
    uint64_t generation;
    int pending, n_slots;
    slot_fn fn;
    void* arg;
    int rc[MAX_DEVICES];
    char err[MAX_DEVICES][256];
} g_pool = {.mu = PTHREAD_MUTEX_INITIALIZER, .go = PTHREAD_COND_INITIALIZER, .done = PTHREAD_COND_INITIALIZER};

static void* pool_main(void* p) {
    const int slot = (int)(intptr_t)p;
    uint64_t seen = 0;
    for (;;) {
        pthread_mutex_lock(&g_pool.mu);
        while (g_pool.generation == seen) pthread_cond_wait(&g_pool.go, &g_pool.mu);
        seen = g_pool.generation;
        const slot_fn fn = g_pool.fn;
        void* const arg = g_pool.arg;
        const int mine = slot < g_pool.n_slots;
        pthread_mutex_unlock(&g_pool.mu);
        if (!mine) continue;
        int rc = fn(slot, 0, arg);
        if (!rc) rc = fn(slot, 1, arg);
        if (rc) snprintf(g_pool.err[slot], sizeof(g_pool.err[slot]), "%s", storm_hip_last_error());
        pthread_mutex_lock(&g_pool.mu);
        g_pool.rc[slot] = rc;
        if (--g_pool.pending == 0) pthread_cond_signal(&g_pool.done);
        pthread_mutex_unlock(&g_pool.mu);
    }
    return NULL;
}

/* Threads only pay when the slots are different GPUs: the HIP runtime serialises calls to ONE device, and the
 * same ordinal configured several times (a rehearsal on one card) is faster driven from one thread (measured,
 * tools/archive/bench_inprocess.py: 246 against 324 us per call for 8 contexts on one MI355X).
 * STORM_HIP_HOST_THREADS=0 / 1 overrides. */
static int use_host_threads(void) {
    const char* e = getenv("STORM_HIP_HOST_THREADS");
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
    for (int a = 0; a < g_n_devices; ++a)
        for (int b = a + 1; b < g_n_devices; ++b)
            if (g_device_ids[a] == g_device_ids[b]) return 0;
    return 1;
}

/* fn(slot, launch) then fn(slot, fetch) on every configured device; 0 if all returned 0 */
static int run_on_devices(slot_fn fn, void* arg, const char* what) {
    const int lo = V0, n = VN;
    /* the worker pool serves one job at a time and drives slots 1 .. n - 1: only for the full configuration (a thread
     * with a narrower view drives its slots itself, typically one) */
    int rc0 = 0, threads = n > 1 && lo == 0 && n == g_n_devices && use_host_threads();
    for (int d = 1; threads && d < n; ++d)
        if (!g_pool.started[d]) {
            if (pthread_create(&g_pool.th[d], NULL, pool_main, (void*)(intptr_t)d) != 0) {
                threads = 0;
            } else {
                pthread_detach(g_pool.th[d]);
                g_pool.started[d] = 1;
            }
        }
    if (!threads) { /* one thread: every device launched before the first result is waited for */
        int launched = 0;
        for (int k = 0; k < n && !rc0; ++k) {
            rc0 = fn(lo + k, 0, arg);
            if (!rc0) launched = k + 1;
        }
        for (int k = 0; k < launched; ++k) {
            const int r = fn(lo + k, 1, arg);
            if (r && !rc0) rc0 = r;
        }
        if (rc0) device_error(what);
        return rc0;
    }
    pthread_mutex_lock(&g_pool.mu);
    g_pool.fn = fn;
    g_pool.arg = arg;
    g_pool.n_slots = n;
    g_pool.pending = n - 1;
    ++g_pool.generation;
    pthread_cond_broadcast(&g_pool.go);
    pthread_mutex_unlock(&g_pool.mu);
    rc0 = fn(0, 0, arg);
    if (!rc0) rc0 = fn(0, 1, arg);
    if (rc0) device_error(what);
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.pending > 0) pthread_cond_wait(&g_pool.done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
    for (int d = 1; d < n; ++d)
        if (g_pool.rc[d]) {
            g_host_error[0] = '\0';
            fprintf(stderr, "[storm_hip] %s (device slot %d): %s\n", what, d, g_pool.err[d]);
            if (!rc0) rc0 = g_pool.rc[d];
        }
    return rc0;
}

int STORM_hip_comm_init(const uint8_t id[128]) {
    if (!id) return -1;
    if (g_comm) {
        host_error("STORM_hip_comm_init: a communicator is already attached (STORM_hip_comm_finalize first)");
        return -1;
    }
    if (tl_view_count) {
        host_error("STORM_hip_comm_init: refused on a thread whose view is narrowed (STORM_hip_set_thread_devices(0, 0) first)");
        return -1;
    }
    storm_hip_ctx_t* ctx = device_ctx(0);
    if (!ctx || storm_hip_comm_init_rank(ctx, id, g_shard_rank, g_shard_count, &g_comm) != STORM_HIP_OK) {
        device_error("STORM_hip_comm_init");
        g_comm = NULL;
        return -1;
    }
    return 0;
}

int STORM_hip_comm_finalize(void) {
    if (g_comm) storm_hip_comm_destroy(g_comm);
    g_comm = NULL;
    return 0;
}

/* The total an entry point returns: this process's partial, summed over the ranks when a communicator is attached.
 * EVERY rank enters the collective, also one whose pass failed (partial == ALL_PAIRS_FAILED: out of memory, a
 * rebuild that failed, ...): two words travel, {partial, failed}, and every rank returns the failure if any rank
 * failed. A rank that returned early instead left the others blocked in ncclAllReduce for good.
 * All failing paths of the all-pairs entry points behind the point where the ranks are known to have taken the
 * same decisions go through here. */
static uint64_t across_ranks(uint64_t partial) {
    if (!g_comm) return partial;
    uint64_t v[2] = {partial == ALL_PAIRS_FAILED ? 0 : partial, partial == ALL_PAIRS_FAILED ? 1u : 0u};
    /* (slot 0's context exists since STORM_hip_comm_init made the communicator on it; the caller's view is the whole
     *  configuration — narrowed views are refused while a communicator is attached — so it holds slot 0's lock) */
    storm_hip_ctx_t* ctx = g_ctx[0] ? g_ctx[0] : device_ctx(0);
    if (!ctx || storm_hip_comm_allreduce_u64s(ctx, g_comm, v, 2) != STORM_HIP_OK) {
        device_error("all-reduce of the shard totals");
        return ALL_PAIRS_FAILED;
    }
    return v[1] ? ALL_PAIRS_FAILED : v[0];
}

/* a STORM_compute_func is only an identity token on the device path (libalgebra.h) */
static int leaf_is_ours(STORM_compute_func f) {
    if (f == NULL || f == STORM_intersect_count_scalar) return 1;
#if defined(__x86_64__)
    if (f == STORM_intersect_count_sse4 || f == STORM_intersect_count_avx2 || f == STORM_intersect_count_avx512) return 1;
#endif
    return 0;
}
static int lleaf_is_ours(STORM_compute_lfunc f) {
    return f == NULL || f == (STORM_compute_lfunc)STORM_intersect_count_scalar_list;
}

/* ------------------------------------------------------------------------------------------
 * libalgebra surface
 * ---------------------------------------------------------------------------------------- */
uint64_t STORM_intersect_count_scalar(const uint64_t* STORM_RESTRICT b1,
                                      const uint64_t* STORM_RESTRICT b2, const size_t n) {
    uint64_t total = 0;
    for (size_t k = 0; k < n; ++k) total += (uint64_t)__builtin_popcountll(b1[k] & b2[k]);
    return total;
}

uint64_t STORM_intersect_count_scalar_list(const uint64_t* STORM_RESTRICT b1,
                                           const uint64_t* STORM_RESTRICT b2,
                                           const uint32_t* STORM_RESTRICT l1,
                                           const uint32_t* STORM_RESTRICT l2, const size_t n1,
                                           const size_t n2) {
    return STORM_intersect_bitmaps_scalar_list(b1, b2, l1, l2, (uint32_t)n1, (uint32_t)n2);
}

STORM_compute_func STORM_get_intersect_count_func(const size_t n_bitmaps_vector) {
    (void)n_bitmaps_vector;
    return STORM_intersect_count_scalar;
}

uint32_t STORM_get_alignment(void) { return 64; }

void* STORM_aligned_malloc(size_t alignment, size_t size) {
    void* p = NULL;
    if (alignment < sizeof(void*)) alignment = sizeof(void*);
    if (size == 0) size = alignment;
    return posix_memalign(&p, alignment, size) == 0 ? p : NULL;
}

void STORM_aligned_free(void* memblock) { free(memblock); }

int STORM_get_cpuid(void) {
    int bits = 0;
    __builtin_cpu_init();
    if (__builtin_cpu_supports("sse4.2")) bits |= STORM_CPUID_runtime_bit_SSE42;
    if (__builtin_cpu_supports("avx2")) bits |= STORM_CPUID_runtime_bit_AVX2;
    if (__builtin_cpu_supports("avx512bw")) bits |= STORM_CPUID_runtime_bit_AVX512BW;
    if (storm_hip_device_count() > 0) bits |= STORM_CPUID_runtime_bit_GFX950;
    return bits;
}

/* ------------------------------------------------------------------------------------------
 * one-pair list helpers (reference storm.c:4-129)
 * ---------------------------------------------------------------------------------------- */
uint64_t STORM_intersect_vector16_cardinality(const uint16_t* STORM_RESTRICT v1,
                                              const uint16_t* STORM_RESTRICT v2,
                                              const uint32_t len1, const uint32_t len2) {
    uint64_t common = 0;
    uint32_t i = 0, j = 0;
    while (i < len1 && j < len2) {
        if (v1[i] < v2[j]) ++i;
        else if (v2[j] < v1[i]) ++j;
        else { ++common; ++i; ++j; }
    }
    return common;
}

uint64_t STORM_intersect_vector32_unsafe(const uint32_t* STORM_RESTRICT v1,
                                         const uint32_t* STORM_RESTRICT v2, const uint32_t len1,
                                         const uint32_t len2, uint32_t* STORM_RESTRICT out) {
    if (!out || !v1 || !v2 || !len1 || !len2) return 0; /* storm.c:81-84 */
    uint64_t n = 0;
    uint32_t i = 0, j = 0;
    while (i < len1 && j < len2) {
        if (v1[i] < v2[j]) ++i;
        else if (v2[j] < v1[i]) ++j;
        else { out[n++] = i++; out[n++] = j++; }
    }
    return n; /* 2 x matches: interleaved (index in v1, index in v2) */
}

static inline uint64_t probe(const uint64_t* words, uint32_t pos) {
    return (words[pos >> 6] >> (pos & 63u)) & 1u;
}

uint64_t STORM_intersect_bitmaps_scalar_list(const uint64_t* STORM_RESTRICT b1,
                                             const uint64_t* STORM_RESTRICT b2,
                                             const uint32_t* l1, const uint32_t* l2,
                                             const uint32_t n1, const uint32_t n2) {
    uint64_t count = 0; /* the shorter list probes the other row (storm.c:116-126) */
    if (n1 < n2) {
        for (uint32_t k = 0; k < n1; ++k) count += probe(b2, l1[k]);
    } else {
        for (uint32_t k = 0; k < n2; ++k) count += probe(b1, l2[k]);
    }
    return count;
}

/* ------------------------------------------------------------------------------------------
 * dense all-pairs on the device
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    storm_hip_matrix_t* m[MAX_DEVICES];
    uint32_t config_generation; /* device configuration these replicas were made for */
} dense_state_t;

static void dense_state_release(dense_state_t* st) {
    if (!st) return;
    for (int d = 0; d < MAX_DEVICES; ++d) {
        if (st->m[d]) storm_hip_matrix_destroy(g_ctx[d], st->m[d]);
        st->m[d] = NULL;
    }
}

/* all configured devices work concurrently on disjoint shards (one host thread each); the host adds the partials */
typedef struct {
    dense_state_t* st;
    uint64_t part[MAX_DEVICES];
    int first, count; /* the caller's device slots (worker threads do not share its thread-local view) */
} dense_job_t;

static int dense_job(int d, int phase, void* arg) {
    dense_job_t* j = (dense_job_t*)arg;
    const uint32_t world = g_shard_count * (uint32_t)j->count;
    const uint32_t rank = g_shard_rank * (uint32_t)j->count + (uint32_t)(d - j->first);
    return phase == 0 ? storm_hip_pairw_dense_begin(g_ctx[d], j->st->m[d], rank, world)
                      : storm_hip_pairw_dense_end(g_ctx[d], &j->part[d]);
}

static uint64_t dense_state_pairw(dense_state_t* st) {
    dense_job_t j;
    j.st = st;
    j.first = V0;
    j.count = VN;
    memset(j.part, 0, sizeof(j.part));
    if (run_on_devices(dense_job, &j, "all-pairs pass (dense)")) return across_ranks(ALL_PAIRS_FAILED);
    uint64_t total = 0;
    for (int d = V0; d < V1; ++d) total += j.part[d];
    return across_ranks(total);
}

/* The raw-buffer wrappers (STORM_wrapper_*) get the caller's matrix anew on every call; what can be
 * kept between calls is the device allocation: one replica set per process, resized and re-uploaded
 * (a hipMalloc + zero fill + hipFree of the matrix per call cost more than the copy at small sizes). */
static dense_state_t g_wrapper_state;
static uint32_t g_wrapper_words = 0;
/* ONE pass at a time per process: the device contexts behind every handle are shared (stream, partial-sum slots,
 * item tables), and so are the wrappers' cached matrices. Every entry point that touches a device takes this lock
 * (recursive: a STORM_contiguous_t's list mirror is a STORM_t of its own); concurrent callers run one after the other. */
static pthread_mutex_t g_slot_mu[MAX_DEVICES];
static pthread_once_t g_slot_mu_once = PTHREAD_ONCE_INIT;
static void device_lock_init(void) {
    pthread_mutexattr_t at;
    pthread_mutexattr_init(&at);
    pthread_mutexattr_settype(&at, PTHREAD_MUTEX_RECURSIVE);
    for (int d = 0; d < MAX_DEVICES; ++d) pthread_mutex_init(&g_slot_mu[d], &at);
    pthread_mutexattr_destroy(&at);
}
/* One lock PER DEVICE SLOT (round 4; rounds 2 - 3: one for the process): a call locks the slots of its thread's view,
 * in ascending order, so caller threads that drive distinct slots (STORM_hip_set_thread_devices) run side by side.
 * The raw-buffer wrappers and everything that reconfigures the devices take all of them. */
static void device_lock_range(int lo, int hi) {
    pthread_once(&g_slot_mu_once, device_lock_init);
    for (int d = lo; d < hi; ++d) pthread_mutex_lock(&g_slot_mu[d]);
}
static void device_unlock_range(int lo, int hi) {
    for (int d = hi - 1; d >= lo; --d) pthread_mutex_unlock(&g_slot_mu[d]);
}
static void device_lock(void) {
    configure_from_env();
    device_lock_range(V0, V1);
}
static void device_unlock(void) { device_unlock_range(V0, V1); }
/* the raw-buffer wrappers keep ONE set of device matrices per process, over the whole configuration: they lock every
 * slot and see every slot, whatever the calling thread's view (saved in *saved, restored by device_unlock_all) */
typedef struct { int first, count; } saved_view_t;
static void device_lock_all(saved_view_t* saved) {
    pthread_once(&g_slot_mu_once, device_lock_init);
    device_lock_range(0, MAX_DEVICES);
    saved->first = tl_view_first;
    saved->count = tl_view_count;
    tl_view_first = tl_view_count = 0;
}
static void device_unlock_all(const saved_view_t* saved) {
    tl_view_first = saved->first;
    tl_view_count = saved->count;
    device_unlock_range(0, MAX_DEVICES);
}

int STORM_hip_set_option(const char* key, int64_t value) {
    if (!key) return -1;
    saved_view_t sv;
    device_lock_all(&sv);
    read_env_options();
    /* validated before it is remembered or applied anywhere: an unknown key or a value out of range leaves no trace
     * (it used to be remembered when no context existed yet, and applied to some of the contexts when some did) */
    int rc = 0;
    if (storm_hip_option_check(key, value) != STORM_HIP_OK) {
        device_error("STORM_hip_set_option");
        rc = -1;
    } else if (remember_host_option(key, value)) {
        host_error("STORM_hip_set_option: more options than the library remembers for contexts still to be made");
        rc = -1;
    }
    for (int d = 0; d < MAX_DEVICES && !rc; ++d)
        if (g_ctx[d] && storm_hip_ctx_set_option(g_ctx[d], key, value) != STORM_HIP_OK) {
            device_error("STORM_hip_set_option");
            rc = -1;
        }
    device_unlock_all(&sv);
    return rc;
}

static uint64_t raw_pairw_locked(uint32_t n_vectors, const uint64_t* vals, uint32_t n_ints);

static uint64_t raw_pairw(uint32_t n_vectors, const uint64_t* vals, uint32_t n_ints) {
    if (n_vectors < 2 || n_ints == 0) return 0;
    if (!vals) {
        host_error("all-pairs wrapper: NULL buffer");
        return ALL_PAIRS_FAILED;
    }
    saved_view_t sv;
    device_lock_all(&sv);
    const uint64_t total = raw_pairw_locked(n_vectors, vals, n_ints);
    device_unlock_all(&sv);
    return total;
}

static uint64_t raw_pairw_locked(uint32_t n_vectors, const uint64_t* vals, uint32_t n_ints) {
    configure_from_env();
    dense_state_t* st = &g_wrapper_state;
    if (st->config_generation != g_config_generation || g_wrapper_words != n_ints) {
        dense_state_release(st);
        st->config_generation = g_config_generation;
        g_wrapper_words = n_ints;
    }
    if (g_n_devices == 1 && g_shard_count == 1 && !g_comm) {
        /* one device, the whole pair space: the rows travel in panels while the panels before are multiplied */
        storm_hip_ctx_t* ctx = device_ctx(0);
        uint64_t total = 0;
        if (!ctx ||
            (!st->m[0] && storm_hip_matrix_create(ctx, n_vectors, n_ints, &st->m[0]) != STORM_HIP_OK) ||
            storm_hip_matrix_resize(ctx, st->m[0], n_vectors) != STORM_HIP_OK ||
            storm_hip_pairw_dense_upload(ctx, st->m[0], vals, n_ints, &total) != STORM_HIP_OK) {
            device_error("all-pairs wrapper: upload + pass");
            dense_state_release(st);
            return ALL_PAIRS_FAILED;
        }
        return total;
    }
    for (int d = 0; d < g_n_devices; ++d) {
        storm_hip_ctx_t* ctx = device_ctx(d);
        if (!ctx ||
            (!st->m[d] && storm_hip_matrix_create(ctx, n_vectors, n_ints, &st->m[d]) != STORM_HIP_OK) ||
            storm_hip_matrix_resize(ctx, st->m[d], n_vectors) != STORM_HIP_OK ||
            storm_hip_matrix_upload(ctx, st->m[d], 0, n_vectors, vals, n_ints) != STORM_HIP_OK) {
            device_error("all-pairs wrapper: upload");
            dense_state_release(st);
            return ALL_PAIRS_FAILED;
        }
    }
    return dense_state_pairw(st);
}

/* reference storm.c:132-150 */
uint64_t STORM_wrapper_diag(const uint32_t n_vectors, const uint64_t* vals,
                            const uint32_t n_ints, const STORM_compute_func f) {
    if (!leaf_is_ours(f)) {
        host_error("STORM_wrapper_diag: foreign STORM_compute_func cannot run on the device");
        return ALL_PAIRS_FAILED;
    }
    return raw_pairw(n_vectors, vals, n_ints);
}

/* reference storm.c:222-279 — block_size is a CPU cache hint, ignored */
uint64_t STORM_wrapper_diag_blocked(const uint32_t n_vectors, const uint64_t* vals,
                                    const uint32_t n_ints, const STORM_compute_func f,
                                    uint32_t block_size) {
    (void)block_size;
    return STORM_wrapper_diag(n_vectors, vals, n_ints, f);
}

/* The list wrappers hand every pair with a "sparse" row (n_alts <= cutoff at storm.c:207,
 * < cutoff at :309) to the list leaf, which probes the shorter position list against the other
 * row's bitmap (storm.c:108-129). When every list holds exactly the set bits of its row that is
 * popcount(row_i & row_j) for every pair under either cutoff convention, which is what the
 * device computes from the bitmaps alone. A list that disagrees with its row has no such
 * meaning; the check below (rows the reference could route to the list leaf: n_alts <= cutoff,
 * the wider of the two conventions) makes the call fail loudly instead of answering a
 * different question. O(listed positions + words of the listed rows) on the host. */
static int alt_lists_match_rows(uint32_t n_vectors, const uint64_t* vals, uint32_t n_ints,
                                const uint32_t* n_alts, const uint32_t* alt_positions,
                                const uint32_t* alt_offsets, uint32_t cutoff) {
    if (!vals || !n_alts) return 0;
    for (uint32_t i = 0; i < n_vectors; ++i) {
        if (n_alts[i] > cutoff) continue; /* never reaches the list leaf through this row alone */
        const uint64_t* row = vals + (uint64_t)i * n_ints;
        uint64_t bits = 0;
        for (uint32_t k = 0; k < n_ints; ++k) bits += (uint64_t)__builtin_popcountll(row[k]);
        if (bits != n_alts[i]) return 0;
        if (n_alts[i] == 0) continue;
        if (!alt_positions || !alt_offsets) return 0;
        const uint32_t* l = alt_positions + alt_offsets[i];
        for (uint32_t k = 0; k < n_alts[i]; ++k) {
            if (l[k] >= (uint64_t)n_ints * 64u || !probe(row, l[k])) return 0;
            if (k != 0 && l[k] <= l[k - 1]) return 0; /* sorted, duplicate-free (storm.h:227) */
        }
    }
    return 1;
}

/* reference storm.c:190-219 */
uint64_t STORM_wrapper_diag_list(const uint32_t n_vectors, const uint64_t* STORM_RESTRICT vals,
                                 const uint32_t n_ints, const uint32_t* STORM_RESTRICT n_alts,
                                 const uint32_t* STORM_RESTRICT alt_positions,
                                 const uint32_t* STORM_RESTRICT alt_offsets,
                                 const STORM_compute_func f, const STORM_compute_lfunc fl,
                                 const uint32_t cutoff) {
    if (!leaf_is_ours(f) || !lleaf_is_ours(fl)) {
        host_error("STORM_wrapper_diag_list: foreign leaf cannot run on the device");
        return ALL_PAIRS_FAILED;
    }
    if (n_vectors >= 2 && n_ints != 0 &&
        !alt_lists_match_rows(n_vectors, vals, n_ints, n_alts, alt_positions, alt_offsets, cutoff)) {
        host_error("STORM_wrapper_diag_list: a position list does not describe its bitmap row; "
                   "the device path counts the bitmaps and refuses inconsistent lists");
        return ALL_PAIRS_FAILED;
    }
    return raw_pairw(n_vectors, vals, n_ints);
}

/* reference storm.c:282-369 */
uint64_t STORM_wrapper_diag_list_blocked(const uint32_t n_vectors,
                                         const uint64_t* STORM_RESTRICT vals,
                                         const uint32_t n_ints,
                                         const uint32_t* STORM_RESTRICT n_alts,
                                         const uint32_t* STORM_RESTRICT alt_positions,
                                         const uint32_t* STORM_RESTRICT alt_offsets,
                                         const STORM_compute_func f,
                                         const STORM_compute_lfunc fl, const uint32_t cutoff,
                                         uint32_t block_size) {
    (void)block_size;
    return STORM_wrapper_diag_list(n_vectors, vals, n_ints, n_alts, alt_positions, alt_offsets,
                                   f, fl, cutoff);
}

/* reference storm.c:153-171 (rectangle A x B^T; the reference never resets its second offset, :164-168 — the
 * intended sum over all (row of A, row of B) is computed). The rows of A are dealt to the configured devices in
 * equal parts, B goes to every one of them; the device matrices are kept between calls like the other
 * wrappers' (resized and re-uploaded: the caller's buffers are new every time). */
static dense_state_t g_square_a, g_square_b;
static uint32_t g_square_words = 0;

typedef struct {
    const uint64_t *vals1, *vals2;
    uint32_t n1, n2, n_ints;
    uint64_t part[MAX_DEVICES];
} square_job_t;

static void square_rows(const square_job_t* j, int d, uint32_t* r0, uint32_t* r1) {
    *r0 = (uint32_t)((uint64_t)j->n1 * (uint32_t)d / (uint32_t)g_n_devices);
    *r1 = (uint32_t)((uint64_t)j->n1 * ((uint32_t)d + 1u) / (uint32_t)g_n_devices);
}

static int square_job(int d, int phase, void* arg) {
    square_job_t* j = (square_job_t*)arg;
    uint32_t r0, r1;
    square_rows(j, d, &r0, &r1);
    if (phase == 1 || r1 == r0) return STORM_HIP_OK;
    storm_hip_ctx_t* ctx = g_ctx[d];
    int rc;
    if (!g_square_a.m[d] && (rc = storm_hip_matrix_create(ctx, r1 - r0, j->n_ints, &g_square_a.m[d]))) return rc;
    if (!g_square_b.m[d] && (rc = storm_hip_matrix_create(ctx, j->n2, j->n_ints, &g_square_b.m[d]))) return rc;
    if ((rc = storm_hip_matrix_resize(ctx, g_square_a.m[d], r1 - r0)) ||
        (rc = storm_hip_matrix_resize(ctx, g_square_b.m[d], j->n2)) ||
        (rc = storm_hip_matrix_upload(ctx, g_square_a.m[d], 0, r1 - r0, j->vals1 + (uint64_t)r0 * j->n_ints, j->n_ints)) ||
        (rc = storm_hip_matrix_upload(ctx, g_square_b.m[d], 0, j->n2, j->vals2, j->n_ints)))
        return rc;
    /* (synchronous: the product's own launch + read-back; the devices overlap through their host threads) */
    return storm_hip_square_dense(ctx, g_square_a.m[d], g_square_b.m[d], &j->part[d]);
}

uint64_t STORM_wrapper_square(const uint32_t n_vectors1, const uint64_t* STORM_RESTRICT vals1,
                              const uint32_t n_vectors2, const uint64_t* STORM_RESTRICT vals2,
                              const uint32_t n_ints, const STORM_compute_func f) {
    if (!leaf_is_ours(f)) {
        host_error("STORM_wrapper_square: foreign STORM_compute_func cannot run on the device");
        return ALL_PAIRS_FAILED;
    }
    if (n_vectors1 == 0 || n_vectors2 == 0 || n_ints == 0) return 0;
    if (!vals1 || !vals2) {
        host_error("STORM_wrapper_square: NULL buffer");
        return ALL_PAIRS_FAILED;
    }
    saved_view_t sv;
    device_lock_all(&sv);
    configure_from_env();
    if (g_square_a.config_generation != g_config_generation || g_square_words != n_ints) {
        dense_state_release(&g_square_a);
        dense_state_release(&g_square_b);
        g_square_a.config_generation = g_square_b.config_generation = g_config_generation;
        g_square_words = n_ints;
    }
    for (int d = 0; d < g_n_devices; ++d)
        if (!device_ctx(d)) {
            device_unlock_all(&sv);
            return ALL_PAIRS_FAILED;
        }
    square_job_t j;
    memset(&j, 0, sizeof(j));
    j.vals1 = vals1;
    j.vals2 = vals2;
    j.n1 = n_vectors1;
    j.n2 = n_vectors2;
    j.n_ints = n_ints;
    uint64_t total = 0;
    if (run_on_devices(square_job, &j, "STORM_wrapper_square")) {
        dense_state_release(&g_square_a);
        dense_state_release(&g_square_b);
        total = ALL_PAIRS_FAILED;
    } else {
        for (int d = 0; d < g_n_devices; ++d) total += j.part[d];
    }
    device_unlock_all(&sv);
    return total;
}

static void wrapper_states_release(void) {
    saved_view_t sv;
    device_lock_all(&sv);
    dense_state_release(&g_wrapper_state);
    dense_state_release(&g_square_a);
    dense_state_release(&g_square_b);
    device_unlock_all(&sv);
}

int STORM_hip_shutdown(void) {
    (void)STORM_hip_comm_finalize();
    wrapper_states_release();
    for (int d = 0; d < MAX_DEVICES; ++d) {
        if (g_ctx[d]) storm_hip_ctx_destroy(g_ctx[d]);
        g_ctx[d] = NULL;
    }
    ++g_config_generation; /* device copies cached in handles are rebuilt on their next use */
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * STORM_contiguous_t (reference storm.h:188-200, storm.c:1001-1346)
 * ---------------------------------------------------------------------------------------- */
STORM_contiguous_t* STORM_contig_new(size_t vector_length) {
    STORM_contiguous_t* h = (STORM_contiguous_t*)calloc(1, sizeof(*h));
    if (!h) return NULL; /* storm.c:1003 */
    h->vector_length = vector_length;
    h->n_bitmaps_vector = (uint32_t)((vector_length + 63) / 64);            /* storm.c:1013 */
    h->alignment = STORM_get_alignment();
    h->intsec_func = STORM_get_intersect_count_func(h->n_bitmaps_vector);
    h->scalar_cutoff = (uint32_t)(vector_length / 200 > 200 ? 200 : vector_length / 200); /* :1016 */
    return h;
}

#define CONTIG_STREAM_ROWS 256u /* batch of finished rows STORM_contig_add streams to the device */
static void contig_stream_rows(STORM_contiguous_t* h);
static void contig_stream_rows_locked(STORM_contiguous_t* h);

/* Device-side construction (reference loop being replaced: the bit setting of STORM_contig_add,
 * storm.c:1103-1115, as far as the DEVICE copy of a row goes; the host bitmap `data` is public and stays).
 * A row reaches its device mirror as sorted positions (4 bytes each, OR-ed into the zeroed row by
 * set_bits_kernel) instead of as its words when that is at most an eighth of the bytes: n <= W / 4 for a row
 * of W words. (The bus moves a row's 8 KiB in 0.35 us; what the host spends per position on the way has to stay
 * below that: tools/bench_contig_add, profiles/r03_f_contig_add.txt — at n = W the positions LOSE, 13 against
 * 3.8 ms for 10000 rows.) Rows below scalar_cutoff have their list in `scalar` already; rows from the cutoff up
 * keep a copy of their positions here from the add until the upload. STORM_HIP_ADD_POSITIONS=0 turns it off. */
typedef struct {
    uint64_t row0, n_rows, m_rows; /* rows [row0, row0 + n_rows) of the container, appended in order */
    uint64_t* off;                 /* n_rows + 1 starts in pos; an empty range = the row goes as words */
    uint32_t* pos;
    uint64_t n_pos, m_pos;
    int broken;           /* an allocation failed: nothing more is kept */
} contig_pending_t;

static int contig_positions_enabled(void) { /* (adders of different handles may run on different threads) */
    static int enabled = -1;
    int v = __atomic_load_n(&enabled, __ATOMIC_RELAXED);
    if (v < 0) {
        const char* e = getenv("STORM_HIP_ADD_POSITIONS");
        v = !(e && e[0] == '0');
        __atomic_store_n(&enabled, v, __ATOMIC_RELAXED);
    }
    return v;
}
static void contig_pending_free(STORM_contiguous_t* h) {
    contig_pending_t* p = (contig_pending_t*)h->hip_pending;
    if (!p) return;
    free(p->off);
    free(p->pos);
    free(p);
    h->hip_pending = NULL;
}
/* row h->n_data (about to be counted) has `distinct` positions, first occurrences of values[0 .. n_values) */
static void contig_pending_stop(contig_pending_t* p) {
    p->broken = 1;
    p->n_rows = 0;
    p->n_pos = 0;
}
static void contig_pending_note(STORM_contiguous_t* h, const uint32_t* values, uint32_t n_values,
                                uint32_t distinct) {
    if (!contig_positions_enabled()) return;
    contig_pending_t* p = (contig_pending_t*)h->hip_pending;
    if (p && p->broken) return;
    if (p && p->row0 + p->n_rows != h->n_data) { /* a gap: start over behind it */
        p->row0 = h->n_data;
        p->n_rows = 0;
        p->n_pos = 0;
    }
    const int keep = distinct >= h->scalar_cutoff && (uint64_t)distinct * 4 <= h->n_bitmaps_vector;
    if (!p) {
        if (!keep) return; /* nothing to remember yet: rows before the first kept one are not covered */
        p = (contig_pending_t*)calloc(1, sizeof(*p));
        if (!p) return;
        p->row0 = h->n_data;
        h->hip_pending = p;
    }
    if (p->n_rows + 2 > p->m_rows) {
        const uint64_t m = p->m_rows ? 2 * p->m_rows : 1024;
        uint64_t* no = (uint64_t*)realloc(p->off, m * sizeof(uint64_t));
        if (!no) { contig_pending_stop(p); return; }
        p->off = no;
        p->m_rows = m;
    }
    if (keep && p->n_pos + distinct > p->m_pos) {
        const uint64_t m = 2 * p->m_pos + distinct + 65536;
        uint32_t* np = (uint32_t*)realloc(p->pos, m * sizeof(uint32_t));
        if (!np) { contig_pending_stop(p); return; }
        p->pos = np;
        p->m_pos = m;
    }
    p->off[p->n_rows] = p->n_pos;
    if (keep)
        for (uint32_t k = 0; k < n_values; ++k)
            if (k == 0 || values[k] != values[k - 1]) p->pos[p->n_pos++] = values[k];
    p->off[++p->n_rows] = p->n_pos;
}
/* positions of row r if it may travel as positions: from `scalar` or from the pending copy; NULL = as words */
static const uint32_t* contig_row_positions(const STORM_contiguous_t* h, uint64_t r, uint32_t* n) {
    if (!contig_positions_enabled()) return NULL;
    const contig_pending_t* p = (const contig_pending_t*)h->hip_pending;
    if (r < h->hip_words_below) return NULL; /* edited in place (STORM_contig_hip_invalidate): as the words they now are */
    if (h->n_scalar[r] < h->scalar_cutoff && h->scalar) {
        if ((uint64_t)h->n_scalar[r] * 4 > h->n_bitmaps_vector) return NULL;
        *n = h->n_scalar[r];
        return h->scalar + h->scalar_offset[r];
    }
    if (p && r >= p->row0 && r < p->row0 + p->n_rows && p->off[r - p->row0 + 1] > p->off[r - p->row0]) {
        *n = (uint32_t)(p->off[r - p->row0 + 1] - p->off[r - p->row0]);
        return p->pos + p->off[r - p->row0];
    }
    return NULL;
}

static void contig_drop_device(STORM_contiguous_t* h) {
    if (h->hip_matrix) {
        dense_state_release((dense_state_t*)h->hip_matrix);
        free(h->hip_matrix);
        h->hip_matrix = NULL;
    }
    h->hip_rows_synced = 0;
}

/* A container whose rows are ALL sparse is mirrored row by row into a private STORM_t: its all-pairs total then
 * costs work proportional to the listed positions (list-probe kernel K4) instead of a dense pass over N x M bits.
 * The reference diverts to its list variants when rows are below scalar_cutoff (<= 200 positions, storm.c:1151-1162);
 * here "sparse" is a row of at most M / 16 positions — 4096 per 65536-bit block on average, the density up to which
 * a STORM_t keeps blocks as lists and K4 beats the matrix cores (M = 65536, N = 10000, tools/storm_benchmark: 0.41 ms
 * against 0.84 at 2621 positions per row, 0.08 against 0.83 at 655, 0.03 at 5; level from 6553 up, where the rows
 * are bitmaps either way). The first denser row, a failed allocation or STORM_contig_hip_invalidate ends it for this
 * container (until STORM_contig_clear); the dense mirror is then brought up to date on demand.
 * STORM_HIP_CONTIG_LISTS=0 in the environment turns it off. */
static int contig_row_is_sparse(const STORM_contiguous_t* h, uint32_t distinct) {
    return (uint64_t)distinct * 16u <= h->vector_length || distinct < h->scalar_cutoff;
}
static void contig_lists_end(STORM_contiguous_t* h) {
    if (h->hip_lists) STORM_free(h->hip_lists);
    h->hip_lists = NULL;
    h->hip_lists_off = 1;
}
static int contig_lists_enabled(void) {
    static int enabled = -1;
    int v = __atomic_load_n(&enabled, __ATOMIC_RELAXED);
    if (v < 0) {
        const char* e = getenv("STORM_HIP_CONTIG_LISTS");
        v = !(e && e[0] == '0');
        __atomic_store_n(&enabled, v, __ATOMIC_RELAXED);
    }
    return v;
}

void STORM_contig_free(STORM_contiguous_t* h) {
    if (!h) return;
    contig_drop_device(h);
    contig_pending_free(h);
    if (h->hip_lists) STORM_free(h->hip_lists);
    STORM_aligned_free(h->data);
    STORM_aligned_free(h->scalar);
    STORM_aligned_free(h->n_scalar);
    free(h->bitmaps);
    free(h->scalar_offset);
    free(h); /* the reference leaks the handle (storm.c:1020-1029); callers never free it */
}

/* Row capacity; returns 0 on success. The reference grows by 512 rows at a time (storm.c:1046, :1082) and copies
 * every row each time: 10000 rows of 8 KiB are copied 20 times, 0.8 GB, and that copy was most of what
 * STORM_contig_add cost (tools/bench_contig_add: 183 ms for the adds, 4 ms for the upload). Here the capacity
 * grows by half (in steps of 512 rows, 512 to begin with), and only the new rows are cleared. */
static int contig_reserve_rows(STORM_contiguous_t* h) {
    if (h->data && h->n_data < h->m_data) return 0;
    const uint64_t half = (h->m_data / 2 + 511) / 512 * 512;
    const uint64_t new_m = h->m_data + (half > 512 ? half : 512);
    const size_t W = h->n_bitmaps_vector;
    uint64_t* nd = (uint64_t*)STORM_aligned_malloc(h->alignment, new_m * W * sizeof(uint64_t));
    uint32_t* nn = (uint32_t*)STORM_aligned_malloc(h->alignment, new_m * sizeof(uint32_t));
    STORM_contiguous_bitmap_t* nb =
        (STORM_contiguous_bitmap_t*)realloc(h->bitmaps, new_m * sizeof(*nb));
    uint64_t* no = (uint64_t*)realloc(h->scalar_offset, new_m * sizeof(uint64_t));
    if (nb) h->bitmaps = nb;
    if (no) h->scalar_offset = no;
    if (!nd || !nn || !nb || !no) {
        STORM_aligned_free(nd);
        STORM_aligned_free(nn);
        return -1;
    }
    if (h->data) memcpy(nd, h->data, h->n_data * W * sizeof(uint64_t));
    memset(nd + h->n_data * W, 0, (new_m - h->n_data) * W * sizeof(uint64_t));
    if (h->n_scalar) memcpy(nn, h->n_scalar, h->n_data * sizeof(uint32_t));
    STORM_aligned_free(h->data);
    STORM_aligned_free(h->n_scalar);
    h->data = nd;
    h->n_scalar = nn;
    h->m_data = new_m;
    return 0;
}

/* the per-row views (storm.h:181-186) always follow the current buffers */
static void contig_rebuild_views(STORM_contiguous_t* h) {
    for (uint64_t i = 0; i < h->m_data; ++i) {
        h->bitmaps[i].data = h->data + i * h->n_bitmaps_vector;
        if (i < h->n_data) {
            h->bitmaps[i].n_scalar = h->n_scalar[i];
            h->bitmaps[i].scalar = h->scalar + h->scalar_offset[i];
        } else {
            h->bitmaps[i].n_scalar = 0;
            h->bitmaps[i].scalar = NULL;
        }
    }
}

int STORM_contig_add(STORM_contiguous_t* h, const uint32_t* values, const uint32_t n_values) {
    if (!h) return -1;
    if (!values) return -2;
    if (n_values == 0) return 0; /* no row appended, storm.c:1034 */

    int views_stale = 0;
    if (!h->scalar) { /* storm.c:1037-1041 */
        h->m_scalar = 512 * 32;
        h->tot_scalar = 0;
        h->scalar = (uint32_t*)STORM_aligned_malloc(h->alignment, h->m_scalar * sizeof(uint32_t));
        if (!h->scalar) return -3;
        views_stale = 1;
    }
    if (!h->data || h->n_data >= h->m_data) {
        if (contig_reserve_rows(h)) return -3;
        views_stale = 1;
    }
    if (h->tot_scalar + n_values >= h->m_scalar) { /* storm.c:1060-1073 */
        const uint64_t add = (uint64_t)5 * n_values < 65535 ? 65535 : (uint64_t)5 * n_values;
        uint32_t* ns =
            (uint32_t*)STORM_aligned_malloc(h->alignment, (h->m_scalar + add) * sizeof(uint32_t));
        if (!ns) return -3;
        memcpy(ns, h->scalar, h->tot_scalar * sizeof(uint32_t));
        STORM_aligned_free(h->scalar);
        h->scalar = ns;
        h->m_scalar += add;
        views_stale = 1;
    }
    if (views_stale) contig_rebuild_views(h);

    uint64_t* row = h->data + h->n_data * h->n_bitmaps_vector;
    uint32_t distinct = 0;
    for (uint32_t k = 0; k < n_values; ++k) { /* storm.c:1103-1115 */
        if (k != 0 && values[k] == values[k - 1]) continue;
        assert(k == 0 || values[k] > values[k - 1]);
        assert(values[k] < h->n_bitmaps_vector * 64ull);
        row[values[k] >> 6] |= 1ULL << (values[k] & 63u);
        ++distinct;
    }
    h->scalar_offset[h->n_data] = h->tot_scalar;
    h->bitmaps[h->n_data].scalar = h->scalar + h->tot_scalar;
    if (distinct < h->scalar_cutoff) { /* storm.c:1119-1129; list kept compact */
        uint32_t* dst = h->scalar + h->tot_scalar;
        uint32_t w = 0;
        for (uint32_t k = 0; k < n_values; ++k) {
            if (k != 0 && values[k] == values[k - 1]) continue;
            dst[w++] = values[k];
        }
        h->tot_scalar += distinct;
    }
    h->n_scalar[h->n_data] = distinct; /* storm.c:1132-1134 */
    h->bitmaps[h->n_data].n_scalar = distinct;
    if (!h->hip_lists_off) { /* list mirror: only while every row is sparse */
        if (!contig_row_is_sparse(h, distinct) || !contig_lists_enabled()) {
            contig_lists_end(h);
        } else {
            if (!h->hip_lists && (h->hip_lists = STORM_new())) h->hip_lists->hip_private = 1;
            if (!h->hip_lists || STORM_add(h->hip_lists, values, n_values) < 0) contig_lists_end(h);
        }
    }
    contig_pending_note(h, values, n_values, distinct);
    ++h->n_data;
    if (h->n_data % CONTIG_STREAM_ROWS == 0) contig_stream_rows(h);
    return (int)n_values;
}

int STORM_contig_clear(STORM_contiguous_t* h) { /* storm.c:1139-1147 */
    if (!h) return -1;
    if (!h->data) return 0;
    memset(h->data, 0, (size_t)h->n_bitmaps_vector * h->m_data * sizeof(uint64_t));
    h->n_data = 0;
    h->tot_scalar = 0;
    contig_drop_device(h);
    contig_pending_free(h);
    if (h->hip_lists) STORM_clear(h->hip_lists);
    h->hip_lists_off = 0;
    h->hip_words_below = 0;
    return 1;
}

/* where row r's positions live: 0 = it travels as words, 1 = in `scalar`, 2 = in the pending copy */
static int contig_row_source(const STORM_contiguous_t* h, uint64_t r) {
    uint32_t n = 0;
    const uint32_t* src = contig_row_positions(h, r, &n);
    if (!src) return 0;
    return (h->scalar && src >= h->scalar && src < h->scalar + h->m_scalar) ? 1 : 2;
}
/* rows [r0, r1) to one replica: runs of rows that travel as positions through set_bits_kernel (the rows of a
 * mirror beyond its synced rows are zero), the runs between them as words. The positions of consecutive rows
 * are consecutive in their source (`scalar` or the pending copy), so a run is handed over where it lies.
 * 0 on success. */
static int contig_send_rows(STORM_contiguous_t* h, storm_hip_ctx_t* ctx, storm_hip_matrix_t* m, uint64_t r0,
                            uint64_t r1) {
    const uint64_t W = h->n_bitmaps_vector;
    const contig_pending_t* p = (const contig_pending_t*)h->hip_pending;
    uint64_t* off = NULL;
    uint64_t m_off = 0;
    int rc = 0;
    uint64_t r = r0;
    while (r < r1 && rc == 0) {
        const int src = contig_row_source(h, r);
        uint64_t e = r + 1;
        while (e < r1 && contig_row_source(h, e) == src) ++e;
        if (src == 2) {
            if (storm_hip_matrix_set_rows_from_positions(ctx, m, r, e - r, p->off + (r - p->row0), p->pos) !=
                STORM_HIP_OK)
                rc = -1;
        } else if (src == 1) {
            if (e - r + 1 > m_off) {
                free(off);
                m_off = e - r + 1;
                off = (uint64_t*)malloc(m_off * sizeof(uint64_t));
            }
            if (!off) { /* out of host memory: the run goes as words */
                m_off = 0;
                if (storm_hip_matrix_upload(ctx, m, r, e - r, h->data + r * W, W) != STORM_HIP_OK) rc = -1;
            } else {
                for (uint64_t i = r; i < e; ++i) off[i - r] = h->scalar_offset[i];
                off[e - r] = h->scalar_offset[e - 1] + h->n_scalar[e - 1];
                if (storm_hip_matrix_set_rows_from_positions(ctx, m, r, e - r, off, h->scalar) != STORM_HIP_OK)
                    rc = -1;
            }
        } else if (storm_hip_matrix_upload(ctx, m, r, e - r, h->data + r * W, W) != STORM_HIP_OK) {
            rc = -1;
        }
        r = e;
    }
    free(off);
    return rc;
}

/* Rows [hip_rows_synced, upto) of `h` go to every replica of its device mirror (created on first
 * use, grown with storm_hip_matrix_resize). Rows of a STORM_contiguous_t never change once added,
 * so what has been uploaded stays valid. Returns 0 on success. */
static int contig_upload_rows(STORM_contiguous_t* h, uint64_t upto) {
    configure_from_env();
    dense_state_t* st = (dense_state_t*)h->hip_matrix;
    if (st && st->config_generation != VIEW_GENERATION) { /* other devices, or another thread's slots */
        contig_drop_device(h);
        st = NULL;
    }
    if (!st) {
        st = (dense_state_t*)calloc(1, sizeof(*st));
        if (!st) return -1;
        st->config_generation = VIEW_GENERATION;
        h->hip_matrix = st;
        h->hip_rows_synced = 0;
    }
    for (int d = V0; d < V1; ++d) {
        storm_hip_ctx_t* ctx = device_ctx(d);
        if (!ctx) return -1;
        if (!st->m[d] &&
            storm_hip_matrix_create(ctx, h->m_data, h->n_bitmaps_vector, &st->m[d]) != STORM_HIP_OK)
            return -1;
        if (storm_hip_matrix_resize(ctx, st->m[d], upto) != STORM_HIP_OK) return -1;
        if (contig_send_rows(h, ctx, st->m[d], h->hip_rows_synced, upto) != 0) return -1;
    }
    h->hip_rows_synced = upto;
    h->hip_rows_capacity = h->m_data;
    contig_pending_t* p = (contig_pending_t*)h->hip_pending;
    if (p) { /* everything it covered is on the device now */
        p->row0 = upto;
        p->n_rows = 0;
        p->n_pos = 0;
    }
    return 0;
}

/* Streaming: STORM_contig_add sends every finished batch of CONTIG_STREAM_ROWS rows to the device mirror
 * (a synchronous copy from the container's pageable rows: the cost of the upload moves from the first
 * all-pairs call into the adds, it does not overlap the host's bit setting), so that the first all-pairs
 * call only has the last partial batch left to copy — and does not pay for the process's HIP initialisation
 * and context (150 - 250 ms) either: the first finished batch creates the context. The reference's harness
 * times exactly one call per row (benchmark.cpp:605-613, :896-918), construction apart.
 * [r6] On by default. Until round 5 batches travelled only once the process already held a device context
 * (i.e. after its first all-pairs call), because a container builder that initialises the GPU is a trap
 * for a process that builds its containers and THEN forks workers (a process that has initialised HIP must
 * not be forked): such a process sets STORM_HIP_STREAM_ROWS=2 (that behaviour) or =0 (never); =1: the default.
 * Best effort: a failure here surfaces at the all-pairs call, as before. */
static int g_stream_state = 0; /* 0 unknown, 1 on from the first batch, 2 on once a context exists, -1 off */

static void contig_stream_rows(STORM_contiguous_t* h) {
    device_lock();
    contig_stream_rows_locked(h);
    device_unlock();
}
/* whether a builder may send rows / blocks to the device now (the caller holds its device slots) */
static int stream_wanted_locked(void) {
    /* (threads on different device slots get here side by side: the switch is read and written atomically) */
    int stream_state = __atomic_load_n(&g_stream_state, __ATOMIC_RELAXED);
    if (stream_state == 0) {
        const char* e = getenv("STORM_HIP_STREAM_ROWS");
        stream_state = (e && e[0] == '0') ? -1 : (e && e[0] == '2') ? 2 : 1;
        __atomic_store_n(&g_stream_state, stream_state, __ATOMIC_RELAXED);
    }
    if (stream_state < 0) return 0;
    if (stream_state == 2 && !g_ctx[V0]) return 0;
    if (stream_state == 1 && !g_ctx[V0]) { /* the first batch of a process creates the context: once, quietly */
        g_quiet_ctx = 1;
        storm_hip_ctx_t* ctx = device_ctx(V0);
        g_quiet_ctx = 0;
        if (!ctx) {
            __atomic_store_n(&g_stream_state, -1, __ATOMIC_RELAXED);
            return 0;
        }
    }
    return 1;
}
static void contig_stream_rows_locked(STORM_contiguous_t* h) {
    if (!stream_wanted_locked()) return;
    /* a container that is all lists so far needs no dense mirror (N x M bits over PCIe for a handful of
     * positions per row): contig_mirror() uploads whatever is missing the day a dense row or the per-pair
     * matrix asks for it */
    if (h->hip_lists && !h->hip_lists_off) return;
    g_quiet_ctx = 1;
    const int rc = contig_upload_rows(h, h->n_data);
    g_quiet_ctx = 0;
    if (rc != 0) {
        contig_drop_device(h);
        __atomic_store_n(&g_stream_state, -1, __ATOMIC_RELAXED);
    }
}

/* make sure the device mirror of `h` is current; NULL on failure */
static dense_state_t* contig_mirror(STORM_contiguous_t* h) {
    dense_state_t* st = (dense_state_t*)h->hip_matrix;
    if (!st || h->hip_rows_synced != h->n_data || st->config_generation != VIEW_GENERATION) {
        if (h->hip_rows_synced > h->n_data) contig_drop_device(h);
        if (contig_upload_rows(h, h->n_data) != 0) {
            device_error("dense upload");
            contig_drop_device(h);
            return NULL;
        }
    }
    return (dense_state_t*)h->hip_matrix;
}

/* the device mirror is rebuilt whenever rows were added since the last all-pairs call */
static uint64_t contig_pairw_device_locked(STORM_contiguous_t* h);
static uint64_t contig_pairw_device(STORM_contiguous_t* h) {
    device_lock();
    const uint64_t total = contig_pairw_device_locked(h);
    device_unlock();
    return total;
}
static uint64_t contig_pairw_device_locked(STORM_contiguous_t* h) {
    if (h->n_data < 2) return 0;
    if (h->hip_lists && !h->hip_lists_off && h->hip_lists->n_conts == h->n_data) {
        /* The list mirror pays while every block column can go to the probe kernel (at most 65535 rows per
         * column: its work is then the listed positions, and lists-only columns hold no 8 KiB pool rows).
         * Beyond that, or when the mirror's arena or its pass fails, the container goes back to its dense
         * mirror for good — N x M bits, which is what the caller allocated anyway. */
        if (h->n_data <= 65535) {
            const uint64_t total = STORM_pairw_intersect_cardinality(h->hip_lists);
            if (total != ALL_PAIRS_FAILED) return total;
        }
        contig_lists_end(h);
    }
    dense_state_t* st = contig_mirror(h);
    return st ? dense_state_pairw(st) : across_ranks(ALL_PAIRS_FAILED);
}

uint64_t STORM_contig_n_rows(const STORM_contiguous_t* h) { return h ? h->n_data : 0; }

/* Extension (storm.h): the per-pair matrix the reference only sums (README.md:41). `op`:
 * 0 intersect, 1 union, 2 symmetric difference. `out` holds out_rows x out_ld uint32, row-major;
 * the n_data x n_data result is written at leading dimension out_ld, entries i >= j are 0.
 * Returns 0, -1 for a NULL handle, -2 for NULL out, -3 on device failure, -4 when the buffer is
 * too small for the rows the handle holds (out_rows < n_data or out_ld < n_data).
 * With several devices configured every GPU writes one band of rows; the bands are cut so that
 * each holds the same number of pairs (row i has n - 1 - i of them). */
static int contig_pairw_matrix_locked(STORM_contiguous_t* h, int op, uint32_t* out, uint64_t out_rows, uint64_t out_ld);
int STORM_contig_pairw_matrix(STORM_contiguous_t* h, int op, uint32_t* out, uint64_t out_rows,
                              uint64_t out_ld) {
    device_lock();
    const int rc = contig_pairw_matrix_locked(h, op, out, out_rows, out_ld);
    device_unlock();
    return rc;
}
/* The strict upper triangle of an n-row device matrix (one replica per slot of the caller) into host memory: every
 * device writes one band of rows, the bands cut so that each holds about the same number of pairs. 0 or -3. */
static int pairw_matrix_bands(storm_hip_matrix_t* const* m, uint64_t n, int op, uint32_t* out, uint64_t out_ld) {
    const uint64_t pairs = n * (n - 1) / 2;
    uint64_t row0 = 0;
    int launched = V0, rc = 0;
    for (int d = V0; d < V1; ++d) {
        uint64_t row1 = n;
        if (d + 1 < V1) { /* first row r (multiple of 256) with pairs above r >= share */
            const uint64_t share = pairs / (uint64_t)VN * (uint64_t)(d - V0 + 1);
            row1 = row0;
            while (row1 < n && row1 * (n - 1) - row1 * (row1 - 1) / 2 < share) row1 += 256;
            if (row1 > n) row1 = n;
        }
        if (row1 > row0) {
            if (storm_hip_pairw_matrix_band_begin(g_ctx[d], m[d], op, row0, row1 - row0, out + row0 * out_ld,
                                                  out_ld) != STORM_HIP_OK) {
                device_error("storm_hip_pairw_matrix_band_begin");
                rc = -3;
                break;
            }
        }
        launched = d + 1;
        row0 = row1;
    }
    for (int d = V0; d < launched; ++d)
        if (storm_hip_pairw_matrix_band_end(g_ctx[d]) != STORM_HIP_OK) {
            device_error("storm_hip_pairw_matrix_band_end");
            rc = -3;
        }
    return rc;
}

static int contig_pairw_matrix_locked(STORM_contiguous_t* h, int op, uint32_t* out, uint64_t out_rows,
                                      uint64_t out_ld) {
    if (!h) return -1;
    if (!out) return -2;
    const uint64_t n = h->n_data;
    if (out_rows < n || out_ld < n) return -4;
    if (n == 0) return 0;
    dense_state_t* st = contig_mirror(h);
    if (!st) return -3;
    return pairw_matrix_bands(st->m, n, op, out, out_ld);
}

uint64_t STORM_contig_pairw_intersect_cardinality(STORM_contiguous_t* h) { /* :1149-1173 */
    if (!h) return (uint64_t)-1;
    return contig_pairw_device(h);
}

uint64_t STORM_contig_pairw_intersect_cardinality_blocked(STORM_contiguous_t* h,
                                                          uint32_t bsize) { /* :1175-1241 */
    (void)bsize;
    if (!h) return (uint64_t)-1;
    return contig_pairw_device(h);
}

uint64_t STORM_contig_pairw_intersect_cardinality_list(STORM_contiguous_t* h) { /* :1243-1263 */
    if (!h) return (uint64_t)-1;
    if (!h->scalar) return (uint64_t)-2;
    if (!h->n_scalar) return (uint64_t)-3;
    return contig_pairw_device(h);
}

uint64_t STORM_contig_pairw_intersect_cardinality_blocked_list(STORM_contiguous_t* h,
                                                               uint32_t bsize) { /* :1265-1346 */
    (void)bsize;
    return STORM_contig_pairw_intersect_cardinality_list(h);
}

/* ------------------------------------------------------------------------------------------
 * STORM_bitmap_t — one 65536-bit block (reference storm.c:372-380, :398-656)
 * ---------------------------------------------------------------------------------------- */
/* Every public function that changes a STORM_t, a row or a block (they are public, storm.h:203-222, and a block
 * does not know the container it belongs to) bumps this process-wide epoch. A handle remembers the epoch its
 * device arena was last verified at: while no mutator has run since, an all-pairs call skips the O(blocks)
 * fingerprint walk. Members edited in place, without any of these functions: STORM_hip_invalidate (storm.h).
 * STORM_HIP_ALWAYS_FINGERPRINT=1 walks the fingerprint on every call, as rounds 2 - 3 did. */
static uint64_t g_mutation_epoch = 1;
static inline void storm_mutated(void) { __atomic_add_fetch(&g_mutation_epoch, 1, __ATOMIC_RELAXED); }
static inline uint64_t storm_epoch(void) { return __atomic_load_n(&g_mutation_epoch, __ATOMIC_RELAXED); }

void STORM_bitmap_init(STORM_bitmap_t* b) {
    storm_mutated();
    if (!b) return;
    memset(b, 0, sizeof(*b));
    b->own_data = 1;
    b->own_scalar = 1;
}

STORM_bitmap_t* STORM_bitmap_new() {
    STORM_bitmap_t* b = (STORM_bitmap_t*)STORM_aligned_malloc(64, sizeof(*b));
    STORM_bitmap_init(b);
    return b;
}

static void bitmap_release_buffers(STORM_bitmap_t* b) {
    if (b->own_data) STORM_aligned_free(b->data);
    if (b->own_scalar) STORM_aligned_free(b->scalar);
    b->data = NULL;
    b->scalar = NULL;
}

void STORM_bitmap_free(STORM_bitmap_t* b) {
    storm_mutated();
    if (!b) return;
    bitmap_release_buffers(b);
    STORM_aligned_free(b);
}

static int bitmap_ensure_words(STORM_bitmap_t* b) {
    if (!b->data) {
        b->data = (uint64_t*)STORM_aligned_malloc(STORM_get_alignment(), BLOCK_WORDS * 8);
        if (!b->data) return -1;
        memset(b->data, 0, BLOCK_WORDS * 8);
    }
    b->n_bitmap = BLOCK_WORDS;
    return 0;
}

static int bitmap_ensure_list(STORM_bitmap_t* b, uint32_t extra) {
    if (b->scalar && b->n_scalar + extra <= b->m_scalar) return 0;
    const uint32_t new_m = b->scalar ? b->n_scalar + extra + 1024 : (extra > 256 ? extra + 256 : 256);
    uint16_t* nl = (uint16_t*)STORM_aligned_malloc(STORM_get_alignment(), new_m * sizeof(uint16_t));
    if (!nl) return -1;
    if (b->scalar) memcpy(nl, b->scalar, b->n_scalar * sizeof(uint16_t));
    if (b->own_scalar) STORM_aligned_free(b->scalar);
    b->scalar = nl;
    b->m_scalar = new_m;
    b->own_scalar = 1;
    return 0;
}

/* bitmap kind: reference storm.c:442-465 (-1 / -2 / -3 for NULL handle / NULL values / empty) */
int STORM_bitmap_add(STORM_bitmap_t* b, const uint32_t* values, const uint32_t n_values) {
    storm_mutated();
    if (!b) return -1;
    if (!values) return -2;
    if (n_values == 0) return -3;
    if (bitmap_ensure_words(b)) return -5;
    const uint32_t base = b->id * BLOCK_BITS;
    for (uint32_t k = 0; k < n_values; ++k) {
        const uint32_t v = values[k] - base;
        assert(values[k] >= base && v < BLOCK_BITS);
        b->n_bits_set += (uint32_t)(probe(b->data, v) == 0);
        b->data[v >> 6] |= 1ULL << (v & 63u);
    }
    return (int)n_values;
}

/* both representations: reference storm.c:468-518 (-1 / -3 / -4) */
int STORM_bitmap_add_with_scalar(STORM_bitmap_t* b, const uint32_t* values,
                                 const uint32_t n_values) {
    storm_mutated();
    if (!b) return -1;
    if (!values) return -3;
    if (n_values == 0) return -4;
    if (bitmap_ensure_words(b) || bitmap_ensure_list(b, n_values)) return -5;
    const uint32_t base = b->id * BLOCK_BITS;
    b->n_scalar_set = 1;
    for (uint32_t k = 0; k < n_values; ++k) {
        const uint32_t v = values[k] - base;
        assert(values[k] >= base && v < BLOCK_BITS);
        if (probe(b->data, v) == 0) {
            b->data[v >> 6] |= 1ULL << (v & 63u);
            b->scalar[b->n_scalar] = (uint16_t)v;
            b->n_scalar = b->n_scalar + 1;
            ++b->n_bits_set;
        }
    }
    return (int)n_values;
}

/* list kind: reference storm.c:521-558 (-1 / -3 / -4); equal neighbours stored once so the
 * list stays duplicate-free, which the list intersections require */
int STORM_bitmap_add_scalar_only(STORM_bitmap_t* b, const uint32_t* values,
                                 const uint32_t n_values) {
    storm_mutated();
    if (!b) return -1;
    if (!values) return -3;
    if (n_values == 0) return -4;
    if (bitmap_ensure_list(b, n_values)) return -5;
    const uint32_t base = b->id * BLOCK_BITS;
    b->n_scalar_set = 1;
    for (uint32_t k = 0; k < n_values; ++k) {
        const uint32_t v = values[k] - base;
        assert(values[k] >= base && v < BLOCK_BITS);
        if (b->n_scalar != 0 && b->scalar[b->n_scalar - 1] == (uint16_t)v) continue;
        b->scalar[b->n_scalar] = (uint16_t)v;
        b->n_scalar = b->n_scalar + 1;
        ++b->n_bits_set;
    }
    return (int)n_values;
}

int STORM_bitmap_clear(STORM_bitmap_t* b) { /* storm.c:561-569: buffers are kept */
    storm_mutated();
    if (!b) return -1;
    if (b->data) memset(b->data, 0, sizeof(uint64_t) * BLOCK_WORDS);
    b->n_scalar = 0;
    b->n_bits_set = 0;
    b->n_bitmap = 0;
    return 1;
}

uint32_t STORM_bitmap_serialized_size(STORM_bitmap_t* b) { /* storm.c:372-380 */
    uint32_t bytes = (uint32_t)sizeof(uint64_t) * b->n_bitmap;
    if (b->n_scalar_set) bytes += (uint32_t)sizeof(uint16_t) * b->n_scalar;
    return bytes + 4 * (uint32_t)sizeof(uint32_t);
}

/* one block pair on the host: the 4-way kind dispatch of storm.c:618-656, counting the probed
 * bit itself in the mixed cases (the reference's `a & b != 0` at :636,:644 is defect D1) */
uint64_t STORM_bitmap_intersect_cardinality_func(STORM_bitmap_t* STORM_RESTRICT b1,
                                                 STORM_bitmap_t* STORM_RESTRICT b2,
                                                 const STORM_compute_func func) {
    if (!b1 || !b2 || b1->id != b2->id) return 0;
    const STORM_compute_func f = func ? func : STORM_intersect_count_scalar;
    if (!b1->n_bitmap && !b2->n_bitmap)
        return STORM_intersect_vector16_cardinality(b1->scalar, b2->scalar, b1->n_scalar,
                                                    b2->n_scalar);
    if (b1->n_bitmap && b2->n_bitmap) return f(b1->data, b2->data, b1->n_bitmap);
    const STORM_bitmap_t* dense = b1->n_bitmap ? b1 : b2;
    const STORM_bitmap_t* list = b1->n_bitmap ? b2 : b1;
    uint64_t count = 0;
    for (uint32_t k = 0; k < list->n_scalar; ++k) count += probe(dense->data, list->scalar[k]);
    return count;
}

uint64_t STORM_bitmap_intersect_cardinality(STORM_bitmap_t* STORM_RESTRICT b1,
                                            STORM_bitmap_t* STORM_RESTRICT b2) { /* :572-616 */
    return STORM_bitmap_intersect_cardinality_func(b1, b2, NULL);
}

/* ------------------------------------------------------------------------------------------
 * STORM_bitmap_cont_t — one row (reference storm.c:383-394, :659-824)
 * ---------------------------------------------------------------------------------------- */
void STORM_bitmap_cont_init(STORM_bitmap_cont_t* r) {
    storm_mutated();
    if (r) memset(r, 0, sizeof(*r));
}

STORM_bitmap_cont_t* STORM_bitmap_cont_new() {
    return (STORM_bitmap_cont_t*)calloc(1, sizeof(STORM_bitmap_cont_t));
}

static void cont_release(STORM_bitmap_cont_t* r) {
    for (uint32_t b = 0; b < r->m_bitmaps; ++b) bitmap_release_buffers(&r->bitmaps[b]);
    STORM_aligned_free(r->bitmaps);
    free(r->block_ids);
    memset(r, 0, sizeof(*r));
}

void STORM_bitmap_cont_free(STORM_bitmap_cont_t* r) {
    storm_mutated();
    if (!r) return;
    cont_release(r);
    free(r);
}

static int cont_reserve(STORM_bitmap_cont_t* r) {
    if (r->n_bitmaps < r->m_bitmaps) return 0;
    const uint32_t new_m = r->m_bitmaps == 0 ? 2 : r->m_bitmaps + 8; /* storm.c:698, :730 */
    STORM_bitmap_t* nb = (STORM_bitmap_t*)STORM_aligned_malloc(64, new_m * sizeof(*nb));
    uint32_t* ni = (uint32_t*)realloc(r->block_ids, new_m * sizeof(uint32_t));
    if (ni) r->block_ids = ni;
    if (!nb || !ni) {
        STORM_aligned_free(nb);
        return -1;
    }
    if (r->bitmaps) memcpy(nb, r->bitmaps, r->m_bitmaps * sizeof(*nb));
    for (uint32_t b = r->m_bitmaps; b < new_m; ++b) STORM_bitmap_init(&nb[b]);
    STORM_aligned_free(r->bitmaps);
    r->bitmaps = nb;
    r->m_bitmaps = new_m;
    return 0;
}

/* split the sorted values by value/65536 and give each run to a block: fewer than 4096
 * values -> list kind, else bitmap kind (reference storm.c:692-758) */
int STORM_bitmap_cont_add(STORM_bitmap_cont_t* r, const uint32_t* values,
                          const uint32_t n_values) {
    storm_mutated();
    if (!r) return -1;
    if (!values) return -2;
    if (n_values == 0) return 0;
    uint32_t start = 0;
    while (start < n_values) {
        const uint32_t block = values[start] / BLOCK_BITS;
        uint32_t stop = start + 1;
        while (stop < n_values && values[stop] / BLOCK_BITS == block) ++stop;
        if (cont_reserve(r)) return -3;
        STORM_bitmap_t* b = &r->bitmaps[r->n_bitmaps];
        b->id = block;
        r->block_ids[r->n_bitmaps] = block;
        const int added = stop - start < STORM_DEFAULT_SCALAR_THRESHOLD
                              ? STORM_bitmap_add_scalar_only(b, values + start, stop - start)
                              : STORM_bitmap_add(b, values + start, stop - start);
        if (added < 0) { /* allocation failed: the block is not counted, the row stays consistent */
            STORM_bitmap_clear(b);
            return -3;
        }
        ++r->n_bitmaps;
        r->prev_inserted_value = values[stop - 1];
        start = stop;
    }
    return 1;
}

int STORM_bitmap_cont_clear(STORM_bitmap_cont_t* r) { /* storm.c:816-824 */
    storm_mutated();
    if (!r) return -1;
    for (uint32_t b = 0; b < r->n_bitmaps; ++b) STORM_bitmap_clear(&r->bitmaps[b]);
    r->n_bitmaps = 0;
    r->prev_inserted_value = 0;
    return 1;
}

uint32_t STORM_bitmap_cont_serialized_size(STORM_bitmap_cont_t* r) { /* storm.c:383-394 */
    uint32_t bytes = 0;
    if (r->bitmaps)
        for (uint32_t b = 0; b < r->n_bitmaps; ++b)
            bytes += STORM_bitmap_serialized_size(&r->bitmaps[b]);
    return bytes + (uint32_t)sizeof(uint32_t) * r->n_bitmaps + 3 * (uint32_t)sizeof(uint32_t);
}

/* one row pair on the host (reference storm.c:790-814) */
uint64_t STORM_bitmap_cont_intersect_cardinality_premade(
    const STORM_bitmap_cont_t* STORM_RESTRICT r1, const STORM_bitmap_cont_t* STORM_RESTRICT r2,
    const STORM_compute_func func, uint32_t* out) {
    if (!r1 || !r2 || !out || r1->n_bitmaps == 0 || r2->n_bitmaps == 0) return 0;
    const uint64_t n = STORM_intersect_vector32_unsafe(r1->block_ids, r2->block_ids,
                                                       r1->n_bitmaps, r2->n_bitmaps, out);
    uint64_t count = 0;
    for (uint64_t k = 0; k < n; k += 2)
        count += STORM_bitmap_intersect_cardinality_func(&r1->bitmaps[out[k]],
                                                         &r2->bitmaps[out[k + 1]], func);
    return count;
}

uint64_t STORM_bitmap_cont_intersect_cardinality(
    const STORM_bitmap_cont_t* STORM_RESTRICT r1,
    const STORM_bitmap_cont_t* STORM_RESTRICT r2) { /* storm.c:760-788 */
    if (!r1 || !r2 || r1->n_bitmaps == 0 || r2->n_bitmaps == 0) return 0;
    const uint32_t cap = 2 * (r1->n_bitmaps < r2->n_bitmaps ? r1->n_bitmaps : r2->n_bitmaps);
    uint32_t* out = (uint32_t*)malloc((cap ? cap : 2) * sizeof(uint32_t));
    if (!out) return 0;
    const uint64_t count = STORM_bitmap_cont_intersect_cardinality_premade(r1, r2, NULL, out);
    free(out);
    return count;
}

/* ------------------------------------------------------------------------------------------
 * STORM_t — sparse container (reference storm.c:827-973)
 * ---------------------------------------------------------------------------------------- */
STORM_t* STORM_new() { return (STORM_t*)calloc(1, sizeof(STORM_t)); }

/* device state of a STORM_t handle, per configured GPU: a replica of the block arena (what the all-pairs totals run
 * on) and/or of the rows as a dense bit matrix (what STORM_pairw_matrix runs on); each is built by its first user */
typedef struct {
    storm_hip_sparse_t* a[MAX_DEVICES];
    storm_hip_matrix_t* m[MAX_DEVICES];
    storm_hip_rowlists_t* l[MAX_DEVICES]; /* [r5] a list-only container's rows as window-ordered positions (K5) */
    int have_arena, have_dense;
    int have_lists; /* 0 not tried, 1 built (on every slot of the view), -1 not eligible */
} sparse_state_t;

/* [r6] Streaming for the sparse container. STORM_add hands every bitmap block it finishes to a block stage on the device
 * (storm_hip_stage_*: 8 KiB into a pinned ring, on its way 4 MiB at a time), so that the first all-pairs call — the one
 * call the reference's harness times, benchmark.cpp:605-613 — builds its arena from blocks that are already in HBM (one
 * gather kernel) instead of carrying 8 KiB per block over the bus: 655 MB and 91 ms at BASELINE c4's 50 % load. The list
 * blocks go the same way (storm_hip_stage_add_list: 2 bytes per position; 420 MB and 14 ms of the first call at 20971
 * draws per row). The first add also creates the context (the
 * process's HIP initialisation: 150 - 250 ms that the first call used to pay). Same switch as the dense container's
 * streaming (STORM_HIP_STREAM_ROWS). What was staged is remembered per block as (row, index, id, set-bit count): a block
 * edited behind STORM_add's back (the public per-row / per-block adders) no longer matches at build time and the arena is
 * then built from the host's blocks as before. The stage is given up once the arena exists. */
typedef struct { uint32_t row, b, id, bits; uint64_t token; } stage_blk_t; /* bits: set bits of a bitmap block, length of a list block */
typedef struct {
    storm_hip_stage_t* stage;
    int slot;                 /* the device slot the stage lives on */
    uint32_t generation;      /* the device configuration it was made under */
    stage_blk_t* blk;
    uint64_t n_blk, m_blk;
    uint64_t n_bitmaps_staged; /* (bitmap tokens count up from 0) */
    int off;                  /* 1: staging was given up for this handle (until STORM_clear) */
} storm_stage_t;

static void storm_stage_drop(STORM_t* h, int keep_off) {
    storm_stage_t* sg = (storm_stage_t*)h->hip_stage;
    if (!sg) return;
    if (sg->stage) storm_hip_stage_destroy(g_ctx[sg->slot], sg->stage);
    sg->stage = NULL;
    free(sg->blk);
    sg->blk = NULL;
    sg->n_blk = sg->m_blk = 0;
    sg->n_bitmaps_staged = 0;
    if (keep_off) {
        sg->off = 1;
    } else {
        free(sg);
        h->hip_stage = NULL;
    }
}

static int stream_wanted_locked(void); /* (below, with the dense container's streaming) */

static void storm_stage_row_locked(STORM_t* h, uint32_t row) {
    if (h->hip_private || !stream_wanted_locked()) return;
    storm_stage_t* sg = (storm_stage_t*)h->hip_stage;
    if (sg && sg->off) return;
    const STORM_bitmap_cont_t* r = &h->conts[row];
    g_quiet_ctx = 1;
    storm_hip_ctx_t* ctx = device_ctx(V0); /* (the first add of a process creates the context here) */
    g_quiet_ctx = 0;
    if (!ctx) return;
    if (row == 0) (void)storm_hip_ctx_reserve_staging(ctx); /* the arena builder's pinned ring: now, not inside the first call */
    for (uint32_t b = 0; b < r->n_bitmaps; ++b) {
        const STORM_bitmap_t* blk = &r->bitmaps[b];
        const int is_list = !blk->n_bitmap;
        if (is_list ? (!blk->n_scalar || !blk->scalar) : !blk->data) continue;
        if (!sg) {
            sg = (storm_stage_t*)calloc(1, sizeof(*sg));
            if (!sg) return;
            h->hip_stage = sg;
        }
        if (!sg->stage) {
            sg->slot = V0;
            sg->generation = VIEW_GENERATION;
            if (storm_hip_stage_create(ctx, &sg->stage) != STORM_HIP_OK) goto give_up;
        }
        if (sg->slot != V0 || sg->generation != VIEW_GENERATION) goto give_up; /* another thread's slots, a new configuration */
        if (sg->n_blk == sg->m_blk) {
            const uint64_t m = sg->m_blk ? sg->m_blk * 2 : 1024;
            stage_blk_t* nb = (stage_blk_t*)realloc(sg->blk, m * sizeof(*nb));
            if (!nb) goto give_up;
            sg->blk = nb;
            sg->m_blk = m;
        }
        uint64_t token = 0;
        if (is_list) { /* (lists too since round 6: 420 MB and 14 ms of the first call at c4's 20971 draws per row) */
            if (storm_hip_stage_add_list(ctx, sg->stage, blk->scalar, blk->n_scalar, &token) != STORM_HIP_OK) goto give_up;
            sg->blk[sg->n_blk++] = (stage_blk_t){row, b, blk->id, blk->n_scalar, token};
        } else {
            if (storm_hip_stage_add(ctx, sg->stage, blk->data, &token) != STORM_HIP_OK || token != sg->n_bitmaps_staged) goto give_up;
            ++sg->n_bitmaps_staged;
            sg->blk[sg->n_blk++] = (stage_blk_t){row, b, blk->id, blk->n_bits_set, token};
        }
    }
    return;
give_up:
    storm_stage_drop(h, 1);
}

static void storm_drop_device(STORM_t* h) {
    if (h->hip_arena) {
        sparse_state_t* st = (sparse_state_t*)h->hip_arena;
        for (int d = 0; d < MAX_DEVICES; ++d) {
            if (st->a[d]) storm_hip_sparse_destroy(g_ctx[d], st->a[d]);
            if (st->m[d]) storm_hip_matrix_destroy(g_ctx[d], st->m[d]);
            if (st->l[d]) storm_hip_rowlists_destroy(g_ctx[d], st->l[d]);
        }
        free(st);
        h->hip_arena = NULL;
    }
    h->hip_dirty = 1;
}

void STORM_free(STORM_t* h) {
    if (!h) return;
    storm_drop_device(h);
    storm_stage_drop(h, 0);
    for (uint32_t i = 0; i < h->m_conts; ++i) cont_release(&h->conts[i]);
    free(h->conts);
    free(h);
}

int STORM_add(STORM_t* h, const uint32_t* values, const uint32_t n_values) { /* :844-866 */
    storm_mutated();
    if (!h) return -1;
    if (h->n_conts == h->m_conts) {
        const uint32_t new_m = h->m_conts + 1024;
        STORM_bitmap_cont_t* nc =
            (STORM_bitmap_cont_t*)realloc(h->conts, (size_t)new_m * sizeof(*nc));
        if (!nc) return -3;
        for (uint32_t i = h->m_conts; i < new_m; ++i) STORM_bitmap_cont_init(&nc[i]);
        h->conts = nc;
        h->m_conts = new_m;
    }
    /* NULL or empty `values` still append an (empty) row and return 1, as storm.c:864 does; only an
     * allocation failure inside the row (-3) undoes the add */
    if (STORM_bitmap_cont_add(&h->conts[h->n_conts], values, n_values) == -3) {
        STORM_bitmap_cont_clear(&h->conts[h->n_conts]);
        return -3;
    }
    ++h->n_conts;
    h->hip_dirty = 1;
    if (!h->hip_private && !(h->hip_stage && ((storm_stage_t*)h->hip_stage)->off)) {
        device_lock();
        storm_stage_row_locked(h, h->n_conts - 1);
        device_unlock();
    }
    return 1;
}

int STORM_clear(STORM_t* h) { /* storm.c:868-875 */
    storm_mutated();
    if (!h) return -1;
    for (uint32_t i = 0; i < h->n_conts; ++i) STORM_bitmap_cont_clear(&h->conts[i]);
    h->n_conts = 0;
    storm_drop_device(h);
    storm_stage_drop(h, 0);
    return 1;
}

uint64_t STORM_serialized_size(const STORM_t* h) { /* storm.c:963-973 */
    if (!h) return 0;
    uint64_t bytes = 0;
    for (uint32_t i = 0; i < h->n_conts; ++i)
        bytes += STORM_bitmap_cont_serialized_size(&h->conts[i]);
    return bytes + 2 * sizeof(uint32_t);
}

/* ------------------------------------------------------------------------------------------
 * Serialized form of a STORM_t. The reference defines only its SIZE (storm.c:372-394, :963-973:
 * 2 words per container, 3 + n_bitmaps words per row, 4 words + 8 n_bitmap + 2 n_scalar bytes per
 * block); this is a byte layout with exactly those sizes, little endian, fields at 2-byte
 * alignment (lists of odd length shift what follows), so STORM_serialize writes exactly
 * STORM_serialized_size(h) bytes:
 *   container : u32 n_rows, u32 magic "STM1"
 *   row       : u32 n_blocks, u32 prev_inserted_value, u32 bytes of this row's blocks,
 *               u32 block_ids[n_blocks]
 *   block     : u32 n_bitmap (0 or 1024; bits 30-31 zero), u32 n_bits_set,
 *               u32 n_scalar | n_scalar_set << 31, u32 id,
 *               u64 data[n_bitmap], u16 scalar[n_scalar] (only if n_scalar_set)
 * ---------------------------------------------------------------------------------------- */
#define STORM_SERIAL_MAGIC 0x314d5453u

static uint8_t* put_u32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); return p + 4; }
static uint32_t get_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

uint64_t STORM_serialize(const STORM_t* h, void* buf, uint64_t capacity) {
    if (!h || !buf) return 0;
    const uint64_t need = STORM_serialized_size(h);
    if (capacity < need) return 0;
    uint8_t* p = (uint8_t*)buf;
    p = put_u32(p, h->n_conts);
    p = put_u32(p, STORM_SERIAL_MAGIC);
    for (uint32_t i = 0; i < h->n_conts; ++i) {
        STORM_bitmap_cont_t* r = &h->conts[i];
        uint32_t payload = 0;
        for (uint32_t b = 0; b < r->n_bitmaps; ++b) payload += STORM_bitmap_serialized_size(&r->bitmaps[b]);
        p = put_u32(p, r->n_bitmaps);
        p = put_u32(p, r->prev_inserted_value);
        p = put_u32(p, payload);
        for (uint32_t b = 0; b < r->n_bitmaps; ++b) p = put_u32(p, r->block_ids[b]);
        for (uint32_t b = 0; b < r->n_bitmaps; ++b) {
            const STORM_bitmap_t* blk = &r->bitmaps[b];
            p = put_u32(p, blk->n_bitmap);
            p = put_u32(p, blk->n_bits_set);
            p = put_u32(p, (uint32_t)blk->n_scalar | ((uint32_t)blk->n_scalar_set << 31));
            p = put_u32(p, blk->id);
            if (blk->n_bitmap) {
                memcpy(p, blk->data, (size_t)blk->n_bitmap * 8);
                p += (size_t)blk->n_bitmap * 8;
            }
            if (blk->n_scalar_set && blk->n_scalar) {
                memcpy(p, blk->scalar, (size_t)blk->n_scalar * 2);
                p += (size_t)blk->n_scalar * 2;
            }
        }
    }
    return (uint64_t)(p - (uint8_t*)buf) == need ? need : 0;
}

/* NULL on a truncated or malformed stream (nothing is leaked) */
STORM_t* STORM_deserialize(const void* buf, uint64_t n_bytes) {
    const uint8_t* p = (const uint8_t*)buf;
    if (!p || n_bytes < 8 || get_u32(p + 4) != STORM_SERIAL_MAGIC) return NULL;
    const uint32_t n_rows = get_u32(p);
    /* a row costs at least its 12 header bytes: a header claiming more rows than the stream can hold must
     * not size an allocation (a 16-byte stream once asked calloc for 2^32 row records) */
    if ((uint64_t)n_rows > (n_bytes - 8) / 12) return NULL;
    STORM_t* h = STORM_new();
    if (!h) return NULL;
    uint64_t at = 8;
    int ok = 1;
    h->conts = (STORM_bitmap_cont_t*)calloc(n_rows ? n_rows : 1, sizeof(*h->conts));
    if (!h->conts) ok = 0;
    h->m_conts = ok ? (n_rows ? n_rows : 1) : 0;
    for (uint32_t i = 0; ok && i < n_rows; ++i) {
        STORM_bitmap_cont_t* r = &h->conts[i];
        if (at + 12 > n_bytes) { ok = 0; break; }
        const uint32_t nb = get_u32(p + at);
        r->prev_inserted_value = get_u32(p + at + 4);
        at += 12;
        if (at + 4ull * nb > n_bytes) { ok = 0; break; }
        h->n_conts = i + 1; /* rows up to here are released by STORM_free on failure */
        if (nb) {
            r->bitmaps = (STORM_bitmap_t*)STORM_aligned_malloc(64, (size_t)nb * sizeof(*r->bitmaps));
            r->block_ids = (uint32_t*)malloc((size_t)nb * sizeof(uint32_t));
            if (!r->bitmaps || !r->block_ids) { ok = 0; break; }
            r->m_bitmaps = nb;
            for (uint32_t b = 0; b < nb; ++b) STORM_bitmap_init(&r->bitmaps[b]);
            memcpy(r->block_ids, p + at, 4ull * nb);
        }
        at += 4ull * nb;
        for (uint32_t b = 0; ok && b < nb; ++b) {
            STORM_bitmap_t* blk = &r->bitmaps[b];
            if (at + 16 > n_bytes) { ok = 0; break; }
            const uint32_t n_bitmap = get_u32(p + at), w2 = get_u32(p + at + 8);
            const uint32_t n_scalar = w2 & 0x7fffffffu, has_list = w2 >> 31;
            if (n_scalar > BLOCK_BITS) { ok = 0; break; }
            blk->n_bits_set = get_u32(p + at + 4);
            blk->id = get_u32(p + at + 12);
            at += 16;
            if ((n_bitmap != 0 && n_bitmap != BLOCK_WORDS) || blk->id != r->block_ids[b] ||
                at + 8ull * n_bitmap + (has_list ? 2ull * n_scalar : 0) > n_bytes ||
                (b && blk->id <= r->block_ids[b - 1])) { ok = 0; break; }
            if (n_bitmap) {
                blk->data = (uint64_t*)STORM_aligned_malloc(STORM_get_alignment(), BLOCK_WORDS * 8);
                if (!blk->data) { ok = 0; break; }
                memcpy(blk->data, p + at, BLOCK_WORDS * 8);
                blk->n_bitmap = BLOCK_WORDS;
                at += BLOCK_WORDS * 8;
            }
            if (has_list) {
                blk->n_scalar_set = 1;
                blk->m_scalar = n_scalar ? n_scalar : 1;
                blk->scalar = (uint16_t*)STORM_aligned_malloc(STORM_get_alignment(), (size_t)blk->m_scalar * 2);
                if (!blk->scalar) { ok = 0; break; }
                memcpy(blk->scalar, p + at, (size_t)n_scalar * 2);
                blk->n_scalar = n_scalar;
                at += (uint64_t)n_scalar * 2;
                /* a list is strictly ascending (sorted, duplicate-free: storm.h:227) — the probe kernel counts
                 * every listed element, the dense path ORs bits: a duplicate would make the total depend on
                 * the path — and, for a list-kind block, holds exactly the block's set bits */
                for (uint32_t k = 1; k < n_scalar; ++k)
                    if (blk->scalar[k] <= blk->scalar[k - 1]) { ok = 0; break; }
                if (!n_bitmap && blk->n_bits_set != n_scalar) ok = 0;
                if (!ok) break;
            }
            r->n_bitmaps = b + 1;
        }
    }
    if (ok && at != n_bytes) ok = 0;
    if (!ok) {
        STORM_free(h);
        return NULL;
    }
    h->hip_dirty = 1;
    return h;
}

/* All-pairs total straight from a serialized STORM_t: the bytes are uploaded as they are and the
 * block arena is built on the device (storm_hip_sparse_create_serialized) — no host containers,
 * no per-bit host work. `buf` must be 2-byte aligned. (uint64_t)-1 on a malformed stream or a
 * device failure. */
static uint64_t serialized_pairw_locked(const void* buf, uint64_t n_bytes);
uint64_t STORM_serialized_pairw_intersect_cardinality(const void* buf, uint64_t n_bytes) {
    device_lock();
    const uint64_t total = serialized_pairw_locked(buf, n_bytes);
    device_unlock();
    return total;
}
static uint64_t serialized_pairw_locked(const void* buf, uint64_t n_bytes) {
    if (!buf || n_bytes < 8 || ((uintptr_t)buf & 1)) return ALL_PAIRS_FAILED;
    if (get_u32((const uint8_t*)buf) < 2) return get_u32((const uint8_t*)buf + 4) == STORM_SERIAL_MAGIC ? 0 : ALL_PAIRS_FAILED;
    configure_from_env();
    storm_hip_sparse_t* arena[MAX_DEVICES] = {0};
    uint64_t total = 0;
    int ok = 1, launched = 0;
    const uint32_t world = g_shard_count * (uint32_t)VN;
    for (int d = V0; ok && d < V1; ++d) {
        storm_hip_ctx_t* ctx = device_ctx(d);
        if (!ctx || storm_hip_sparse_create_serialized(ctx, buf, n_bytes, &arena[d]) != STORM_HIP_OK ||
            storm_hip_pairw_sparse_begin(ctx, arena[d], g_shard_rank * (uint32_t)VN + (uint32_t)(d - V0),
                                         world) != STORM_HIP_OK) {
            device_error("serialized all-pairs");
            ok = 0;
        } else {
            launched = d + 1;
        }
    }
    for (int d = V0; d < launched; ++d) {
        uint64_t part = 0;
        if (storm_hip_pairw_sparse_end(g_ctx[d], &part) != STORM_HIP_OK) ok = 0;
        total += part;
    }
    for (int d = 0; d < MAX_DEVICES; ++d)
        if (arena[d]) storm_hip_sparse_destroy(g_ctx[d], arena[d]);
    return across_ranks(ok ? total : ALL_PAIRS_FAILED);
}

/* Fingerprint of what the device arena was built from: rows, blocks per row, and per block its
 * id, kind and set-bit count. STORM_bitmap_cont_add / STORM_bitmap_add are public (storm.h:203-222)
 * and a caller may use them on h->conts[i] directly, behind STORM_add's back; O(blocks) per call
 * makes such an edit rebuild the arena instead of returning the old total. (In-place edits of a
 * block's words that keep its set-bit count are not seen: call STORM_hip_invalidate.) */
static uint64_t storm_fingerprint_rows(const STORM_t* h, uint32_t r0, uint32_t r1) {
    /* four independent accumulators and one multiply per block on the chain: the first version (FNV, three
     * dependent multiplies per block) cost 0.23 ms per all-pairs call at c4's 80000 blocks — more than the
     * kernel below 0.5 % density */
    uint64_t acc[4] = {1469598103934665603ull ^ r0, 0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full,
                       0x165667b19e3779f9ull};
    uint64_t k = 0;
    for (uint32_t i = r0; i < r1; ++i) {
        const STORM_bitmap_cont_t* r = &h->conts[i];
        acc[i & 3u] = (acc[i & 3u] ^ ((uint64_t)r->n_bitmaps + ((uint64_t)i << 32))) * 1099511628211ull;
        for (uint32_t b = 0; b < r->n_bitmaps; ++b, ++k) {
            const STORM_bitmap_t* blk = &r->bitmaps[b];
            const uint64_t v = (((uint64_t)blk->id << 32) | blk->n_bits_set) * 0xff51afd7ed558ccdull ^
                               ((((uint64_t)blk->n_bitmap << 32) | blk->n_scalar) + k) * 0xc4ceb9fe1a85ec53ull;
            acc[k & 3u] = (acc[k & 3u] ^ v) * 1099511628211ull;
        }
    }
    return (acc[0] ^ (acc[1] << 1 | acc[1] >> 63)) + (acc[2] ^ (acc[3] << 3 | acc[3] >> 61));
}

/* The walk is memory-bound on the block records (128 B each, one cache line of them read): four quarters of the
 * rows, three of them on helper threads (created on first use, parked on a condition variable) once a container has
 * enough rows to pay for waking them — 0.13 -> 0.05 ms at c4's 80000 blocks. The value is the same either way. */
#define FP_PARTS 4
static struct {
    pthread_t th[FP_PARTS];
    int started;
    pthread_mutex_t job; /* the helpers serve one caller at a time (callers on different device slots run side by side) */
    pthread_mutex_t mu;
    pthread_cond_t go, done;
    uint64_t generation;
    int pending;
    const STORM_t* h;
    uint64_t part[FP_PARTS];
} g_fp = {.job = PTHREAD_MUTEX_INITIALIZER,
          .mu = PTHREAD_MUTEX_INITIALIZER,
          .go = PTHREAD_COND_INITIALIZER,
          .done = PTHREAD_COND_INITIALIZER};

static void fp_bounds(const STORM_t* h, int q, uint32_t* r0, uint32_t* r1) {
    *r0 = (uint32_t)((uint64_t)h->n_conts * (uint64_t)q / FP_PARTS);
    *r1 = (uint32_t)((uint64_t)h->n_conts * (uint64_t)(q + 1) / FP_PARTS);
}
static void* fp_main(void* p) {
    const int q = (int)(intptr_t)p;
    uint64_t seen = 0;
    for (;;) {
        pthread_mutex_lock(&g_fp.mu);
        while (g_fp.generation == seen) pthread_cond_wait(&g_fp.go, &g_fp.mu);
        seen = g_fp.generation;
        const STORM_t* h = g_fp.h;
        pthread_mutex_unlock(&g_fp.mu);
        uint32_t r0, r1;
        fp_bounds(h, q, &r0, &r1);
        const uint64_t v = storm_fingerprint_rows(h, r0, r1);
        pthread_mutex_lock(&g_fp.mu);
        g_fp.part[q] = v;
        if (--g_fp.pending == 0) pthread_cond_signal(&g_fp.done);
        pthread_mutex_unlock(&g_fp.mu);
    }
    return NULL;
}
static uint64_t storm_fingerprint(const STORM_t* h) {
    uint64_t part[FP_PARTS];
    /* a second caller that finds the helpers busy walks its rows itself rather than wait for them */
    int threaded = h->n_conts >= 4096 && pthread_mutex_trylock(&g_fp.job) == 0;
    if (threaded && !g_fp.started) {
        int ok = 1;
        for (int q = 1; q < FP_PARTS && ok; ++q) {
            if (pthread_create(&g_fp.th[q], NULL, fp_main, (void*)(intptr_t)q) != 0) ok = 0;
            else pthread_detach(g_fp.th[q]);
        }
        g_fp.started = ok ? 1 : -1; /* (a partial set of helpers is never used: -1 = serial for good) */
    }
    if (threaded && g_fp.started == 1) {
        pthread_mutex_lock(&g_fp.mu);
        g_fp.h = h;
        g_fp.pending = FP_PARTS - 1;
        ++g_fp.generation;
        pthread_cond_broadcast(&g_fp.go);
        pthread_mutex_unlock(&g_fp.mu);
        uint32_t r0, r1;
        fp_bounds(h, 0, &r0, &r1);
        part[0] = storm_fingerprint_rows(h, r0, r1);
        pthread_mutex_lock(&g_fp.mu);
        while (g_fp.pending > 0) pthread_cond_wait(&g_fp.done, &g_fp.mu);
        for (int q = 1; q < FP_PARTS; ++q) part[q] = g_fp.part[q];
        pthread_mutex_unlock(&g_fp.mu);
    } else {
        for (int q = 0; q < FP_PARTS; ++q) {
            uint32_t r0, r1;
            fp_bounds(h, q, &r0, &r1);
            part[q] = storm_fingerprint_rows(h, r0, r1);
        }
    }
    if (threaded) pthread_mutex_unlock(&g_fp.job);
    uint64_t f = 0x9e3779b97f4a7c15ull ^ h->n_conts;
    for (int q = 0; q < FP_PARTS; ++q) f = (f ^ part[q]) * 1099511628211ull;
    return f;
}

/* The device state of a handle: thrown away and started afresh (empty, fingerprinted) when the handle is marked
 * dirty or was built for another device configuration / another thread's slots. *fresh = 1 then: what is built into
 * it next comes from the container as it is now. NULL: out of memory. */
static sparse_state_t* storm_state(STORM_t* h, int* fresh) {
    *fresh = 0;
    if (!h->hip_arena || h->hip_dirty || h->hip_generation != VIEW_GENERATION) {
        storm_drop_device(h);
        sparse_state_t* st = (sparse_state_t*)calloc(1, sizeof(*st));
        if (!st) {
            host_error("STORM_t device state: out of host memory");
            return NULL;
        }
        h->hip_arena = st;
        h->hip_dirty = 0;
        h->hip_generation = VIEW_GENERATION;
        h->hip_fingerprint = storm_fingerprint(h);
        *fresh = 1;
    }
    return (sparse_state_t*)h->hip_arena;
}

/* One replica per device slot of the caller of (dense = 0) the block arena or (dense = 1) the dense row matrix.
 * Nothing is flattened here: the device library gets the block headers and a POINTER to every block's list or
 * bitmap where it lies in the containers, ships the raw data through its pinned ring and lays it out on the
 * device (storm_hip_sparse_create_blocks / storm_hip_matrix_create_from_blocks). A first call at c4's 20971 draws
 * per row cost 0.6 - 0.9 s with the host-side flattening and element layout of rounds 2 - 3. */
static int storm_build_device(STORM_t* h, sparse_state_t* st, int dense) {
    uint64_t n_blocks = 0;
    for (uint32_t i = 0; i < h->n_conts; ++i) n_blocks += h->conts[i].n_bitmaps;
    uint64_t* row_off = (uint64_t*)malloc((h->n_conts + 1ull) * sizeof(uint64_t));
    uint32_t* ids = (uint32_t*)malloc((n_blocks + 1) * sizeof(uint32_t));
    uint8_t* kinds = (uint8_t*)malloc(n_blocks + 1);
    uint32_t* lens = (uint32_t*)malloc((n_blocks + 1) * sizeof(uint32_t));
    const void** ptrs = (const void**)malloc((n_blocks + 1) * sizeof(void*));
    int rc = -1, not_eligible = 0;
    if (row_off && ids && kinds && lens && ptrs) {
        uint64_t nb = 0;
        for (uint32_t i = 0; i < h->n_conts; ++i) {
            row_off[i] = nb;
            for (uint32_t b = 0; b < h->conts[i].n_bitmaps; ++b, ++nb) {
                const STORM_bitmap_t* blk = &h->conts[i].bitmaps[b];
                ids[nb] = blk->id;
                kinds[nb] = blk->n_bitmap ? 1 : 0;
                lens[nb] = blk->n_bitmap ? 0 : blk->n_scalar;
                ptrs[nb] = blk->n_bitmap ? (const void*)blk->data : (const void*)blk->scalar;
            }
        }
        row_off[h->n_conts] = nb;
        rc = 0;
        /* [r6] the bitmap blocks STORM_add has already sent to this device: a token per block while every bitmap block of
         * the container still is the block that was staged, in the order it was staged (else: from the host, as before) */
        storm_stage_t* sg = dense != 1 ? (storm_stage_t*)h->hip_stage : NULL; /* (the arena and, for its lists, K5) */
        uint64_t* tokens = NULL;
        if (sg && sg->stage && !sg->off && sg->generation == VIEW_GENERATION && sg->slot >= V0 && sg->slot < V1 &&
            (tokens = (uint64_t*)malloc((n_blocks + 1) * sizeof(uint64_t))) != NULL) {
            uint64_t k = 0, at = 0;
            int ok = 1;
            for (uint32_t i = 0; i < h->n_conts && ok; ++i)
                for (uint32_t b = 0; b < h->conts[i].n_bitmaps; ++b, ++at) {
                    const STORM_bitmap_t* blk = &h->conts[i].bitmaps[b];
                    tokens[at] = ~0ull;
                    if (blk->n_bitmap ? !blk->data : (!blk->n_scalar || !blk->scalar)) continue; /* (never staged) */
                    if (k < sg->n_blk && sg->blk[k].row == i && sg->blk[k].b == b && sg->blk[k].id == blk->id &&
                        sg->blk[k].bits == (blk->n_bitmap ? blk->n_bits_set : blk->n_scalar)) {
                        tokens[at] = sg->blk[k++].token;
                    } else {
                        ok = 0;
                        break;
                    }
                }
            if (!ok || k != sg->n_blk) {
                free(tokens);
                tokens = NULL;
            }
        }
        for (int d = V0; d < V1 && rc == 0; ++d) {
            storm_hip_ctx_t* ctx = device_ctx(d);
            const int r = !ctx ? -1
                          : dense == 2 ? ((tokens && d == sg->slot)
                                              ? storm_hip_rowlists_create_blocks_staged(ctx, h->n_conts, n_blocks, row_off, ids,
                                                                                        kinds, lens, ptrs, sg->stage, tokens, &st->l[d])
                                              : storm_hip_rowlists_create_blocks(ctx, h->n_conts, n_blocks, row_off, ids, kinds,
                                                                                 lens, ptrs, &st->l[d]))
                          : dense ? storm_hip_matrix_create_from_blocks(ctx, h->n_conts, n_blocks, row_off, ids, kinds,
                                                                        lens, ptrs, &st->m[d])
                          : (tokens && d == sg->slot)
                              ? storm_hip_sparse_create_blocks_staged(ctx, h->n_conts, n_blocks, row_off, ids, kinds, lens,
                                                                      ptrs, sg->stage, tokens, &st->a[d])
                              : storm_hip_sparse_create_blocks(ctx, h->n_conts, n_blocks, row_off, ids, kinds, lens,
                                                               ptrs, &st->a[d]);
            if (r != STORM_HIP_OK) {
                device_error(dense == 2 ? "storm_hip_rowlists_create_blocks"
                             : dense    ? "storm_hip_matrix_create_from_blocks"
                                        : "storm_hip_sparse_create_blocks");
                rc = -1;
            }
            if (dense == 2 && rc == 0 && !st->l[d]) not_eligible = 1; /* (the same answer on every slot) */
        }
        free(tokens);
        if (dense == 0 && h->hip_stage) storm_stage_drop(h, 1); /* the arena holds the blocks now (or the build failed) */
    } else {
        host_error("STORM_t device state: out of host memory");
    }
    free(row_off); free(ids); free(kinds); free(lens); free((void*)ptrs);
    if (rc == 0 && !(dense == 2 && not_eligible)) {
        if (dense == 2) st->have_lists = 1;
        else if (dense) st->have_dense = 1;
        else st->have_arena = 1;
    } else {
        for (int d = 0; d < MAX_DEVICES; ++d) {
            if (!dense && st->a[d]) { storm_hip_sparse_destroy(g_ctx[d], st->a[d]); st->a[d] = NULL; }
            if (dense == 1 && st->m[d]) { storm_hip_matrix_destroy(g_ctx[d], st->m[d]); st->m[d] = NULL; }
            if (dense == 2 && st->l[d]) { storm_hip_rowlists_destroy(g_ctx[d], st->l[d]); st->l[d] = NULL; }
        }
        if (dense == 2 && rc == 0) st->have_lists = -1;
    }
    return rc;
}

/* all configured devices work concurrently on disjoint shards (one host thread each); the host adds the partials */
typedef struct {
    sparse_state_t* st;
    uint64_t part[MAX_DEVICES];
    const STORM_t* check; /* fingerprint this container while the devices work (the first slot's thread) */
    uint64_t fingerprint;
    int first, count; /* the caller's device slots */
} sparse_job_t;

static int sparse_job(int d, int phase, void* arg) {
    sparse_job_t* j = (sparse_job_t*)arg;
    const uint32_t world = g_shard_count * (uint32_t)j->count;
    const uint32_t rank = g_shard_rank * (uint32_t)j->count + (uint32_t)(d - j->first);
    if (phase == 0) return storm_hip_pairw_sparse_begin(g_ctx[d], j->st->a[d], rank, world);
    if (d == j->first && j->check) j->fingerprint = storm_fingerprint(j->check); /* every device has been launched */
    return storm_hip_pairw_sparse_end(g_ctx[d], &j->part[d]);
}

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
#define HOST_LAP(what)                                                                        \
    do {                                                                                      \
        if (lap_on) {                                                                         \
            const double t_ = now_ms();                                                       \
            fprintf(stderr, "[storm_host] %-34s %8.2f ms\n", what, t_ - lap_t0);              \
            lap_t0 = t_;                                                                      \
        }                                                                                     \
    } while (0)
static uint64_t storm_pairw_device_locked(STORM_t* h);
static uint64_t storm_pairw_device(STORM_t* h) {
    device_lock();
    const uint64_t total = storm_pairw_device_locked(h);
    device_unlock();
    return total;
}
static int always_fingerprint(void) {
    static int v = -1;
    int x = __atomic_load_n(&v, __ATOMIC_RELAXED);
    if (x < 0) {
        const char* e = getenv("STORM_HIP_ALWAYS_FINGERPRINT");
        x = e && e[0] == '1';
        __atomic_store_n(&v, x, __ATOMIC_RELAXED);
    }
    return x;
}
static uint64_t storm_pairw_device_locked(STORM_t* h) {
    if (h->n_conts < 2) return 0;
    configure_from_env();
    const int lap_on = getenv("STORM_HIP_TIMING") != NULL;
    double lap_t0 = lap_on ? now_ms() : 0;
    /* A cached arena is checked against the container (storm_fingerprint, O(blocks): 0.1 - 0.3 ms at c4) WHILE the
     * pass runs on it: the pass is launched first, the fingerprint is computed on the caller's thread behind the
     * launches, and only a mismatch — a caller edited rows through the public adders — throws the total away,
     * rebuilds the arena and runs again. */
    int verified = 0; /* 1: the state was started from the container as it is now */
    sparse_state_t* st = storm_state(h, &verified);
    if (!st) return across_ranks(ALL_PAIRS_FAILED);
    HOST_LAP("state, fingerprint");
    for (;;) {
        if (!st->have_arena && storm_build_device(h, st, 0)) {
            storm_drop_device(h);
            return across_ranks(ALL_PAIRS_FAILED);
        }
        HOST_LAP("arena (built or kept)");
        sparse_job_t j;
        j.st = st;
        const uint64_t epoch = storm_epoch(); /* read before the pass: a mutator running meanwhile makes the next call check again */
        j.check = (!verified && !h->hip_private && (h->hip_epoch != epoch || always_fingerprint())) ? h : NULL;
        j.fingerprint = 0;
        j.first = V0;
        j.count = VN;
        memset(j.part, 0, sizeof(j.part));
        if (run_on_devices(sparse_job, &j, "all-pairs pass (STORM_t)")) return across_ranks(ALL_PAIRS_FAILED);
        HOST_LAP("pass");
        if (j.check && j.fingerprint != h->hip_fingerprint) {
            storm_drop_device(h);
            if (!(st = storm_state(h, &verified))) return across_ranks(ALL_PAIRS_FAILED);
            continue;
        }
        h->hip_epoch = epoch; /* verified (or just built) at this epoch */
        uint64_t total = 0;
        for (int d = V0; d < V1; ++d) total += j.part[d];
        return across_ranks(total);
    }
}

/* Extension (storm.h): the per-pair matrix of a STORM_t — what STORM_bitmap_cont_intersect_cardinality (storm.c:790-814)
 * returns for rows i < j, for every pair at once. The rows are laid out as a dense bit matrix on the device (built
 * on the first call, kept with the handle like the arena, checked against the container's fingerprint) and the tile
 * kernels of STORM_contig_pairw_matrix write the triangle. */
static int storm_lists_worthwhile(const STORM_t* h);
static int storm_pairw_matrix_locked(STORM_t* h, int op, uint32_t* out, uint64_t out_rows, uint64_t out_ld) {
    if (!h) return -1;
    if (!out) return -2;
    const uint64_t n = h->n_conts;
    if (out_rows < n || out_ld < n) return -4;
    if (n == 0) return 0;
    configure_from_env();
    int fresh = 0;
    sparse_state_t* st = storm_state(h, &fresh);
    if (!st) return -3;
    const uint64_t epoch = storm_epoch();
    if (!fresh && !h->hip_private && (h->hip_epoch != epoch || always_fingerprint()) &&
        storm_fingerprint(h) != h->hip_fingerprint) { /* rows edited through the public adders behind STORM_add */
        storm_drop_device(h);
        if (!(st = storm_state(h, &fresh))) return -3;
    }
    h->hip_epoch = epoch;
    /* [r5] one device, a list-only container that is sparse enough: straight from the lists (K5), no dense replica */
    if (VN == 1 && device_ctx(V0) && storm_hip_rowlists_worthwhile(g_ctx[V0], NULL) &&
        (st->have_lists != 0 || storm_lists_worthwhile(h))) {
        if (st->have_lists == 0 && storm_build_device(h, st, 2)) return -3;
        if (st->have_lists == 1 && storm_hip_rowlists_worthwhile(g_ctx[V0], st->l[V0])) {
            if (storm_hip_rowlists_pairw_matrix(g_ctx[V0], st->l[V0], op, out, out_ld) != STORM_HIP_OK) {
                device_error("storm_hip_rowlists_pairw_matrix");
                return -3;
            }
            return 0;
        }
    }
    if (!st->have_dense && storm_build_device(h, st, 1)) return -3;
    return pairw_matrix_bands(st->m, n, op, out, out_ld);
}

int STORM_pairw_matrix(STORM_t* h, int op, uint32_t* out, uint64_t out_rows, uint64_t out_ld) {
    device_lock();
    const int rc = storm_pairw_matrix_locked(h, op, out, out_rows, out_ld);
    device_unlock();
    return rc;
}

/* The same into DEVICE memory (storm.h: STORM_pairw_matrix_device): the 4 N^2 bytes of output never cross the bus — at the
 * README's STORM_t shape (N = 10000: 400 MB) the copy to pageable host memory was half of a call. One device slot only:
 * `d_out` lives on one GPU. */
static int one_slot_or_refuse(const char* who) {
    configure_from_env();
    if (VN == 1) return 0;
    char msg[160];
    snprintf(msg, sizeof(msg), "%s: the output lives on ONE device: narrow the thread's view to one slot (STORM_hip_set_thread_devices)", who);
    host_error(msg);
    return -5;
}
/* whether the per-pair matrix of `h` should come from its row lists (K5): the rule of storm_hip_rowlists_worthwhile on the
 * counts the block headers give — asked BEFORE the lists are built, so that a container the rule sends to the dense replica
 * does not pay for both */
static int storm_lists_worthwhile(const STORM_t* h) {
    uint64_t n_elems = 0, max_id = 0;
    for (uint32_t i = 0; i < h->n_conts; ++i)
        for (uint32_t b = 0; b < h->conts[i].n_bitmaps; ++b) {
            const STORM_bitmap_t* blk = &h->conts[i].bitmaps[b];
            if (blk->n_bitmap) return 0; /* a bitmap block: the dense replica's case */
            n_elems += blk->n_scalar;
            if (blk->n_scalar && blk->id > max_id) max_id = blk->id;
        }
    return storm_hip_rowlists_worthwhile_counts(g_ctx[V0], h->n_conts, n_elems, (max_id + 1) * 65536ull);
}

int STORM_pairw_matrix_device(STORM_t* h, int op, uint32_t* d_out, uint64_t out_rows, uint64_t out_ld) {
    if (!h) return -1;
    if (!d_out) return -2;
    const int lap_on = getenv("STORM_HIP_TIMING") != NULL;
    double lap_t0 = lap_on ? now_ms() : 0;
    device_lock();
    int rc = one_slot_or_refuse("STORM_pairw_matrix_device");
    const uint64_t n = h->n_conts;
    if (!rc && (out_rows < n || out_ld < n)) rc = -4;
    if (!rc && n != 0) {
        int fresh = 0;
        sparse_state_t* st = storm_state(h, &fresh);
        if (!st) rc = -3;
        if (!rc) {
            const uint64_t epoch = storm_epoch();
            if (!fresh && !h->hip_private && (h->hip_epoch != epoch || always_fingerprint()) &&
                storm_fingerprint(h) != h->hip_fingerprint) {
                storm_drop_device(h);
                if (!(st = storm_state(h, &fresh))) rc = -3;
            }
            if (!rc) h->hip_epoch = epoch;
        }
        /* [r5] a list-only container that is sparse enough: straight from the lists (K5, storm_hip_lists.hip) — no dense
         * replica is built at all then */
        int from_lists = 0;
        HOST_LAP("state, fingerprint");
        if (!rc && !device_ctx(V0)) rc = -3; /* (contexts are opened on first use) */
        if (!rc && (op == 0 || op == 1 || op == 2) && storm_hip_rowlists_worthwhile(g_ctx[V0], NULL) && /* (NULL: are the lists switched on at all) */
            (st->have_lists != 0 || storm_lists_worthwhile(h))) {
            if (st->have_lists == 0 && storm_build_device(h, st, 2)) rc = -3;
            from_lists = !rc && st->have_lists == 1 && storm_hip_rowlists_worthwhile(g_ctx[V0], st->l[V0]);
        }
        HOST_LAP("row lists built");
        if (!rc && from_lists) {
            if (storm_hip_rowlists_pairw_matrix_device(g_ctx[V0], st->l[V0], op, d_out, out_ld) != STORM_HIP_OK) {
                device_error("storm_hip_rowlists_pairw_matrix_device");
                rc = -3;
            }
            HOST_LAP("matrix from the lists");
        } else {
            if (!rc && !st->have_dense && storm_build_device(h, st, 1)) rc = -3;
            if (!rc && storm_hip_pairw_matrix_device(g_ctx[V0], st->m[V0], op, d_out, out_ld) != STORM_HIP_OK) {
                device_error("storm_hip_pairw_matrix_device");
                rc = -3;
            }
        }
    }
    device_unlock();
    return rc;
}

int STORM_contig_pairw_matrix_device(STORM_contiguous_t* h, int op, uint32_t* d_out, uint64_t out_rows, uint64_t out_ld) {
    if (!h) return -1;
    if (!d_out) return -2;
    device_lock();
    int rc = one_slot_or_refuse("STORM_contig_pairw_matrix_device");
    const uint64_t n = h->n_data;
    if (!rc && (out_rows < n || out_ld < n)) rc = -4;
    if (!rc && n != 0) {
        dense_state_t* st = contig_mirror(h);
        if (!st) rc = -3;
        else if (storm_hip_pairw_matrix_device(g_ctx[V0], st->m[V0], op, d_out, out_ld) != STORM_HIP_OK) {
            device_error("storm_hip_pairw_matrix_device");
            rc = -3;
        }
    }
    device_unlock();
    return rc;
}

uint64_t STORM_n_rows(const STORM_t* h) { return h ? h->n_conts : 0; }

/* Extensions (storm.h): forget the device copy of a handle whose public members were edited
 * in place (the reference structs are not opaque, storm.h:157-200); the next all-pairs call
 * uploads again. */
int STORM_hip_invalidate(STORM_t* h) {
    if (!h) return -1;
    storm_drop_device(h);
    storm_stage_drop(h, 1); /* (blocks edited in place: what was staged is not what they hold) */
    return 0;
}

int STORM_contig_hip_invalidate(STORM_contiguous_t* h) {
    if (!h) return -1;
    contig_drop_device(h);
    contig_lists_end(h); /* rows edited in place: the list mirror cannot follow them ... */
    /* ... and the rows so far travel as the words they now are, not as the positions they were added with (the
     * watermark lives in the handle: nothing here depends on an allocation) */
    h->hip_words_below = h->n_data;
    contig_pending_t* p = (contig_pending_t*)h->hip_pending;
    if (p) {
        p->row0 = h->n_data;
        p->n_rows = 0;
        p->n_pos = 0;
    }
    return 0;
}

uint64_t STORM_pairw_intersect_cardinality(STORM_t* h) { /* storm.c:877-895 */
    if (!h) return (uint64_t)-1;
    return storm_pairw_device(h);
}

uint64_t STORM_pairw_intersect_cardinality_blocked(STORM_t* h, uint32_t bsize) { /* :897-961 */
    (void)bsize;
    if (!h) return (uint64_t)-1;
    return storm_pairw_device(h);
}
