/*
 * storm.h — the StormBitmaps container API, served by the MI355X-native library
 * libstorm_hip.so (stormbitmaps_amd/). Source-compatible with the reference header
 * (mklarqvist/StormBitmaps storm.h): same type names, same public struct members in the same
 * order, same function signatures and return conventions, so a C/C++ caller of the reference
 * recompiles against this header and links -lstorm_hip instead of storm.c.
 *
 * What differs behind the API
 *   - The all-pairs entry points (STORM_contig_pairw_*, STORM_pairw_*, STORM_wrapper_*) run on
 *     the GPU: hand-written gfx950 kernels reached through the C-ABI shim in storm_hip.h.
 *     There is no CPU fallback; without a usable device they return (uint64_t)-1 and leave
 *     a message in storm_hip_last_error().
 *   - `bsize` / `block_size` arguments are accepted and ignored as tuning hints: the device
 *     tiling is chosen internally and the integer result does not depend on it.
 *   - Results follow the intended semantics (the mathematically exact count). The reference
 *     deviates from it in three sparse regimes (SURVEY.md §8 a-note, defects D1-D3).
 *   - *_free() also releases the handle itself and every device buffer.
 *   - Both container structs carry private members after the reference's public ones.
 *
 * Every declaration cites the reference lines it stands in for (storm.h / storm.c).
 */
#ifndef STORM_H_MI355X_DROPIN
#define STORM_H_MI355X_DROPIN

#include <assert.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "libalgebra/libalgebra.h" /* reference storm.h:33 */

/* reference storm.h:35-47 — kept for callers that size blocks with them */
#ifndef STORM_CACHE_BLOCK_SIZE
#define STORM_CACHE_BLOCK_SIZE 256e3
#endif
#ifndef STORM_DEFAULT_BLOCK_SIZE
#define STORM_DEFAULT_BLOCK_SIZE 65536
#endif
#ifndef STORM_DEFAULT_SCALAR_THRESHOLD
#define STORM_DEFAULT_SCALAR_THRESHOLD 4096
#endif

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what these headers declare is its whole export list */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ------------------------------------------------------------------ list kernels (host) --
 * Single-pair helpers on host memory (reference storm.h:56-61, storm.c:4-129). They are not
 * on the all-pairs path; they exist so that one-pair callers keep working. */
uint64_t STORM_intersect_vector16_cardinality(const uint16_t* STORM_RESTRICT v1,
                                              const uint16_t* STORM_RESTRICT v2,
                                              const uint32_t len1, const uint32_t len2);
uint64_t STORM_intersect_vector32_unsafe(const uint32_t* STORM_RESTRICT v1,
                                         const uint32_t* STORM_RESTRICT v2, const uint32_t len1,
                                         const uint32_t len2, uint32_t* STORM_RESTRICT out);
uint64_t STORM_intersect_bitmaps_scalar_list(const uint64_t* STORM_RESTRICT b1,
                                             const uint64_t* STORM_RESTRICT b2,
                                             const uint32_t* l1, const uint32_t* l2,
                                             const uint32_t n1, const uint32_t n2);

/* list-aware leaf signature (reference storm.h:66-67) */
typedef uint64_t (*STORM_compute_lfunc)(const uint64_t*, const uint64_t*, const uint32_t*,
                                        const uint32_t*, const size_t, const size_t);

/* ------------------------------------------------------ raw-buffer all-pairs (device) ----
 * sum over row pairs of popcount(row_i & row_j) for `n_vectors` rows of `n_ints` 64-bit words
 * in a caller-owned HOST buffer (reference storm.h:95-148, storm.c:132-369). The buffer is
 * copied to the device per call. `f` / `fl` must be NULL or leaves exported by this library.
 * The *_list variants return the same exact count as the plain ones (the position lists are
 * a CPU-side shortcut; the device kernel is density independent). */
uint64_t STORM_wrapper_diag(const uint32_t n_vectors, const uint64_t* vals,
                            const uint32_t n_ints, const STORM_compute_func f);
uint64_t STORM_wrapper_diag_blocked(const uint32_t n_vectors, const uint64_t* vals,
                                    const uint32_t n_ints, const STORM_compute_func f,
                                    uint32_t block_size);
/* every row of vals1 against every row of vals2 (storm.c:153-171, intent of storm.h:72-77) */
uint64_t STORM_wrapper_square(const uint32_t n_vectors1, const uint64_t* STORM_RESTRICT vals1,
                              const uint32_t n_vectors2, const uint64_t* STORM_RESTRICT vals2,
                              const uint32_t n_ints, const STORM_compute_func f);
uint64_t STORM_wrapper_diag_list(const uint32_t n_vectors, const uint64_t* STORM_RESTRICT vals,
                                 const uint32_t n_ints, const uint32_t* STORM_RESTRICT n_alts,
                                 const uint32_t* STORM_RESTRICT alt_positions,
                                 const uint32_t* STORM_RESTRICT alt_offsets,
                                 const STORM_compute_func f, const STORM_compute_lfunc fl,
                                 const uint32_t cutoff);
uint64_t STORM_wrapper_diag_list_blocked(const uint32_t n_vectors,
                                         const uint64_t* STORM_RESTRICT vals,
                                         const uint32_t n_ints,
                                         const uint32_t* STORM_RESTRICT n_alts,
                                         const uint32_t* STORM_RESTRICT alt_positions,
                                         const uint32_t* STORM_RESTRICT alt_offsets,
                                         const STORM_compute_func f,
                                         const STORM_compute_lfunc fl, const uint32_t cutoff,
                                         uint32_t block_size);

/* ------------------------------------------------------------------------ containers ---- */
typedef struct STORM_bitmap_s STORM_bitmap_t;
typedef struct STORM_bitmap_cont_s STORM_bitmap_cont_t;
typedef struct STORM_s STORM_t;
typedef struct STORM_contiguous_bitmap_s STORM_contiguous_bitmap_t;
typedef struct STORM_contiguous_s STORM_contiguous_t;

/* one 65536-bit block of one row: a sorted uint16 list OR a 1024-word bitmap
 * (reference storm.h:158-166) */
struct STORM_bitmap_s {
    STORM_ALIGN(64) uint64_t* data;
    STORM_ALIGN(64) uint16_t* scalar;
    uint32_t n_bitmap : 30, own_data : 1, own_scalar : 1;
    uint32_t n_bits_set;
    uint32_t n_scalar : 31, n_scalar_set : 1, n_missing;
    uint32_t m_scalar;
    uint32_t id;
};

/* one row: its blocks in ascending id order (reference storm.h:168-173) */
struct STORM_bitmap_cont_s {
    STORM_bitmap_t* bitmaps;
    uint32_t* block_ids;
    uint32_t n_bitmaps, m_bitmaps;
    uint32_t prev_inserted_value;
};

/* sparse container (reference storm.h:175-178) + private device state */
struct STORM_s {
    STORM_bitmap_cont_t* conts;
    uint32_t n_conts, m_conts;
    /* private */
    void* hip_arena;        /* device replicas of the flattened arena, rebuilt when dirty */
    uint32_t hip_dirty;
    uint32_t hip_generation; /* device configuration the arena was built for */
    uint64_t hip_fingerprint; /* rows / blocks / set-bit counts the arena was built from */
    uint32_t hip_private;     /* a container the library keeps for itself (the list mirror of a
                                 STORM_contiguous_t): nobody edits its members, no fingerprint per call */
    uint64_t hip_epoch;       /* the mutation epoch (storm_host.c) the device arena was last verified at */
    void* hip_stage;          /* [r6] the bitmap blocks STORM_add has already sent to the device (storm_host.c) */
};

/* one row of the dense container (reference storm.h:181-186) */
struct STORM_contiguous_bitmap_s {
    uint64_t* data;
    uint32_t* scalar;
    uint32_t n_scalar;
};

/* dense container (reference storm.h:188-200) + private device state */
struct STORM_contiguous_s {
    uint64_t* data;
    uint32_t* scalar;
    uint32_t* n_scalar;
    STORM_contiguous_bitmap_t* bitmaps;
    uint64_t n_data, m_data;
    uint64_t tot_scalar, m_scalar;
    uint64_t vector_length;
    uint32_t n_bitmaps_vector;
    STORM_compute_func intsec_func;
    uint32_t alignment;
    uint32_t scalar_cutoff;
    /* private */
    uint64_t* scalar_offset; /* start of each row's list in `scalar` (per row)     */
    void* hip_matrix;        /* storm_hip_matrix_t*: device mirror of `data`        */
    uint64_t hip_rows_synced;
    uint64_t hip_rows_capacity;
    STORM_t* hip_lists;      /* the same rows as a STORM_t while EVERY row is below scalar_cutoff:   */
    uint32_t hip_lists_off;  /* such a container goes through the list-probe kernel (see storm_host.c) */
    void* hip_pending;       /* positions of rows not yet on the device (STORM_contig_add; storm_host.c)    */
    uint64_t hip_words_below; /* rows below this one were edited in place (STORM_contig_hip_invalidate): sent as words */
};

/* per-block API (reference storm.h:203-212, storm.c:372-380, :398-656) */
STORM_bitmap_t* STORM_bitmap_new();
void STORM_bitmap_init(STORM_bitmap_t* all);
void STORM_bitmap_free(STORM_bitmap_t* bitmap);
int STORM_bitmap_add(STORM_bitmap_t* bitmap, const uint32_t* values, const uint32_t n_values);
int STORM_bitmap_add_with_scalar(STORM_bitmap_t* bitmap, const uint32_t* values,
                                 const uint32_t n_values);
int STORM_bitmap_add_scalar_only(STORM_bitmap_t* bitmap, const uint32_t* values,
                                 const uint32_t n_values);
uint64_t STORM_bitmap_intersect_cardinality(STORM_bitmap_t* STORM_RESTRICT bitmap1,
                                            STORM_bitmap_t* STORM_RESTRICT bitmap2);
uint64_t STORM_bitmap_intersect_cardinality_func(STORM_bitmap_t* STORM_RESTRICT bitmap1,
                                                 STORM_bitmap_t* STORM_RESTRICT bitmap2,
                                                 const STORM_compute_func func);
int STORM_bitmap_clear(STORM_bitmap_t* bitmap);
uint32_t STORM_bitmap_serialized_size(STORM_bitmap_t* bitmap);

/* per-row API (reference storm.h:215-222, storm.c:383-394, :659-824) */
STORM_bitmap_cont_t* STORM_bitmap_cont_new();
void STORM_bitmap_cont_init(STORM_bitmap_cont_t* bitmap);
void STORM_bitmap_cont_free(STORM_bitmap_cont_t* bitmap);
int STORM_bitmap_cont_add(STORM_bitmap_cont_t* bitmap, const uint32_t* values,
                          const uint32_t n_values);
int STORM_bitmap_cont_clear(STORM_bitmap_cont_t* bitmap);
uint64_t STORM_bitmap_cont_intersect_cardinality(
    const STORM_bitmap_cont_t* STORM_RESTRICT bitmap1,
    const STORM_bitmap_cont_t* STORM_RESTRICT bitmap2);
uint64_t STORM_bitmap_cont_intersect_cardinality_premade(
    const STORM_bitmap_cont_t* STORM_RESTRICT bitmap1,
    const STORM_bitmap_cont_t* STORM_RESTRICT bitmap2, const STORM_compute_func func,
    uint32_t* out);
uint32_t STORM_bitmap_cont_serialized_size(STORM_bitmap_cont_t* bitmap);

/* sparse container (reference storm.h:225-232, storm.c:827-973).
 * STORM_add: values sorted ascending; returns 1 (an empty input still appends an empty row).
 * STORM_pairw_*: sum over row pairs i<j of |row_i ∩ row_j|, computed on the GPU;
 * NULL handle or device failure -> (uint64_t)-1. */
STORM_t* STORM_new();
void STORM_free(STORM_t* bitmap);
int STORM_add(STORM_t* bitmap, const uint32_t* values, const uint32_t n_values);
int STORM_clear(STORM_t* bitmap);
uint64_t STORM_pairw_intersect_cardinality(STORM_t* bitmap);
uint64_t STORM_pairw_intersect_cardinality_blocked(STORM_t* bitmap, uint32_t bsize);
uint64_t STORM_serialized_size(const STORM_t* bitmap);
/* Extension: the serialized form whose SIZE the reference defines (STORM_serialized_size,
 * storm.c:372-394, :963-973) but never writes. STORM_serialize writes exactly
 * STORM_serialized_size(h) bytes (layout: storm_host.c) and returns that count, 0 if `capacity`
 * is too small. STORM_deserialize returns NULL on a malformed stream.
 * STORM_serialized_pairw_intersect_cardinality computes the all-pairs total of a serialized
 * container without building it on the host: the bytes are uploaded as they are and unpacked into
 * the block arena by the device (`buf` 2-byte aligned; (uint64_t)-1 on failure). */
uint64_t STORM_serialize(const STORM_t* bitmap, void* buf, uint64_t capacity);
STORM_t* STORM_deserialize(const void* buf, uint64_t n_bytes);
uint64_t STORM_serialized_pairw_intersect_cardinality(const void* buf, uint64_t n_bytes);
/* (reference storm.h:231 declares STORM_intersect_cardinality_square but never defines it,
 *  storm.c:975; nothing to stand in for.) */

/* dense container (reference storm.h:235-242, storm.c:1001-1346).
 * STORM_contig_add: returns n_values; 0 for an empty input (no row appended); -1 / -2 for a
 * NULL handle / NULL values.
 * STORM_contig_pairw_*: GPU; NULL handle or device failure -> (uint64_t)-1; the *_list
 * variants return (uint64_t)-2 / -3 before the first add, like the reference. */
STORM_contiguous_t* STORM_contig_new(size_t vector_length);
void STORM_contig_free(STORM_contiguous_t* bitmap);
int STORM_contig_add(STORM_contiguous_t* bitmap, const uint32_t* values,
                     const uint32_t n_values);
int STORM_contig_clear(STORM_contiguous_t* bitmap);
uint64_t STORM_contig_pairw_intersect_cardinality(STORM_contiguous_t* bitmap);
uint64_t STORM_contig_pairw_intersect_cardinality_blocked(STORM_contiguous_t* bitmap,
                                                          uint32_t bsize);
uint64_t STORM_contig_pairw_intersect_cardinality_list(STORM_contiguous_t* bitmap);
uint64_t STORM_contig_pairw_intersect_cardinality_blocked_list(STORM_contiguous_t* bitmap,
                                                               uint32_t bsize);

/* Extension: the per-pair matrix the reference only sums (README.md:41). op: 0 = intersect,
 * 1 = union, 2 = symmetric difference. `out` holds out_rows x out_ld uint32, row-major; entry
 * (i, j) = popcount(row_i OP row_j) for i < j < n_data at out[i * out_ld + j], 0 for i >= j.
 * Returns 0; -1 NULL handle, -2 NULL out, -3 device failure (see STORM_hip_error), -4 when
 * out_rows or out_ld is smaller than the number of rows the handle holds (nothing is written).
 * STORM_contig_n_rows: rows appended so far (empty inputs append none, storm.c:1034). */
uint64_t STORM_contig_n_rows(const STORM_contiguous_t* bitmap);
int STORM_contig_pairw_matrix(STORM_contiguous_t* bitmap, int op, uint32_t* out, uint64_t out_rows,
                              uint64_t out_ld);
/* The same for a STORM_t: entry (i, j), i < j, is what STORM_bitmap_cont_intersect_cardinality(&conts[i], &conts[j])
 * returns (storm.c:790-814; the LD use case of README.md:165-167 on the sparse container), or the union / symmetric
 * difference count for op 1 / 2. The device keeps the rows as a dense bit matrix for this (65536 x (largest block
 * id + 1) bits per row; rows beyond 2^25 bits are refused with -3). Same return codes; STORM_n_rows: rows added. */
uint64_t STORM_n_rows(const STORM_t* bitmap);
int STORM_pairw_matrix(STORM_t* bitmap, int op, uint32_t* out, uint64_t out_rows, uint64_t out_ld);
/* Both with the output left in DEVICE memory: `d_out` is a device pointer (hipMalloc) to out_rows x out_ld uint32 on the one
 * device slot the calling thread drives (-5 when its view spans several: STORM_hip_set_thread_devices). Entries i >= j of
 * the n x n window are written as 0 only inside the tiles the kernel touches: clear the buffer once if they matter. The
 * 4 n^2 bytes then never cross the bus (half of a STORM_pairw_matrix call at n = 10000). Same return codes otherwise. */
int STORM_pairw_matrix_device(STORM_t* bitmap, int op, uint32_t* d_out, uint64_t out_rows, uint64_t out_ld);
int STORM_contig_pairw_matrix_device(STORM_contiguous_t* bitmap, int op, uint32_t* d_out, uint64_t out_rows,
                                     uint64_t out_ld);

/* ------------------------------------------------------------- extensions (not in ref) ---
 * Device selection for the entry points above. By default device 0 computes everything.
 * STORM_hip_set_devices(n, ids): the pair space is sharded over the listed GPUs of this node
 * (one replica of the data per GPU, disjoint shards of the tile list) and the per-GPU partial
 * sums are added on the host. Multi-PROCESS runs (one rank per GPU, RCCL all-reduce) use
 * STORM_hip_set_shard(rank, world): each process then returns only its shard's partial. */
int STORM_hip_set_devices(int n_devices, const int* device_ids);
/* The containers keep a device copy of their rows between all-pairs calls. It follows every
 * change made through STORM_add / STORM_clear / STORM_contig_add / STORM_contig_clear, and for
 * STORM_t also edits made with the public per-row / per-block adders directly on h->conts[i]
 * (a fingerprint of rows, blocks and set-bit counts is compared on every call). What it cannot
 * see is a caller writing into the public buffers in place (h->data, bitmaps[i].data: the
 * reference structs are not opaque): after such an edit call the matching function below, or the
 * next all-pairs call answers for the rows as they were. Returns 0, -1 for a NULL handle. */
int STORM_hip_invalidate(STORM_t* bitmap);
int STORM_contig_hip_invalidate(STORM_contiguous_t* bitmap);
/* Caller threads on distinct GPUs: after STORM_hip_set_devices(n, ids), a thread that calls
 * STORM_hip_set_thread_devices(first_slot, n_slots) drives only the device slots [first_slot, first_slot + n_slots)
 * with its all-pairs calls (n_slots = 0: all slots again) — the handles it uses keep their device mirrors there — and
 * only those slots are locked, so threads on distinct slots run side by side (one lock per device slot; the raw-buffer
 * STORM_wrapper_* calls keep one set of device matrices per process and lock all slots). Returns 0, -1 if the run of
 * slots is not inside the configuration. A handle is still for one thread at a time, as in the reference. */
int STORM_hip_set_thread_devices(int first_slot, int n_slots);
/* A context option (include/storm_hip.h: storm_hip_ctx_set_option — kernel forms, work-list shaping) for every device
 * context behind the handles, existing and future; the environment variable STORM_HIP_OPTIONS="key=value,key=value" does the
 * same. Tuning and A/B measurements: no option changes a result. 0, or -1 (unknown key / value out of range). */
int STORM_hip_set_option(const char* key, int64_t value);
int STORM_hip_set_shard(uint32_t shard_rank, uint32_t shard_count);
/* What the last all-pairs call ran, over this process's devices (storm_hip.h: storm_hip_last_pass_report):
 * out[0] mask of STORM_HIP_RAN_*, out[1] dense 64-bit word pairs, out[2] list-probe lookups, out[3] rows a
 * lookup stands for. For harnesses that price a row against the roof of the kernel that ran. */
int STORM_hip_last_pass(uint64_t out[4]);
const char* STORM_hip_error(void);
/* Multi-PROCESS runs, one process per GPU: every process sets its shard (STORM_hip_set_shard) and joins one
 * RCCL communicator; from then on every all-pairs entry point above returns the SUM over all processes (the
 * shard partials all-reduced over xGMI: ncclAllReduce of one uint64), so host code written for the reference
 * needs nothing else. Rank 0 obtains the 128-byte id and hands it to the other processes (pipe, file, MPI);
 * STORM_hip_comm_init is collective — every process calls it, after it was forked / started and before its
 * first all-pairs call. Return 0, or -1 with the reason in STORM_hip_error().
 * (tools/storm_benchmark.cpp --ranks N: fork before any HIP call, id through a pipe.) */
int STORM_hip_comm_unique_id(uint8_t id[128]);
int STORM_hip_comm_init(const uint8_t id[128]);
int STORM_hip_comm_finalize(void);
/* Threading. Like the reference (no locks anywhere in storm.c), a HANDLE is not thread-safe: one thread at a
 * time per STORM_t / STORM_contiguous_t. Different handles may be used from different threads: every entry point
 * that touches a device (the all-pairs calls, STORM_contig_pairw_matrix, the raw-buffer wrappers, the streaming of
 * STORM_contig_add) locks the device slots it drives — one lock per configured device, the contexts behind the
 * handles are shared — so concurrent passes are safe; threads whose views (STORM_hip_set_thread_devices) are
 * disjoint slots run side by side, threads on the same slots one after the other. The device selection
 * (STORM_hip_set_devices / _set_shard, the environment) is read on first use: change it only while no other
 * thread is inside the library.
 * STORM_hip_shutdown(): releases the wrappers' cached device matrices and every device context (handles that
 * still hold device copies re-create them on their next all-pairs call). Returns 0. */
int STORM_hip_shutdown(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* STORM_H_MI355X_DROPIN */
