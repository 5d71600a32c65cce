/*
 * storm_hip.h — C-ABI of the MI355X (gfx950) device shim behind the storm.h API.
 *
 * This is the drop-in boundary for the pairwise AND+popcount hot path: plain pointers and
 * sizes, no C++ or torch types. Host code (C) calls these; they launch hand-written HIP
 * kernels (stormbitmaps_amd/csrc/storm_hip.hip). Each entry point names the reference
 * interface it replaces (file:line in mklarqvist/StormBitmaps).
 *
 * Conventions
 *   - every function returns 0 on success or a negative STORM_HIP_E* code; the message of the
 *     last failure on the calling thread is available from storm_hip_last_error().
 *   - there is NO CPU fallback: without a usable gfx950 device every compute entry point
 *     fails with STORM_HIP_ENODEV / STORM_HIP_EHIP.
 *   - `stream` arguments are hipStream_t passed as void* (NULL = the default stream).
 *   - a "dense matrix" is the device mirror of STORM_contiguous_t::data (storm.h:188-200):
 *     row-major uint64 rows, bit v of a row in word v/64 at bit v%64 (storm.c:1114). On the
 *     device the rows are padded with zeros to a multiple of 64 words and the row count to a
 *     multiple of 128 so that no kernel needs a ragged-edge path.
 */
#ifndef STORM_HIP_H_
#define STORM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what these headers declare is its whole export list */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define STORM_HIP_OK 0
#define STORM_HIP_EINVAL (-1) /* bad argument                                  */
#define STORM_HIP_ENODEV (-2) /* no HIP device / not gfx950                    */
#define STORM_HIP_EHIP (-3)   /* a HIP runtime call failed                     */
#define STORM_HIP_ENOMEM (-4) /* host or device allocation failed              */

typedef struct storm_hip_ctx_s storm_hip_ctx_t;       /* one device + stream + workspace */
typedef struct storm_hip_matrix_s storm_hip_matrix_t; /* dense bitmap matrix in HBM      */
typedef struct storm_hip_sparse_s storm_hip_sparse_t; /* flattened STORM_t arena in HBM  */
typedef struct storm_hip_rowlists_s storm_hip_rowlists_t; /* rows of a list-only STORM_t as window-ordered positions (K5) */

/* ---- library / device ---- */
const char* storm_hip_last_error(void);
int storm_hip_device_count(void);
/* arch string of `device` ("gfx950:sramecc+:xnack-") into buf */
int storm_hip_device_arch(int device, char* buf, size_t buflen);
/* PCI address of `device` ("0000:75:00.0") into buf: which physical GPU a rank drives */
int storm_hip_device_pci_bus_id(int device, char* buf, size_t buflen);

/* ---- context: device, stream, reusable workspace ---- */
int storm_hip_ctx_create(int device, void* stream, storm_hip_ctx_t** out);
int storm_hip_ctx_set_stream(storm_hip_ctx_t* ctx, void* stream);
int storm_hip_ctx_synchronize(storm_hip_ctx_t* ctx);
void storm_hip_ctx_destroy(storm_hip_ctx_t* ctx);

/* ---- dense matrix lifecycle (device mirror of STORM_contiguous_t, storm.c:1001-1147) ---- */
int storm_hip_matrix_create(storm_hip_ctx_t* ctx, uint64_t n_rows, uint32_t n_words,
                            storm_hip_matrix_t** out); /* zero-filled */
/* change the logical row count (device mirror of a container that grows row by row, storm.c:1078):
 * growing reallocates with amortised doubling and keeps the rows; new rows are zero until uploaded */
int storm_hip_matrix_resize(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m, uint64_t n_rows);
/* copy n_rows host rows (row stride = src_stride_words) into rows [row0, row0+n_rows) */
int storm_hip_matrix_upload(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m, uint64_t row0,
                            uint64_t n_rows, const uint64_t* host_rows,
                            uint64_t src_stride_words);
/* same from a device buffer (e.g. a torch tensor's data_ptr), device-to-device */
int storm_hip_matrix_import(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m, uint64_t row0,
                            uint64_t n_rows, const void* device_rows,
                            uint64_t src_stride_words);
/* copy rows back to the host (tests) */
int storm_hip_matrix_download(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, uint64_t row0,
                              uint64_t n_rows, uint64_t* host_rows, uint64_t dst_stride_words);
/* device-side construction from sorted position lists in CSR form (host pointers):
 * replaces the bit-setting loop of STORM_contig_add, storm.c:1103-1115 */
int storm_hip_matrix_set_rows_from_positions(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m,
                                             uint64_t row0, uint64_t n_rows,
                                             const uint64_t* offsets, const uint32_t* positions);
/* synthetic fill on the device: identical bits to storm_synth_positions() (storm_synth.h) */
int storm_hip_matrix_fill_synthetic(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m,
                                    uint64_t n_bits, uint32_t draws, uint64_t seed);
int storm_hip_matrix_clear(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m);
void storm_hip_matrix_destroy(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m);
uint64_t storm_hip_matrix_rows(const storm_hip_matrix_t* m);
uint32_t storm_hip_matrix_words(const storm_hip_matrix_t* m);
uint64_t storm_hip_matrix_stride_words(const storm_hip_matrix_t* m);
void* storm_hip_matrix_device_ptr(const storm_hip_matrix_t* m);

/* ---- the hot path --------------------------------------------------------------------
 * total = sum over row pairs i<j of popcount(row_i & row_j), restricted to this shard's share
 * of the pair space (shard_rank of shard_count; 0 of 1 = everything). The shards partition
 * the pairs, so the sum of the per-shard totals over all ranks is the full total.
 * Replaces: STORM_contig_pairw_intersect_cardinality[_blocked] (storm.c:1149-1241),
 *           STORM_wrapper_diag[_blocked] (storm.c:132-150, :222-279) and, through them, the
 *           libalgebra leaf STORM_compute_func (call sites storm.c:1167,1205,1217,1227,1236).
 * _launch: asynchronous on the ctx stream; d_total is a DEVICE pointer to one uint64.
 * plain   : synchronous; *h_total on the host. */
int storm_hip_pairw_dense_launch(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m,
                                 uint32_t shard_rank, uint32_t shard_count, uint64_t* d_total);
/* The same total for rows that are still in the CALLER's memory: `host_rows` (src_stride_words words per row, >= the
 * matrix's) replace all rows of `m`, travelling in panels of whole 256-row tiles on a second stream while the panel before is
 * being multiplied (the pairs whose later row lies in a panel are multiplied as soon as it has landed); *h_total on the
 * host, synchronous. What STORM_wrapper_diag[_blocked] (storm.c:132-150, :222-279) run on one device: those entry points
 * get the caller's buffer anew on every call. Matrices below 2048 rows: the copy, then the pass. */
int storm_hip_pairw_dense_upload(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m, const uint64_t* host_rows,
                                 uint64_t src_stride_words, uint64_t* h_total);
int storm_hip_pairw_dense(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m,
                          uint32_t shard_rank, uint32_t shard_count, uint64_t* h_total);
/* split form, so that one host thread can keep several GPUs busy: _begin launches into the
 * ctx's own result word, _end waits for it and copies it to the host */
int storm_hip_pairw_dense_begin(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m,
                                uint32_t shard_rank, uint32_t shard_count);
int storm_hip_pairw_dense_end(storm_hip_ctx_t* ctx, uint64_t* h_total);
/* rectangle: sum over i in A, j in B of popcount(A_i & B_j) — STORM_wrapper_square,
 * storm.c:153-171 (intended semantics, storm.h:72-77) */
int storm_hip_square_dense(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* a,
                           const storm_hip_matrix_t* b, uint64_t* h_total);
/* per-pair counts of one tile (tests / materialised output):
 * out[(i-i0)*(j1-j0)+(j-j0)] = popcount(row_i & row_j); out is a HOST buffer */
int storm_hip_tile_counts(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, uint64_t i0,
                          uint64_t i1, uint64_t j0, uint64_t j1, uint32_t* h_out);
/* Sibling pair counts of the upstream leaf library (union / symmetric-difference cardinality,
 * reference README.md:24-26). They are not separate kernels: with the per-row
 * set-bit counts n_i,  |a|b| = n_a + n_b - |a&b|  and  |a^b| = n_a + n_b - 2|a&b|, so both ride
 * on the intersect path. */
#define STORM_HIP_OP_AND 0
#define STORM_HIP_OP_OR 1
#define STORM_HIP_OP_XOR 2
/* h_counts[i] = popcount(row_i), n_rows entries (host pointer) */
int storm_hip_row_counts(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, uint32_t* h_counts);
/* *h_total = sum_{i<j} popcount(row_i OP row_j) */
int storm_hip_pairw_dense_op(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op,
                             uint64_t* h_total);

/* Materialised strict upper triangle of XX^T on the matrix cores (the product the reference
 * deliberately does not write out, README.md:41):
 *   out[i * ld + j] = popcount(row_i OP row_j) for i < j; other entries are left untouched.
 * _device: `d_out` is a DEVICE pointer to n_rows x ld uint32 (ld >= n_rows); synchronous.
 * plain  : `h_out` is a HOST n_rows x n_rows uint32 buffer; entries i >= j come back as 0.
 * Rows of 2^24 bits and more (beyond exact f32 accumulation in one go) are cut along k and the
 * parts added; the limit is 2^25 bits per row (32-bit DMA offsets of the tile kernel). */
int storm_hip_pairw_matrix_device(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op,
                                  uint32_t* d_out, uint64_t ld);
int storm_hip_pairw_matrix(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op,
                           uint32_t* h_out);
/* A band of that triangle, for matrices whose n_rows^2 output does not fit at once: rows
 * [row0, row0 + n_band_rows) only, written from output row 0:
 *   d_out[(i - row0) * ld + j] = popcount(row_i OP row_j), row0 <= i < row0 + n_band_rows, i < j.
 * `d_out` is a DEVICE pointer to n_band_rows x ld uint32, ld >= n_rows. Bands of a few thousand
 * rows keep the matrix cores as busy as the whole triangle does. */
int storm_hip_pairw_matrix_band_device(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op,
                                       uint64_t row0, uint64_t n_band_rows, uint32_t* d_out,
                                       uint64_t ld);
/* The same band into HOST memory: h_out[(i - row0) * ld + j], n_band_rows x ld uint32 with
 * ld >= n_rows; entries i >= j come back as 0. _begin only enqueues (one band per GPU can be in
 * flight from one host thread), _end waits for the context's stream. */
int storm_hip_pairw_matrix_band_begin(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op,
                                      uint64_t row0, uint64_t n_band_rows, uint32_t* h_out,
                                      uint64_t ld);
int storm_hip_pairw_matrix_band_end(storm_hip_ctx_t* ctx);
int storm_hip_pairw_matrix_band(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op,
                                uint64_t row0, uint64_t n_band_rows, uint32_t* h_out, uint64_t ld);
/* Materialised rectangle A x B (the two-matrix product XY^T):
 *   out[i * ld + j] = popcount(a_i OP b_j) for every row i of `a` and j of `b` (same row width).
 * _device: `d_out` is a DEVICE pointer, a->n_rows x ld uint32 with ld >= b->n_rows; synchronous.
 * plain  : `h_out` is a HOST a->n_rows x b->n_rows uint32 buffer. */
int storm_hip_square_matrix_device(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* a,
                                   const storm_hip_matrix_t* b, int op, uint32_t* d_out, uint64_t ld);
int storm_hip_square_matrix(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* a,
                            const storm_hip_matrix_t* b, int op, uint32_t* h_out);

/* sum_c C(n_c,2) on the device — verification identity only, never the product
 * path: used by tests at sizes where a CPU pairwise oracle is infeasible */
int storm_hip_column_identity(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m,
                              uint64_t* h_total);

/* kernel selection / tuning knobs (benchmarks; defaults are what ships; none changes a result)
 *   key "variant": -1 = auto (default): matrix-core strips (4) whenever 64 shadow rows fit 32-bit
 *                  DMA offsets, else the popcount kernel (2)
 *                  0/1/2 = K1 popcount kernel, B operand direct / via VGPR->LDS / via LDS-DMA
 *                  3 = K2 FP4 matrix-core tiles, 4 = K2s FP4 matrix-core strips (256-row A tiles),
 *                  5 = wide strips (512-row A tiles)
 *   key "keep_shadow": 1 = keep the FP4 shadow of a dense matrix between all-pairs calls while the
 *                  matrix is unchanged (default 0). Only valid when the matrix is modified through
 *                  this library alone (upload / import / set_rows / fill / clear), never through
 *                  storm_hip_matrix_device_ptr. storm.h handles use it.
 *   key "time_kernels": see storm_hip_kernel_time
 *   key "sparse_probe": sparse container, block columns whose blocks are all lists: -1 = auto (the
 *                  list-probe kernel K4 when the mean list has <= 1000 positions, else the dense path),
 *                  0 = never, 1 = every eligible column
 *   key "seg_rows": K1 B rows per work item (default 256)
 *   key "chunks_per_item": K1 k-chunks (64 words each) per work item, 0 = auto
 *   key "k2_stages_per_item": K2 tile kernel k-slice length in 128-bit stages (default 32)
 *   keys "k2_max_run" (128), "k2_tail_slices" (3), "k2_tail_run" (32): strip work-list shaping;
 *        "k2_persistent" (0): per-XCD work queues; "k2_ring" (4): LDS ring depth 3..5;
 *        "k2_pitch_pad" (-1 = auto): extra bytes per shadow row; "k2_lds_pad": cap workgroups per CU;
 *        "k2_matrix_split" (1): cut the last round of matrix-output tiles along k;
 *        "k2_shape" (16): MFMA form of the default strips, 16 = 16x16x128, 32 = 32x32x64;
 *        "k2_strip_operands" (0): operands of the strips: 0 / 5 = the bit matrix itself, the FP4 image of every B
 *        stage built in the LDS by the workgroup (K2b, the default); 4 = FP4 shadow (expansion pass + strips); 2 = one
 *        stage stream per workgroup on bit operands (K2q); 1 / 3: tools build only;
 *        "k2_tile_shape" (0): materialised-output kernel: 0 = by the matrix (5 for a dense matrix, 2 for the dense
 *        replica of a sparse container); 5 = tilering_kernel: both operands as FP4 images built once per workgroup in the
 *        LDS, 16x16x128 MFMAs, SIMD partners half a stage apart ("k2_ring_sync" (0): 0 = one barrier per stage, 1 = arrival
 *        counters in the LDS); 2 / 1 = bit operands inflated to FP4 in registers, 32x32x64 MFMAs (two / one wave per SIMD);
 *        3 / 4 = B as LDS images, A in registers; 16 / 32 = the FP4-shadow kernels (1, 16: tools build only);
 *        "k2_fold_inline" (-1): the all-pairs total is folded inside the last kernel of the pass by the workgroup dispatched
 *        last (-1: for short launches, where the fold launch and its gaps are a fifth of a pass; 1: always; 0: never);
 *        "k2_tile_cost_diag" (63), "k2_tile_cost_ragged" (30): percent of a full tile's time the
 *        planner assumes for diagonal / ragged-column tiles when it cuts the last round into k-parts;
 *        "k2_shadow_budget_mb" (98304): when the FP4 shadow (4 x the bits) of a matrix would exceed
 *        this many MiB the pass runs k-chunk by k-chunk over a compact shadow of one chunk
 *        (HBM-tiled: bounded footprint for any M x N; 0 = never chunk)
 *   key "k2_debug", "k2_ring" >= 11: timing probes only (results may then be wrong), see
 *        storm_hip_mfma.hip
 *   read-only "variant_used": what the last dense launch ran; "n_cus" */
int storm_hip_ctx_set_option(storm_hip_ctx_t* ctx, const char* key, int64_t value);
/* Whether storm_hip_ctx_set_option would accept (key, value), without a context (STORM_HIP_OK, or STORM_HIP_EINVAL with
 * storm_hip_last_error() saying why): callers that remember options for contexts still to be made validate with it. */
int storm_hip_option_check(const char* key, int64_t value);
/* Allocates the context's pinned staging ring (24 MiB, what the sparse arena builder ships lists and blocks through) now
 * instead of inside the first arena build: callers that know an arena is coming (storm.h's STORM_add) call it while nobody
 * is waiting for a result (5 - 6 ms of a first all-pairs call otherwise). */
int storm_hip_ctx_reserve_staging(storm_hip_ctx_t* ctx);
int64_t storm_hip_ctx_get_option(storm_hip_ctx_t* ctx, const char* key);
/* Tuning aid: with option "k2_ring" = 18 the strip kernel records, per work item, its start and
 * end on the 100 MHz device counter, three phase marks (operands in / diagonal phase done / main
 * loop done, 16 bits each, relative to the start) and the XCC_ID register of the workgroup's first
 * wave. out[i*8 ..] = {start, end, phases, xcc_id, a_row0, diag, stages, k-slice}; results of the
 * launch are unaffected. `out` may be NULL to query the item count. */
int storm_hip_debug_strip_trace(storm_hip_ctx_t* ctx, uint64_t* out, uint64_t capacity_items,
                                uint64_t* n_items);

/* Duration of the dominant kernel alone (the popcount, tile or strip kernel — not the FP4
 * expansion or the final fold), for roofline accounting: with option "time_kernels" = 1 every
 * pairwise launch brackets that kernel with HIP events on the launch stream. This call waits for
 * the stream, returns the summed duration and the number of launches since the last call, and
 * starts a new series. "time_kernels" = 0 pauses the bracketing and keeps the series, 2 resumes it
 * (1 starts a new one): bench.py brackets every 4th step at N > 1. */
int storm_hip_kernel_time(storm_hip_ctx_t* ctx, double* sum_ms, uint64_t* launches);
/* work decomposition of the last dense launch: out[0]=work items, [1]=k-chunks per item,
 * [2]=word-pairs executed incl. zero padding (popcount kernel) / k-chunks of the pass (matrix-core
 * strips: 1 unless the shadow budget forced HBM tiling), [3]=segments (popcount kernel) / block columns
 * counted by the list-probe kernel (sparse container) */
int storm_hip_last_launch_info(storm_hip_ctx_t* ctx, uint64_t out[4]);

/* ---- multi-GPU work split of the default (matrix-core strip) path, host-only ---------------
 * The work list shard `shard_rank` of `shard_count` would multiply for an n_rows x n_words
 * matrix, as 5 uint32 per item: {a_row0, diag, j0, j1, ks} =
 *   A tile rows [a_row0, a_row0 + 256) x k-slice ks (bits [256 ks, 256 ks + 256) of every row) x
 *   (diag ? the strict upper triangle inside the A tile : nothing) + the 64-row B blocks
 *   [j0, j1) (rows [64 j0, 64 j1)), i.e. the pairs (i, j), i in the A tile, j in those blocks.
 * Ownership: whole k-slices, in units of 4 (one 128-byte line of the bit matrix), go
 * to shard (ks / 4) % shard_count; the leftover slices (fewer than 4 x shard_count) are cut along
 * the pair space, longest item first onto the least loaded shard. The lists of all shards tile (pair, k-slice) space exactly once. Touches no
 * device; `out` may be NULL to query the item count. Shards the reference loop storm.c:1199-1238. */
int storm_hip_strip_plan(uint64_t n_rows, uint32_t n_words, uint32_t shard_rank,
                         uint32_t shard_count, uint32_t* out, uint64_t capacity_items,
                         uint64_t* n_items);
/* The same with the operand form and the ownership mode spelled out (storm_hip_strip_plan = form 0, mode 0):
 *   form 0: the FP4-shadow strips: slice ks = bits [256 ks, 256 ks + 256) of every row;
 *   form 1: the strips on bit operands (K2b, the DEFAULT path): slice ks = class pair ks & 1 — the bits b of
 *           the 512-bit chunk ks / 2 (bits [512 (ks / 2), +512)) with (b % 4) / 2 == ks & 1 — 256 bit positions
 *           as well; 2 x ceil(n_words / 8) slices;
 *   pair_space 0: whole k-slices first, leftover slices along the pair space (above);
 *   pair_space 1: EVERY slice is cut along the pair space (context option k2_shard_pairs): a shard multiplies
 *           its share of every slice's items (longest first onto the least loaded shard, the load carried
 *           from slice to slice). */
int storm_hip_strip_plan2(uint64_t n_rows, uint32_t n_words, uint32_t shard_rank, uint32_t shard_count,
                          int form, int pair_space, uint32_t* out, uint64_t capacity_items, uint64_t* n_items);
/* The same with the work-list options spelled out — the context options of the same names, and the device's
 * compute-unit count (context option "n_cus", read-only) — i.e. EXACTLY the list a context with these options
 * launches for this shard: one function derives the shaping for the device path and for this planner. max_run = 0
 * selects the run length automatically (64 / 96 / 128, whichever schedules the SLOWEST of the shard_count ranks
 * shortest: the choice never depends on shard_rank, so that all ranks cut the slices they deal among themselves at
 * the same length); *run_chosen (may be NULL) receives the run length used. storm_hip_strip_plan2 = the defaults
 * (max_run 0, tail_run 32, tail_slices 3, lpt_rounds 6, 256 compute units). */
int storm_hip_strip_plan3(uint64_t n_rows, uint32_t n_words, uint32_t shard_rank, uint32_t shard_count,
                          int form, int pair_space, int max_run, int tail_run, int tail_slices, int lpt_rounds,
                          uint32_t n_cus, uint32_t* out, uint64_t capacity_items, uint64_t* n_items,
                          int* run_chosen);

/* The item list of the materialised-output kernel for matrices of few 256 x 256 tiles (K2h, tile128_kernel; DESIGN.md
 * §4) on a device of `n_cus` compute units, as 8 uint32 per item, in launch order:
 *   {I, J, first chunk, chunks, tile, part, n_parts, narrow}
 *   = the pairs (row of the 128-row tile I, row of the 128-row tile J) over the 512-bit chunks [first, first + chunks) of
 *     every row; `tile` numbers the tiles, an item is part `part` of the `n_parts` its tile is cut into along k (their
 *     sums meet inside the launch; `narrow`: through windows of 16-bit counts).
 * n_rows_b == 0: the triangle of one matrix of n_rows_a rows (tiles I <= J), band_rows != 0: only the output rows
 * [band_row0, band_row0 + band_rows); n_rows_b != 0: the rectangle A x B (B's tiles count on behind A's rows padded to a
 * multiple of 256). slots_per_cu / min_chunks / diag_cost_pct: the context options k2_part_slots / k2_part_min_chunks /
 * k2_part_cost_diag. The items of a tile cover its chunks exactly once, the tiles every pair of the output exactly once.
 * Cuts the reference loop storm.c:1199-1238 (its per-pair results kept). Host only; `out` may be NULL to query the count. */
int storm_hip_matrix_plan(uint64_t n_rows_a, uint64_t n_rows_b, uint32_t n_words, uint64_t band_row0, uint64_t band_rows,
                          uint32_t n_cus, int slots_per_cu, int min_chunks, int diag_cost_pct, uint32_t* out,
                          uint64_t capacity_items, uint64_t* n_items);

/* The same for the one-launch stage stream on bit operands (K2q, the default for matrices of up to 8192
 * rows on one device): the segments shard `shard_rank` of `shard_count` walks on a device of
 * `n_cus` compute units, workgroup by workgroup, as 8 uint32 per segment:
 *   {workgroup, a_blk, ks, b_first, n_b, range_nb, diag, stages}
 *   = A tile = the 64-row blocks [a_blk, a_blk + 4) x k-slice ks (bits [512 ks, 512 ks + 512) of every row);
 *     diag ? the pairs inside the tile (strictly above the diagonal) : nothing; then the n_b later blocks
 *     (b_first + i) % range_nb, i < n_b, whole: the pairs (row of the tile, row of the block).
 * Tile pairs are dealt cyclically (tile I takes the (T - 1) / 2 tiles behind it, wrapping around; with an
 * even number of tiles the opposite one goes to the lower tile on even k-slices and to the upper on odd ones),
 * so the segments of all shards cover every unordered row pair x k-slice exactly once. Shards the reference
 * loop storm.c:1199-1238 into contiguous parts of the k-slice-major stage stream. Host only. */
int storm_hip_stream_plan(uint64_t n_rows, uint32_t n_words, uint32_t shard_rank, uint32_t shard_count,
                          uint32_t n_cus, uint32_t* out, uint64_t capacity_segments, uint64_t* n_segments,
                          uint32_t* n_workgroups);

/* ---- the 8-byte exchange of a multi-process run (one process per GPU), over RCCL / xGMI ----------------
 * Every rank computes its shard's partial (storm_hip_pairw_dense(..., shard_rank, shard_count), or a storm.h
 * handle after STORM_hip_set_shard) and the partials are summed: ncclAllReduce(count 1, ncclUint64, ncclSum)
 * — SURVEY §5; the reference is single-process and has no counterpart. librccl.so is loaded on first use
 * (dlopen; STORM_HIP_RCCL names another build), single-GPU users need none.
 *   rank 0: storm_hip_comm_unique_id(id); hand the 128 bytes to the other ranks (pipe, file, MPI, ...);
 *   every rank, AFTER forking / starting its own process and BEFORE the first collective:
 *     storm_hip_comm_init_rank(ctx, id, rank, world, &comm)      (collective: all ranks call it)
 *   per all-pairs call: storm_hip_comm_allreduce_u64(ctx, comm, &value)   value := sum over ranks, or
 *     storm_hip_pairw_dense_begin(...) + storm_hip_comm_allreduce_result(ctx, comm, &total): the shard's
 *     partial is reduced where the pass left it (the context's result word), one host wait for both.
 * tools/storm_benchmark.cpp --ranks N is a complete example (fork before any HIP call, id through a pipe). */
#define STORM_HIP_COMM_ID_BYTES 128
typedef struct storm_hip_comm_s storm_hip_comm_t;
/* What the last all-pairs pass of this context ran: out[0] = mask of STORM_HIP_RAN_*; out[1] = 64-bit word pairs
 * multiplied by the dense kernels (pairs x words, this shard's share); out[2] = positions streamed by the
 * list-probe kernel (= its lookups: one 2-byte LDS read + one add each); out[3] = rows of the group one lookup
 * stands for (128: a lookup is 128 of the reference's per-pair list tests, storm.c:4-73, folded into a count).
 * For a harness that prices a row against the roof of the kernel that ran (tools/storm_benchmark.cpp). */
#define STORM_HIP_RAN_POPCOUNT 1u   /* pairw_dense_kernel (VALU popcount)                 roof: VALU issue     */
#define STORM_HIP_RAN_FP4_TILES 2u  /* pairw_fp4_kernel on the FP4 shadow                 roof: FP4 matrix cores */
#define STORM_HIP_RAN_FP4_STRIPS 4u /* strip16_fp4_kernel on the FP4 shadow               roof: FP4 matrix cores */
#define STORM_HIP_RAN_BITSTREAM 8u  /* bitstream_kernel (K2q), bit operands               roof: FP4 matrix cores */
#define STORM_HIP_RAN_BIT_STRIPS 16u /* strip16_bits_kernel (K2b), bit operands (default) roof: FP4 matrix cores */
#define STORM_HIP_RAN_LIST_PROBE 32u /* probe_lists_kernel (K4), list x list blocks       roof: LDS lookups    */
#define STORM_HIP_RAN_TILES_OUT 128u  /* tilering_kernel / tilebits8_kernel: per-pair output of a dense matrix (or replica) roof: FP4 matrix cores */
#define STORM_HIP_RAN_LISTS_MATRIX 64u /* lists_matrix_kernel (K5), per-pair output from the lists: out[2] = its table lookups, out[3] = 64 */
int storm_hip_last_pass_report(storm_hip_ctx_t* ctx, uint64_t out[4]);

int storm_hip_comm_unique_id(uint8_t id[STORM_HIP_COMM_ID_BYTES]);
int storm_hip_comm_init_rank(storm_hip_ctx_t* ctx, const uint8_t id[STORM_HIP_COMM_ID_BYTES], uint32_t rank,
                             uint32_t world, storm_hip_comm_t** out);
int storm_hip_comm_allreduce_u64(storm_hip_ctx_t* ctx, storm_hip_comm_t* comm, uint64_t* value);
/* up to 8 words in one collective (e.g. {partial, failure flag}: a failed rank still enters it) */
int storm_hip_comm_allreduce_u64s(storm_hip_ctx_t* ctx, storm_hip_comm_t* comm, uint64_t* values, uint32_t n);
int storm_hip_comm_allreduce_result(storm_hip_ctx_t* ctx, storm_hip_comm_t* comm, uint64_t* total);
uint32_t storm_hip_comm_rank(const storm_hip_comm_t* comm);
uint32_t storm_hip_comm_world(const storm_hip_comm_t* comm);
void storm_hip_comm_destroy(storm_hip_comm_t* comm);

/* ---- sparse (STORM_t) arena: flattened rows -> blocks (storm.h:157-178) --------------
 * Host-side flat description of all rows' 65536-bit blocks:
 *   row_block_offset[n_rows+1]   CSR over blocks
 *   block_id[n_blocks]           ascending within a row (storm.c:711-723)
 *   block_kind[n_blocks]         0 = uint16 list, 1 = 1024-word bitmap (storm.c:745-749)
 *   block_data_offset[n_blocks]  offset into list_pool (uint16 units) or bitmap_pool (words)
 *   block_n[n_blocks]            list length (kind 0)
 * Replaces STORM_pairw_intersect_cardinality[_blocked] (storm.c:877-961) and the per-pair
 * dispatch STORM_bitmap_cont_intersect_cardinality_premade / STORM_bitmap_intersect_
 * cardinality_func (storm.c:790-814, :618-656). */
int storm_hip_sparse_create(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                            const uint64_t* row_block_offset, const uint32_t* block_id,
                            const uint8_t* block_kind, const uint64_t* block_data_offset,
                            const uint32_t* block_n, const uint16_t* list_pool,
                            uint64_t list_pool_len, const uint64_t* bitmap_pool,
                            uint64_t bitmap_pool_words, storm_hip_sparse_t** out);
/* the same arena from a serialized STORM_t (STORM_serialize, storm.h; sizes as reference
 * storm.c:372-394): the host walks the headers only, the payload bytes go up as they are and both
 * block kinds are unpacked on the device. `buf` 2-byte aligned. */
/* The same from per-block POINTERS into the caller's own containers (what storm.h's STORM_t handles use: nothing
 * is flattened on the host): block_ptr[b] = the block's sorted uint16 list (block_n[b] entries, kind 0; 2-byte
 * aligned) or its 1024 words (kind 1; any alignment). The library walks the block headers, ships the raw lists
 * and bitmaps through a pinned staging ring and lays the list elements out on the device. */
int storm_hip_sparse_create_blocks(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                   const uint64_t* row_block_offset, const uint32_t* block_id,
                                   const uint8_t* block_kind, const uint32_t* block_n,
                                   const void* const* block_ptr, storm_hip_sparse_t** out);
/* [r6] Block stage: the bitmap blocks of a container travel to the device while the caller is still building it
 * (storm.h's STORM_add hands every bitmap block it finishes to one), so that the first all-pairs call — the one call
 * the reference's harness times, benchmark.cpp:605-613 — does not carry them over the bus. storm_hip_stage_add copies
 * the block's 1024 words into a pinned ring (sent 4 MiB at a time into 64 MiB device chunks) and returns its token;
 * storm_hip_sparse_create_blocks_staged builds the arena of storm_hip_sparse_create_blocks with the pool rows gathered
 * from the stage when EVERY bitmap block carries a valid token (token[b] < storm_hip_stage_count), and from block_ptr
 * otherwise; list blocks: a token of storm_hip_stage_add_list or ~0 (from block_ptr). The stage may be destroyed once the arena exists. */
typedef struct storm_hip_stage_s storm_hip_stage_t;
int storm_hip_stage_create(storm_hip_ctx_t* ctx, storm_hip_stage_t** out);
int storm_hip_stage_add(storm_hip_ctx_t* ctx, storm_hip_stage_t* stage, const uint64_t* words, uint64_t* token);
/* The same for a LIST block (kind 0: n ascending 16-bit positions, 1 <= n <= 65536): the token is the list's place in the
 * stage and goes into token[b] of storm_hip_sparse_create_blocks_staged; a list block whose token is ~0 travels from
 * block_ptr[b] at build time, as in storm_hip_sparse_create_blocks (list and bitmap tokens are separate sequences). */
int storm_hip_stage_add_list(storm_hip_ctx_t* ctx, storm_hip_stage_t* stage, const uint16_t* list, uint32_t n, uint64_t* token);
uint64_t storm_hip_stage_count(const storm_hip_stage_t* stage);
void storm_hip_stage_destroy(storm_hip_ctx_t* ctx, storm_hip_stage_t* stage);
int storm_hip_sparse_create_blocks_staged(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                          const uint64_t* row_block_offset, const uint32_t* block_id,
                                          const uint8_t* block_kind, const uint32_t* block_n,
                                          const void* const* block_ptr, storm_hip_stage_t* stage, const uint64_t* token,
                                          storm_hip_sparse_t** out);
int storm_hip_sparse_create_serialized(storm_hip_ctx_t* ctx, const void* buf, uint64_t n_bytes,
                                       storm_hip_sparse_t** out);
/* The rows of such a container as a DENSE bit matrix on the device (row width = 65536 x (largest block id + 1) bits,
 * at most 2^25): what the per-pair output of a STORM_t runs on (storm.h: STORM_pairw_matrix) — a row pair of the
 * reference (block-id merge + 4-way kind dispatch, storm.c:790-814, :618-656) is popcount(row_i & row_j) over
 * exactly these bits. Same block description as storm_hip_sparse_create_blocks; blocks travel through the pinned
 * ring as they lie in the containers and are unpacked by the device. Destroy with storm_hip_matrix_destroy. */
int storm_hip_matrix_create_from_blocks(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                        const uint64_t* row_block_offset, const uint32_t* block_id,
                                        const uint8_t* block_kind, const uint32_t* block_n,
                                        const void* const* block_ptr, storm_hip_matrix_t** out);
void storm_hip_sparse_destroy(storm_hip_ctx_t* ctx, storm_hip_sparse_t* s);
/* [r5] K5 — the per-pair matrix of a LIST-ONLY container straight from its lists (replaces, for every pair at once,
 * STORM_bitmap_cont_intersect_cardinality, storm.c:790-814, with two list blocks meeting in
 * STORM_intersect_vector16_cardinality, storm.c:4-73). Same block description as storm_hip_sparse_create_blocks.
 * *out stays NULL (and the call returns STORM_HIP_OK) when the container is not eligible — a bitmap block, a row of
 * more than 65535 positions, more than 2^26 positions in all: the dense replica (storm_hip_matrix_create_from_blocks)
 * is the path then. _worthwhile: 1 when the lists are expected to beat the dense replica's multiply (l = NULL: 1 unless the
 * path is switched off, i.e. whether building the lists is worth a try; option
 * `matrix_lists`: -1 by density, 0 never, 1 whenever eligible; `matrix_lists_density`: the crossover in 1/10000).
 * _pairw_matrix_device: out[i * ld + j] = popcount(row_i OP row_j) for i < j, device memory, complete on return;
 * entries i >= j are not written. */
int storm_hip_rowlists_create_blocks(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                     const uint64_t* row_block_offset, const uint32_t* block_id,
                                     const uint8_t* block_kind, const uint32_t* block_n,
                                     const void* const* block_ptr, storm_hip_rowlists_t** out);
/* ... with the lists taken from a block stage (storm_hip_stage_add_list) when EVERY non-empty block carries a token
 * (token[b] != ~0); from block_ptr otherwise. The stage is only read. */
int storm_hip_rowlists_create_blocks_staged(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                            const uint64_t* row_block_offset, const uint32_t* block_id,
                                            const uint8_t* block_kind, const uint32_t* block_n,
                                            const void* const* block_ptr, storm_hip_stage_t* stage, const uint64_t* token,
                                            storm_hip_rowlists_t** out);
void storm_hip_rowlists_destroy(storm_hip_ctx_t* ctx, storm_hip_rowlists_t* l);
int storm_hip_rowlists_worthwhile(storm_hip_ctx_t* ctx, const storm_hip_rowlists_t* l);
/* the same rule from the counts alone (rows, listed positions, bits per row), BEFORE anything is built: a container the
 * rule sends to the dense replica should not pay for row lists it will not use (9 ms of a first call at 3670 positions
 * per row of BASELINE c4's shape) */
int storm_hip_rowlists_worthwhile_counts(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_elems, uint64_t n_bits);
int storm_hip_rowlists_pairw_matrix_device(storm_hip_ctx_t* ctx, const storm_hip_rowlists_t* l, int op,
                                           uint32_t* d_out, uint64_t ld);
/* ... and into host memory: whole rows, zeros at i >= j (what storm_hip_pairw_matrix writes) */
int storm_hip_rowlists_pairw_matrix(storm_hip_ctx_t* ctx, const storm_hip_rowlists_t* l, int op, uint32_t* h_out,
                                    uint64_t ld);
uint64_t storm_hip_rowlists_n_elems(const storm_hip_rowlists_t* l);

int storm_hip_pairw_sparse(storm_hip_ctx_t* ctx, const storm_hip_sparse_t* s,
                           uint32_t shard_rank, uint32_t shard_count, uint64_t* h_total);
/* split form (one host thread, one arena replica per GPU): _begin launches this shard into the
 * context's result word, _end waits for it and copies it to the host */
int storm_hip_pairw_sparse_begin(storm_hip_ctx_t* ctx, const storm_hip_sparse_t* s,
                                 uint32_t shard_rank, uint32_t shard_count);
int storm_hip_pairw_sparse_end(storm_hip_ctx_t* ctx, uint64_t* h_total);
/* work census of the last sparse call: out[0]=list×list block pairs, [1]=list×bitmap,
 * [2]=bitmap×bitmap, [3]=block columns */
int storm_hip_sparse_last_census(storm_hip_ctx_t* ctx, uint64_t out[4]);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* STORM_HIP_H_ */
