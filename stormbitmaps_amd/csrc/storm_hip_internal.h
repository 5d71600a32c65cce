// storm_hip_internal.h — shared between the .hip translation units of libstorm_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <cstring>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#include "storm_hip.h"

struct storm_hip_ctx_s;
typedef struct storm_hip_ctx_s storm_hip_ctx_t;
struct storm_hip_matrix_s;

namespace storm {

void set_error(const char* fmt, ...);

#define STORM_HIP_TRY(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            ::storm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                               __FILE__, __LINE__);                                          \
            return STORM_HIP_EHIP;                                                           \
        }                                                                                    \
    } while (0)

// No C++ exception may cross the C boundary: every extern "C" entry point that can reach a
// std::vector or operator new runs its body through this (bad_alloc -> STORM_HIP_ENOMEM).
template <typename Fn>
static inline int guarded(const char* what, Fn&& fn) noexcept {
    try {
        return fn();
    } catch (const std::bad_alloc&) {
        set_error("%s: out of host memory", what);
        return STORM_HIP_ENOMEM;
    } catch (const std::exception& e) {
        set_error("%s: %s", what, e.what());
        return STORM_HIP_EHIP;
    } catch (...) {
        set_error("%s: unknown C++ exception", what);
        return STORM_HIP_EHIP;
    }
}

// Geometry of the dense kernel (see DESIGN.md "K1").
constexpr int kLanes = 64;                           // gfx950 wavefront
constexpr int kWaves = 4;                            // waves per workgroup
constexpr int kThreads = kLanes * kWaves;            // 256
constexpr int kChunkWords = 64;                      // k-chunk: one 64-bit word per lane
constexpr int kRowsPerWave = 32;                     // A rows held in VGPRs by one wave
constexpr int kABlockRows = kWaves * kRowsPerWave;   // 128 A rows per workgroup
constexpr int kRowPad = 256;                         // allocated rows: a multiple of this, zero beyond n_rows (a strip's A tile)
constexpr int kStageRows = 32;                       // B rows per LDS stage
constexpr int kSlots = 4096;  // partial-sum slots (uint64 each)
constexpr int kSlotsExtra = 8;  // words behind the slots: work-queue heads, zeroed by the fold

struct Seg {            // one (A block, B row range) segment of the upper triangle
    uint32_t a_row0;    // first A row of the block (kABlockRows rows are loaded from here)
    uint32_t a_end;     // A rows >= a_end are treated as all-zero (block-column / matrix edge)
    uint32_t j_lo;      // B rows [j_lo, j_hi)
    uint32_t j_hi;      // j_lo == a_row0 marks a diagonal segment: count only pairs i < j
};

// shared launcher of the dense kernel over an arbitrary segment table (dense + sparse paths)
int launch_pairw_segments(storm_hip_ctx_t* ctx, const uint64_t* X, uint64_t stride_words,
                          const Seg* d_segs, uint32_t n_segs, uint64_t seg_row_sum,
                          uint64_t* d_total);

}  // namespace storm

struct storm_hip_ctx_s {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cus = 0;
    // workspace
    unsigned long long* d_slots = nullptr;   // kSlots partial sums
    unsigned long long* d_scalar = nullptr;  // one uint64 result
    unsigned long long* h_scalar = nullptr;  // pinned host word the result is read back through
    // [r5] result mailbox: a pinned, device-visible host word the LAST kernel of a synchronous all-pairs call writes the
    // total into directly; the host polls it (sentinel ~0 = not there yet) instead of queueing an 8-byte copy behind the
    // kernel and sleeping in hipStreamSynchronize — 5 - 7 us of a call that is 18 - 30 us at the sparse end and at LD-window
    // row counts. Option result_mailbox (1 on / 0 off: the copy + synchronize of rounds 1 - 4).
    unsigned long long* h_mail = nullptr;
    unsigned long long* d_mail = nullptr;    // the same word as the device sees it
    int result_mailbox = 1;
    int sync_poll_us = 0;                    // [r6] synchronous matrix-output calls poll hipStreamQuery this long before they park in hipStreamSynchronize (0: park at once)
    bool mail_armed = false;                 // the call in flight was launched into the mailbox
    // [r6] device / pinned allocations whose release (hipFree waits for the device: ~0.2 ms each, eleven of them behind an
    // arena build) is put off until the call AFTER the one that made them obsolete, or the context's end
    std::vector<void*> deferred_free, deferred_host_free;
    uint32_t deferred_age = 0;
    hipEvent_t stage_ev[3] = {nullptr, nullptr, nullptr};   // the ring's rotation lives in the context: one Stager after another continues it and waits for the copy that last left a buffer
    bool stage_used[3] = {false, false, false};
    int stage_next = 0;
    void* h_stage_ring = nullptr;            // pinned staging ring of the sparse arena builder (storm_hip_sparse.hip: Stager), allocated on first use
    storm::Seg* d_segs = nullptr;            // segment table of the last geometry
    size_t segs_capacity = 0;
    // cache key of d_segs
    uint64_t seg_rows_n = 0;
    uint32_t seg_shard_rank = 0, seg_shard_count = 0, seg_len = 0;
    uint32_t n_segs = 0;
    uint64_t seg_row_sum = 0;  // sum over segments of (j_hi - j_lo)
    // options
    int variant = -1;       // -1 auto, 0/1/2 popcount kernel (B path), 3 MFMA tiles, 4 MFMA strips
    int variant_used = 2;   // what the last dense launch ran
    int probe_bundle = -1;  // [r6] list-probe kernel: -1 / 1 = one group of 128 rows per workgroup (probe_lists_kernel), 4 = bundles of four groups per workgroup (probe_lists_fat_kernel: a quarter of the loads and workgroups, measured 5 - 25 % SLOWER at every c4 load, profiles/r06_*_probe_bundle.txt)
    int sparse_probe = -1;  // sparse container: list-probe kernel for columns of short lists (-1 auto, 0 never, 1 always)
    int matrix_lists = -1;  // per-pair matrix of a list-only sparse container from its lists (K5, storm_hip_lists.hip): -1 by density, 0 never, 1 whenever eligible
    int matrix_lists_kernel = 0;  // K5: 0 = by the row length, 1 = lists_matrix_kernel (windows), 2 = lists_hash_kernel where a group fits the table
    int matrix_lists_hash_min_log2 = 6;  // ... the hash kernel from this group size on (log2 of 8 .. 64 rows)
    int matrix_lists_debug = 0;   // timing ablations of lists_matrix_kernel (wrong results): 1 no element loads, 4 no barriers, 8 no stores, 16 no steps
    int matrix_lists_permille_x10 = 80;  // ... the crossover density in 1/10000 of the dense replica's bits ([r6] 60 -> 80: the window kernel takes 5.2 ms at 0.8 % where the dense replica takes 6.4; level at 0.95 %)
    int seg_rows = 256;
    int chunks_per_item = 0;
    // info of the last dense launch
    uint64_t last_info[4] = {0, 0, 0, 0};
    uint64_t sparse_census[4] = {0, 0, 0, 0};
    // What the last all-pairs pass ran (storm_hip_last_pass_report): [0] kernels, a mask of STORM_HIP_RAN_*;
    // [1] 64-bit word pairs multiplied on the matrix cores or by the popcount kernel (this shard's share,
    // algorithmic: pairs x words); [2] positions the list-probe kernel streamed = its lookups; [3] rows one
    // lookup stands for (128). A harness prices every row against the roof of the kernel that ran.
    uint64_t pass_report[4] = {0, 0, 0, 0};
    // K2 (MFMA FP4) state: nibble-expanded shadow of the matrix + item table
    uint8_t* d_x4 = nullptr;
    size_t x4_capacity = 0;
    void* d_items = nullptr;
    size_t items_capacity = 0;
    uint64_t items_key[4] = {0, 0, 0, 0};  // rows, stages, shard rank/count, stages per item
    uint32_t n_items = 0;
    void* panel_lists = nullptr;     // work lists of the row panels of storm_hip_pairw_dense_upload (std::vector<PanelList>*)
    hipStream_t copy_stream = nullptr;   // ... its copies travel here while ctx->stream multiplies the panel before
    void* d_strip_items = nullptr;
    size_t strip_capacity = 0;
    uint64_t strip_key[4] = {0, 0, 0, 0};
    uint32_t n_strip_items = 0;
    int k2_stages_per_item = 32;
    int k2_max_run = 0;     // strips: B stages per item at most; 0 = 64 / 96 / 128, whichever list schedules shortest (ensure_strip_items)
    int k2_ring = 4;        // K2s: LDS ring depth (3, 4 or 5)
    int k2_shadow_budget_mb = 96 * 1024;  // K2s: FP4 shadow above this many MiB -> k-chunked passes (0 = never)
    int k2_strip_operands = 0;  // strips: 0 = K2b; 5 = bit operands, FP4 image of every B stage built in the LDS (strip16_bits_kernel, K2b); 2 = bit operands, one stage stream per workgroup, one launch (bitstream_kernel, K2q); 4 = FP4 shadow (strip16_fp4_kernel / strip_fp4_kernel); 1 = bit operands, one item per workgroup (stripbits_kernel)
    int k2_stream_max_rows = 8192;  // auto: matrices up to this many rows take K2q
    int k2_shard_pairs = 0;         // ownership among shards: 0 = whole k-slices first (leftover slices along the pair space), 1 = every slice along the pair space
    int k2_matrix_pad = -1;         // matrices created from now on: rows that are a multiple of 1 KiB get this many 512-byte chunks more of pitch (0: dense pitch; -1: by the pitch, pitch_pad_chunks)
    int k2_fold_inline = -1;        // K2b: the workgroup dispatched last folds the partial sums inside the launch: -1 = for short launches (<= 4096 workgroups, where the fold launch and its gaps are a fifth of a pass), 1 = always (level at N = 10000), 0 = never (a fold launch behind the strips); profiles/r05_c_fold_ab.jsonl
    int k2_operands_used = 4;       // what the last strip launch ran (1, 2 or 4)
    int k2_tile_shape = 0;  // write-mode tile kernel: 0 = by the matrix (5 for a dense matrix, 2 for the dense replica of a sparse container: sparse operands let tilebits8_kernel, which sits at the socket's power cap, clock higher; crossover near 20 % density, profiles/r05_g_*); 5 = tilering_kernel (both operands as FP4 images in the LDS, 16x16x128); 2 = tilebits8_kernel (bit operands inflated in registers, 32x32x64); 3 / 4 = K2tb; 1 / 16 / 32: tools build
    int k2_wave_below = 400;      // [r6] k2_tile_shape 0: matrices (bands, rectangles) of fewer 256 x 256 tiles than this take tile128_kernel (K2h: 128 x 128 tiles, cut along k where they are too few, the parts' sums meeting inside the launch; no window clearing, no atomics into the output)
    int k2_part_slots = 0;        // K2h: segments per CU the tiles' chunk stream is cut into: 0 = two once that leaves segments of 4 x k2_part_min_chunks, else one; 1 / 2: forced
    int k2_part_min_chunks = 8;   // K2h: a k-part is at least this many 512-bit chunks
    int k2_part_narrow = 1;       // K2h: windows of 16-bit counts where every part of a tile covers fewer than 2^16 bits of k (half the bytes the part that ends the tile has to read)
    int k2_part_cost_diag = 80;   // K2h: what a chunk of a tile on the diagonal costs next to one of a full tile, percent
    uint32_t n_part_items = 0;    // items of the K2h list cached in d_items (items_key)
    uint32_t* d_tickets = nullptr;   // K2h: one arrival counter per tile (zero between launches)
    size_t tickets_capacity = 0;
    bool tickets_dirty = false;   // a launch failed: clear the tickets before the next one
    int k2_tile_shape_eff = 2;  // what the call in flight runs (set by launch_pairw_matrix / launch_square_matrix)
    int k2_ring_sync = 0;   // tilering_kernel: 0 = one s_barrier per stage; 1 = arrival counters in the LDS (waves may drift a stage apart; measured 2 % slower)
    int k2_ring_cost_diag = 78, k2_ring_cost_ragged = 40;  // the same for tilering_kernel (k2_tile_shape = 5)
    int k2_tile_cost_diag = 63, k2_tile_cost_ragged = 30;  // percent of a full tile (tilebits8_kernel): what the planner assumes when it cuts the last round
    int k2_shape = 16;      // K2s: MFMA shape of the default strip kernel: 16 = 16x16x128 (default), 32 = 32x32x64
    int k2_persistent = 0;  // K2s: workgroups pull items from per-XCD queues (0: one item per workgroup)
    uint32_t strip_queue_base[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t strip_queue_count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // "keep_shadow": reuse the FP4 shadow across all-pairs calls while (matrix, generation, shard,
    // layout) are unchanged. Only valid if the matrix is modified through the library alone.
    int keep_shadow = 0;
    uint64_t x4_key[4] = {0, 0, 0, 0};
    int k2_matrix_split = 1; // matrix output: cut the last round's tiles along k to fill the CUs
    int k2_matrix_parts = 0;  // ... 0: the parts add into the cleared output with atomics; 1: every part writes its own window and reduce_parts_kernel adds them up (round 5: the tile kernel gets 10 us faster at 1024 rows, the second kernel costs more than the clearing and the atomics did: profiles/r05_k_matrix_sizes.jsonl)
    uint32_t* d_parts = nullptr;   // the parts' windows (256 x 256 uint32 each)
    size_t parts_capacity = 0;
    int k2_matrix_min_part = 32;  // ... into parts of at least this many 128-bit stages (a multiple of 4)
    int k2_pitch_pad = -1;  // K2/K2s: extra bytes per row of the FP4 shadow (multiple of 128; -1 = auto)
    int k2_lds_pad = 0;     // K2s: bytes of unused dynamic LDS per workgroup (caps workgroups per CU)
    int k2_lpt_rounds = 6;  // K2s: XCD lists of at most this many rounds of 128 items are sorted longest-first
    int k2_tail_slices = 3; // K2s: last slices of every XCD's list are cut into short runs ...
    int k2_tail_run = 32;   //      ... of at most this many stages, merged longest-first
    // "time_kernels" option: HIP event pairs around the dominant kernel of every pairwise launch
    // (the popcount kernel, the tile kernel or the strip kernel), read by storm_hip_kernel_time
    int time_kernels = 0;
    std::vector<hipEvent_t> kernel_events;  // begin/end alternating
    size_t kernel_events_used = 0;
    uint32_t* d_band = nullptr;    // device staging of the host-output matrix calls (band x n_rows uint32)
    size_t band_capacity = 0;
    void* d_positions = nullptr;   // staging of storm_hip_matrix_set_rows_from_positions: offsets, then positions
    size_t positions_capacity = 0;
    uint32_t* d_counts = nullptr;  // row-count scratch of the matrix-output paths
    size_t counts_capacity = 0;
    unsigned long long* d_trace = nullptr;  // k2_ring = 18: per-item schedule trace of the strip kernel
    size_t trace_capacity = 0;
    uint32_t trace_items = 0;
    // K2q (bitstream_kernel): segment table + per-workgroup bounds of the last geometry
    void* d_bitsegs = nullptr;
    size_t bitsegs_capacity = 0;
    void* d_bitfirst = nullptr;
    size_t bitfirst_capacity = 0;
    uint64_t bit_key[4] = {0, 0, 0, 0};
    uint32_t n_bit_groups = 0, n_bit_segs = 0, bit_max_stages = 0;
    uint64_t bit_stages = 0;
    int k2_stream_groups_per_cu = 0;  // K2q: workgroups per CU (0 = by the length of the stream: 1, 2 or 3)
    int k2_stream_min_piece = 6;      // K2q: stages a workgroup should at least have before a CU's share is cut further
    int k2_stream_min_run = 2;        // K2q: a cut leaves at least this many later blocks on either side
    // K2q: length of the shares of the second / third workgroup of a CU in percent of the first one's, when the
    // stream is one round of 3 workgroups per CU (build_bitstream: the SIMD arbiter serves the oldest wave)
    int k2_stream_w3_1 = 120, k2_stream_w3_2 = 60;
    int k2_wave_ring = 0;             // K2w: stages of a wave's private ring (0 = by workgroups per CU: 8 / 4 / 3)
    bool trace_is_stream = false;     // d_trace holds per-workgroup words of bitstream_kernel (no strip items)
    int k2_debug = 0;  // timing probes (wrong results): 1 = all items on tile (0,0), 2 = no XCD grouping
};

namespace storm {
// K2: all-pairs total of a dense matrix through v_mfma_f32_32x32x64_f8f6f4 (storm_hip_mfma.hip)
int launch_pairw_mfma(storm_hip_ctx_t* ctx, const storm_hip_matrix_s* m, uint32_t shard_rank,
                      uint32_t shard_count, uint64_t* d_total);
struct RowRange {
    uint64_t r0, r1;     // rows [r0, r1) form one all-pairs problem; r0 is a multiple of the A tile (256; 512 for wide strips)
    uint64_t a_end = 0;  // 0: the whole triangle. Otherwise only the pairs with the EARLIER row below a_end (a
                         // multiple of the A tile from r0, or >= r1): the rows [r0, a_end) among themselves and
                         // against everything behind them — a block column's bitmap rows, with its list rows,
                         // which the list-probe kernel pairs with each other, behind them
    uint64_t back_from = ~0ull;  // != ~0: a row PANEL [back_from, r1) that has just arrived (a multiple of the A tile): only the
                                 // pairs whose LATER row lies in the panel — every A tile of the panel against all the blocks in
                                 // front of it (from r0) plus its own triangle (popcount(a & b) is symmetric: the new rows are the
                                 // stationary operand, the rows already there stream past in long runs)
};
int launch_pairw_mfma_ranges(storm_hip_ctx_t* ctx, const uint64_t* X, uint64_t stride_words,
                             uint64_t n_rows_src, uint64_t n_rows_dst,
                             const std::vector<RowRange>& ranges, uint32_t shard_rank,
                             uint32_t shard_count, int strip_mode, uint64_t* d_total,
                             uint64_t shadow_generation = 0, uint32_t n_words_logical = 0);
int launch_pairw_bits_upload(storm_hip_ctx_t* ctx, storm_hip_matrix_s* m, const uint64_t* host_rows,
                             uint64_t src_stride_words, uint64_t* d_total);   // storm_hip_mfma.hip
int stage_gather_lists(storm_hip_ctx_t* ctx, storm_hip_stage_s* stage, const std::vector<uint64_t>& ltable, uint16_t* d_lists,
                       uint64_t** d_table);   // storm_hip_sparse.hip: staged list blocks -> a block-ordered list buffer
int strip_operands_of(const storm_hip_ctx_t* ctx);   // 5 = K2b, 4 = FP4 strips, ... (storm_hip_mfma.hip)
int launch_pairw_bits_ranges(storm_hip_ctx_t* ctx, const uint8_t* X, uint64_t pitch_bytes,
                             const std::vector<RowRange>& ranges, uint32_t n_kslices2, uint32_t shard_rank,
                             uint32_t shard_count, uint64_t* d_total, bool slots_hold_sums = false,
                             uint32_t a_tile = 256u);   // a_tile 512: strip16_bits2_kernel (the last tile's rows up to the multiple of 512 must exist and be zero)
// pairs x words of a set of row ranges, divided among shard_count shards (the report's algorithmic word pairs)
static inline uint64_t ranges_word_pairs(const std::vector<RowRange>& ranges, uint64_t words, uint32_t shard_count) {
    uint64_t pairs = 0;
    for (const RowRange& rg : ranges) {
        const uint64_t n = rg.r1 - rg.r0, a = rg.a_end ? std::min(rg.a_end, rg.r1) - rg.r0 : n;
        pairs += a * (a - (a != 0)) / 2 + a * (n - a);
    }
    return pairs * words / (shard_count ? shard_count : 1);
}
int launch_pairw_matrix(storm_hip_ctx_t* ctx, const storm_hip_matrix_s* m, int op, uint32_t* d_out,
                        uint64_t ld, uint64_t band_row0 = 0, uint64_t band_rows = ~0ull,
                        bool sync = true);
int launch_square_mfma(storm_hip_ctx_t* ctx, const storm_hip_matrix_s* a,
                       const storm_hip_matrix_s* b, uint64_t* d_total);
int launch_square_matrix(storm_hip_ctx_t* ctx, const storm_hip_matrix_s* a,
                         const storm_hip_matrix_s* b, int op, uint32_t* d_out, uint64_t ld);
int launch_row_counts(storm_hip_ctx_t* ctx, const storm_hip_matrix_s* m, uint32_t* d_counts);
void release_mfma_state(storm_hip_ctx_t* ctx);
// releases what was put off (storm_hip_ctx_s::deferred_free); `aged`: only what an earlier call left behind
void drain_deferred(storm_hip_ctx_t* ctx, bool aged);
// one empty launch per translation unit (storm_hip_ctx_create): loads the TU's code object ahead of the first real call
void warm_mfma_code(hipStream_t stream);
void warm_sparse_code(hipStream_t stream);
void warm_lists_code(hipStream_t stream);
// ctx->d_scalar -> *h_total through the context's pinned word; synchronises the stream (storm_hip.hip)
int fetch_result_word(storm_hip_ctx_t* ctx, uint64_t* h_total);
// where a synchronous call launches its total: the mailbox (armed with the sentinel) or, without one, ctx->d_scalar
uint64_t* result_target(storm_hip_ctx_t* ctx);
// the armed mailbox's value once it has arrived (polls; falls back to a stream synchronize); STORM_HIP_EHIP if it never does
int wait_mailbox(storm_hip_ctx_t* ctx, uint64_t* value);
// folds ctx->d_slots into *d_total (device pointer) and re-zeroes the slots (storm_hip.hip)
int launch_fold_slots(storm_hip_ctx_t* ctx, uint64_t* d_total);
// record the next event of the "time_kernels" series on the launch stream (no-op when off)
void kernel_time_mark(storm_hip_ctx_t* ctx);
uint64_t next_matrix_generation();
// Host data -> device through a small ring of pinned buffers: the pieces of a chunk are packed by a few threads
// while the previous chunk's copy is in flight (pageable uploads of 1.7 GB were 0.15 s of c4's first call).
struct Piece { const void* src; size_t bytes; size_t dst_off; size_t zero_after = 0; };  // dst_off: byte offset from the chunk's device base; zero_after: bytes of zeros behind the piece
struct Stager {
    static constexpr size_t kBuf = 8u << 20;  // 8 MiB per buffer: 0.16 ms of PCIe time each
    static constexpr int kBufs = 3;
    storm_hip_ctx_t* ctx;
    hipEvent_t (&ev)[kBufs];   // (state of the context: see stage_ev)
    bool (&used)[kBufs];
    int& next;
    explicit Stager(storm_hip_ctx_t* c) : ctx(c), ev(c->stage_ev), used(c->stage_used), next(c->stage_next) {}
    static_assert(kBuf * kBufs == ((size_t)24 << 20), "storm_hip_ctx_reserve_staging allocates the same ring");
    int init() {
        if (!ctx->h_stage_ring) {
            if (hipHostMalloc(&ctx->h_stage_ring, kBuf * kBufs, hipHostMallocDefault) != hipSuccess) {
                ctx->h_stage_ring = nullptr;
                set_error("sparse_create: hipHostMalloc of the %zu-byte staging ring failed", kBuf * kBufs);
                return STORM_HIP_ENOMEM;
            }
        }
        for (int i = 0; i < kBufs; ++i)
            if (!ev[i] && hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return STORM_HIP_EHIP;
        return STORM_HIP_OK;
    }
    // pieces[p0, p1): consecutive in the device destination (dst_off ascending from 0, no gaps beyond `span`)
    int send(uint8_t* d_base, const Piece* pieces, size_t n, size_t span) {
        uint8_t* buf = static_cast<uint8_t*>(ctx->h_stage_ring) + (size_t)next * kBuf;
        if (used[next] && hipEventSynchronize(ev[next]) != hipSuccess) return STORM_HIP_EHIP;
        const unsigned parts = span >= (1u << 20) ? 4u : 1u;   // (2 / 4 / 8 / 12 packers: 31 / 24 / 28 / 32 ms for c4's 420 MB of lists)
        auto pack = [&](unsigned part) {
            for (size_t i = n * part / parts; i < n * (part + 1) / parts; ++i) {
                memcpy(buf + pieces[i].dst_off, pieces[i].src, pieces[i].bytes);
                if (pieces[i].zero_after) memset(buf + pieces[i].dst_off + pieces[i].bytes, 0, pieces[i].zero_after);
            }
        };
        if (parts > 1) {
            std::thread helpers[3];
            for (unsigned t = 1; t < parts; ++t) helpers[t - 1] = std::thread(pack, t);
            pack(0);
            for (unsigned t = 1; t < parts; ++t) helpers[t - 1].join();
        } else {
            pack(0);
        }
        if (hipMemcpyAsync(d_base, buf, span, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipEventRecord(ev[next], ctx->stream) != hipSuccess)
            return STORM_HIP_EHIP;
        used[next] = true;
        next = (next + 1) % kBufs;
        return STORM_HIP_OK;
    }
    // Rows of `stride` bytes from d_base, the first `bytes` of each from one host piece, zeros behind (whole rows per
    // chunk: stride <= kBuf).
    int send_rows(uint8_t* d_base, const std::vector<const void*>& rows, size_t bytes, size_t stride) {
        const size_t per_chunk = kBuf / stride;
        std::vector<Piece> chunk;
        for (size_t r0 = 0; r0 < rows.size(); r0 += per_chunk) {
            const size_t n = std::min(per_chunk, rows.size() - r0);
            chunk.clear();
            for (size_t i = 0; i < n; ++i) chunk.push_back({rows[r0 + i], bytes, i * stride, stride - bytes});
            if (int rc = send(d_base + r0 * stride, chunk.data(), n, n * stride)) return rc;
        }
        return STORM_HIP_OK;
    }
    // A run of pieces that is contiguous on the device from d_base (piece i starts where piece i - 1 ended),
    // cut into chunks of at most one buffer. A single piece larger than a buffer is cut as well.
    int send_run(uint8_t* d_base, const std::vector<std::pair<const void*, size_t>>& run) {
        std::vector<Piece> chunk;
        size_t chunk_base = 0, at = 0, fill = 0;
        auto flush = [&]() -> int {
            if (chunk.empty()) return STORM_HIP_OK;
            const int rc = send(d_base + chunk_base, chunk.data(), chunk.size(), fill);
            chunk.clear();
            chunk_base = at;
            fill = 0;
            return rc;
        };
        for (const auto& pc : run) {
            const uint8_t* src = static_cast<const uint8_t*>(pc.first);
            size_t left = pc.second;
            while (left) {
                if (fill == kBuf)
                    if (int rc = flush()) return rc;
                const size_t take = std::min(left, kBuf - fill);
                chunk.push_back({src, take, fill});
                src += take;
                left -= take;
                fill += take;
                at += take;
            }
        }
        return flush();
    }
};

// A host table -> device memory on the context's stream. Small ones as they are (the runtime stages them); from 512 KiB on
// through the pinned ring when the context has one: a large pageable source that is unmapped soon after an asynchronous copy —
// a std::vector freed at the end of a build — made the next wide kernel start 10 - 30 ms late (LAB_NOTES "the late kernel").
inline int upload_bytes(storm_hip_ctx_t* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (bytes == 0) return STORM_HIP_OK;
    if (bytes < (512u << 10) || !ctx->h_stage_ring)
        return hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess ? STORM_HIP_OK : STORM_HIP_EHIP;
    Stager stager(ctx);
    if (int rc = stager.init()) return rc;
    return stager.send_run(static_cast<uint8_t*>(d_dst), {{h_src, bytes}});
}


}  // namespace storm

// Pad of a row pitch that is a multiple of 1 KiB, in 512-byte chunks. The bit-operand strips (K2b) read 64-byte
// pieces of 64 consecutive rows: at a power-of-two pitch they fall into a handful of L2 sets and memory channels.
// Measured per pass (profiles/r04_a_pitch_pad.txt, r04_k_pitch_pad.jsonl), pad 0 / 1 / 2 / 4 chunks:
//   10000 rows x  8 KiB (the headline shape)   0.766 / 0.756 / 0.758 / 0.758 ms    (16 and 32 KiB rows: all the same)
//   10000 rows x 64 KiB                        6.57  / 6.22  / 5.97  / 5.99
//   20000 rows x 128 KiB                          -  / 46.1  / 57.9  / 46.0       (2^k + 2^(k-7) bytes is the bad pitch)
// so: one chunk below 64 KiB of pitch (6 % of memory at the headline shape), four from there on.
static inline uint64_t pitch_pad_chunks(int option, uint64_t stride_words) {
    if (option == 0 || stride_words % 128 != 0) return 0;
    return option > 0 ? (uint64_t)option : stride_words >= 8192 ? 4u : 1u;
}

struct storm_hip_matrix_s {
    uint64_t* d = nullptr;
    uint64_t n_rows = 0;        // logical rows
    uint64_t n_rows_pad = 0;    // allocated rows (multiple of kABlockRows)
    uint32_t n_words = 0;       // logical words per row
    uint64_t stride_words = 0;  // allocated words per row (multiple of kChunkWords)
    uint64_t generation = 0;    // changes with every mutation through the library (see "keep_shadow")
    bool sparse_origin = false; // the dense replica of a sparse container (storm_hip_matrix_create_from_blocks): mostly zeros
};
