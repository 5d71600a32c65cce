// storm_hip_lists.hip — K5: the per-pair matrix of a LIST-ONLY sparse container, from the lists themselves.
//
// What it replaces: STORM_bitmap_cont_intersect_cardinality (storm.c:790-814) called for every pair of rows — the block-id
// merge of storm.c:75-106 and, for two list blocks, STORM_intersect_vector16_cardinality (storm.c:4-73, kind dispatch
// :618-656). Until round 5 the device path of that matrix was a dense replica of the rows (N x M bits: 655 MB at the
// README's STORM_t shape) multiplied by the tile kernels whatever the density: 6.4 ms at 524 positions per row of 524288,
// where the rows hold 21 MB of positions between them.
//
// Formulation (a hash join per output tile, the table in the LDS):
//   data   : every listed position as ONE 32-bit element, (row & 255) << 13 | position & 8191, ordered by WINDOW of 8192
//            positions first and by GROUP of 64 rows inside a window; off[g][w] = first element of (window w, group g).
//            The elements of a group — or of four consecutive groups, a CHUNK of 256 rows — in a window are contiguous.
//   item   : one output tile = group gi (64 "A" rows) x chunk cj (256 "far" rows), all windows. One workgroup of 1024
//            threads, one item.
//   LDS    : table[8192 positions][2 planes] of 64-bit row masks (128 KiB): bit a of table[p][q] = "A row a lists position
//            p of the window that owns plane q"; counters[64][256] of 16 bits (32 KiB) = the tile.
//   step   : for every window in which both sides list something — look every far element up (one ds_read_b64; for
//            every set bit one 16-bit LDS increment: that is one intersecting position of one row pair), while the A
//            elements of the NEXT such window are toggled into the other plane and those of the PREVIOUS one out of it.
//            Set and clear are both XOR: they commute, so the two may run in any order within the step even when they
//            hit the same bit, and a step needs ONE barrier.
//   output : the tile's strict upper part is written once, row segments of 1 KiB; OR / XOR counts from the row lengths.
//            Nothing is zero-filled and nothing is added to in global memory; entries i >= j are not touched (as the tile
//            kernels leave them).
//   work   : lookups = N / 64 x elements / 2; increments = the matrix's sum. Both fall with the density, where the dense
//            multiply does not: see worthwhile() for the crossover.
// Eligible: every block a list, every row at most 65535 positions (16-bit counters), at most 2^26 elements, at most
// 2^24 (group, window) cells. Anything else keeps the dense replica.
#include "storm_hip_internal.h"

#include <cstring>
#include <memory>

using namespace storm;

namespace {

constexpr int kLmThreads = 1024;
constexpr uint32_t kLmGroup = 64;        // A rows per tile
constexpr uint32_t kLmChunkGroups = 4;   // far rows per tile: 4 groups = 256
constexpr uint32_t kLmChunk = kLmGroup * kLmChunkGroups;
constexpr uint32_t kLmWinBits = 13;
constexpr uint32_t kLmWin = 1u << kLmWinBits;
constexpr uint32_t kLmTableBytes = kLmWin * 16u;                 // 2 planes x 8 bytes per position
constexpr uint32_t kLmCountBytes = kLmGroup * kLmChunk * 2u;     // 16-bit counters
static_assert(kLmTableBytes + kLmCountBytes <= 160u * 1024u, "LDS of a gfx950 CU");

struct LmItem { uint32_t gi, cj; };

// counts per (group, window) cell, then the elements into their cells (order inside a cell: whatever the atomics give)
__global__ __launch_bounds__(256) void lists_count_kernel(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ row_off,
                                                          uint32_t n_rows, uint32_t n_windows, uint32_t* __restrict__ cells) {
    const uint32_t row = blockIdx.x;
    if (row >= n_rows) return;
    const uint32_t g = row / kLmGroup;
    for (uint32_t e = row_off[row] + threadIdx.x; e < row_off[row + 1]; e += 256u)
        atomicAdd(&cells[(uint64_t)g * n_windows + (pos[e] >> kLmWinBits)], 1u);
}
__global__ __launch_bounds__(256) void lists_place_kernel(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ row_off,
                                                          uint32_t n_rows, uint32_t n_windows, uint32_t* __restrict__ cursor,
                                                          uint32_t* __restrict__ elems) {
    const uint32_t row = blockIdx.x;
    if (row >= n_rows) return;
    const uint32_t g = row / kLmGroup;
    for (uint32_t e = row_off[row] + threadIdx.x; e < row_off[row + 1]; e += 256u) {
        const uint32_t p = pos[e];
        const uint32_t at = atomicAdd(&cursor[(uint64_t)g * n_windows + (p >> kLmWinBits)], 1u);
        elems[at] = ((row & (kLmChunk - 1u)) << kLmWinBits) | (p & (kLmWin - 1u));
    }
}

struct LmWin { uint32_t ab, ae, fb, fe; bool ok; };

__global__ __launch_bounds__(kLmThreads, 1) void lists_matrix_kernel(
    const uint32_t* __restrict__ elems, const uint32_t* __restrict__ off, uint32_t n_windows,
    const uint32_t* __restrict__ rowlen, const LmItem* __restrict__ items, uint32_t n_rows, int op,
    uint32_t* __restrict__ out, uint64_t ld) {
    __shared__ __attribute__((aligned(16))) uint32_t table[kLmTableBytes / 4u];   // [position][plane][2 words]
    __shared__ __attribute__((aligned(16))) uint32_t cnt[kLmCountBytes / 4u];     // [a][j / 2]: two 16-bit counters per word
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const LmItem it = items[blockIdx.x];
    for (uint32_t w = tid * 4u; w < kLmTableBytes / 4u; w += (uint32_t)kLmThreads * 4u)
        *reinterpret_cast<uint4*>(&table[w]) = uint4{0u, 0u, 0u, 0u};
    for (uint32_t w = tid * 4u; w < kLmCountBytes / 4u; w += (uint32_t)kLmThreads * 4u)
        *reinterpret_cast<uint4*>(&cnt[w]) = uint4{0u, 0u, 0u, 0u};

    const uint32_t* offA0 = off + (uint64_t)it.gi * n_windows;
    const uint32_t* offA1 = offA0 + n_windows;
    const uint32_t* offF0 = off + (uint64_t)(it.cj * kLmChunkGroups) * n_windows;
    const uint32_t* offF1 = offF0 + (uint64_t)kLmChunkGroups * n_windows;

    // the windows in which both sides list something, 64 at a time: every wave finds the same ones (uniform control flow)
    uint32_t wb = 0;
    uint64_t mask = 0;
    uint32_t l_ab = 0, l_ae = 0, l_fb = 0, l_fe = 0;
    auto load_windows = [&]() {
        const uint32_t w = wb + lane;
        const bool in = w < n_windows;
        l_ab = in ? offA0[w] : 0u;
        l_ae = in ? offA1[w] : 0u;
        l_fb = in ? offF0[w] : 0u;
        l_fe = in ? offF1[w] : 0u;
        mask = __ballot(l_ae > l_ab && l_fe > l_fb);
    };
    auto next_window = [&]() -> LmWin {
        for (;;) {
            if (mask) {
                const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(__ffsll((unsigned long long)mask) - 1));
                mask &= mask - 1ull;
                return LmWin{(uint32_t)__builtin_amdgcn_readlane((int)l_ab, (int)b), (uint32_t)__builtin_amdgcn_readlane((int)l_ae, (int)b),
                             (uint32_t)__builtin_amdgcn_readlane((int)l_fb, (int)b), (uint32_t)__builtin_amdgcn_readlane((int)l_fe, (int)b), true};
            }
            wb += 64u;
            if (wb >= n_windows) return LmWin{0u, 0u, 0u, 0u, false};
            load_windows();
        }
    };
    auto toggle = [&](uint32_t b, uint32_t e, uint32_t plane) {
        for (uint32_t i = b + tid; i < e; i += (uint32_t)kLmThreads) {
            const uint32_t v = elems[i];
            const uint32_t a = (v >> kLmWinBits) & (kLmGroup - 1u), p = v & (kLmWin - 1u);
            atomicXor(&table[p * 4u + plane * 2u + (a >> 5)], 1u << (a & 31u));
        }
    };
    auto lookup = [&](uint32_t b, uint32_t e, uint32_t plane) {
        for (uint32_t i = b + tid; i < e; i += (uint32_t)kLmThreads) {
            const uint32_t v = elems[i];
            const uint32_t j = (v >> kLmWinBits) & (kLmChunk - 1u), p = v & (kLmWin - 1u);
            const uint2 m = *reinterpret_cast<const uint2*>(&table[p * 4u + plane * 2u]);
            const uint32_t inc = 1u << (16u * (j & 1u));
            for (uint32_t x = m.x; x; x &= x - 1u)
                atomicAdd(&cnt[((uint32_t)__builtin_ctz(x) * kLmChunk + j) >> 1], inc);
            for (uint32_t x = m.y; x; x &= x - 1u)
                atomicAdd(&cnt[((32u + (uint32_t)__builtin_ctz(x)) * kLmChunk + j) >> 1], inc);
        }
    };

    load_windows();
    LmWin nxt = next_window();
    __syncthreads();   // the zeroed table
    if (nxt.ok) toggle(nxt.ab, nxt.ae, 0u);
    __syncthreads();
    LmWin prv{0u, 0u, 0u, 0u, false};
    for (uint32_t k = 0; nxt.ok; ++k) {
        const LmWin cur = nxt;
        nxt = next_window();
        lookup(cur.fb, cur.fe, k & 1u);
        if (prv.ok) toggle(prv.ab, prv.ae, (k + 1u) & 1u);   // out of the plane the next window takes ...
        if (nxt.ok) toggle(nxt.ab, nxt.ae, (k + 1u) & 1u);   // ... and the next window in (XOR both: any order)
        __syncthreads();
        prv = cur;
    }
    // the tile: rows gi * 64 + a, columns cj * 256 + j, strict upper part
    const uint32_t row0 = it.gi * kLmGroup, col0 = it.cj * kLmChunk;
    for (uint32_t idx = tid; idx < kLmGroup * kLmChunk; idx += (uint32_t)kLmThreads) {
        const uint32_t a = idx / kLmChunk, j = idx % kLmChunk;
        const uint32_t row = row0 + a, col = col0 + j;
        if (row < n_rows && col < n_rows && col > row) {
            uint32_t c = (cnt[idx >> 1] >> (16u * (j & 1u))) & 0xffffu;
            if (op == STORM_HIP_OP_OR) c = rowlen[row] + rowlen[col] - c;
            else if (op == STORM_HIP_OP_XOR) c = rowlen[row] + rowlen[col] - 2u * c;
            out[(uint64_t)row * ld + col] = c;
        }
    }
}

}  // namespace

struct storm_hip_rowlists_s {
    uint64_t n_rows = 0, n_elems = 0, n_bits = 0;
    uint32_t n_groups = 0;   // groups of 64 rows, a multiple of 4
    uint32_t n_windows = 0;
    uint32_t* d_elems = nullptr;
    uint32_t* d_off = nullptr;      // [n_groups + 1][n_windows]
    uint32_t* d_rowlen = nullptr;
    LmItem* d_items = nullptr;
    uint32_t n_items = 0;
};

extern "C" {

void storm_hip_rowlists_destroy(storm_hip_ctx_t* ctx, storm_hip_rowlists_t* l) {
    if (!l) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(l->d_elems);
    (void)hipFree(l->d_off);
    (void)hipFree(l->d_rowlen);
    (void)hipFree(l->d_items);
    delete l;
}

int storm_hip_rowlists_create_blocks(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                     const uint64_t* row_block_offset, const uint32_t* block_id,
                                     const uint8_t* block_kind, const uint32_t* block_n,
                                     const void* const* block_ptr, storm_hip_rowlists_t** out) {
    return guarded("storm_hip_rowlists_create_blocks", [&]() -> int {
        if (!ctx || !out) {
            set_error("rowlists_create: NULL context or output");
            return STORM_HIP_EINVAL;
        }
        *out = nullptr;
        if (n_rows == 0) return STORM_HIP_OK;
        if (!row_block_offset || (n_blocks && (!block_id || !block_kind || !block_n || !block_ptr))) {
            set_error("rowlists_create: NULL descriptor array");
            return STORM_HIP_EINVAL;
        }
        if (row_block_offset[0] != 0 || row_block_offset[n_rows] != n_blocks) {
            set_error("rowlists_create: row_block_offset must run from 0 to n_blocks");
            return STORM_HIP_EINVAL;
        }
        if (n_rows >= (1ull << 24)) return STORM_HIP_OK;
        // ---- eligibility: lists only, short rows, few elements
        uint64_t n_elems = 0, max_pos = 0;
        for (uint64_t r = 0; r < n_rows; ++r) {
            if (row_block_offset[r] > row_block_offset[r + 1] || row_block_offset[r + 1] > n_blocks) {
                set_error("rowlists_create: row_block_offset is not a CSR over the blocks");
                return STORM_HIP_EINVAL;
            }
            uint64_t len = 0;
            for (uint64_t b = row_block_offset[r]; b < row_block_offset[r + 1]; ++b) {
                if (block_kind[b] != 0) return STORM_HIP_OK;   // a bitmap block: the dense replica's case
                if (b > row_block_offset[r] && block_id[b] <= block_id[b - 1]) {
                    set_error("rowlists_create: block ids of row %llu are not ascending", (unsigned long long)r);
                    return STORM_HIP_EINVAL;
                }
                if (block_id[b] >= 65536u || block_n[b] > 65536u || (block_n[b] && (!block_ptr[b] || ((uintptr_t)block_ptr[b] & 1)))) {
                    set_error("rowlists_create: block %llu: id, length or list pointer out of range", (unsigned long long)b);
                    return STORM_HIP_EINVAL;
                }
                len += block_n[b];
                if (block_n[b]) max_pos = std::max<uint64_t>(max_pos, (uint64_t)block_id[b] * 65536u + 65535u);
            }
            if (len > 65535u) return STORM_HIP_OK;   // 16-bit counters
            n_elems += len;
        }
        if (n_elems == 0 || n_elems > (1ull << 26)) return STORM_HIP_OK;
        const uint32_t n_windows = (uint32_t)(max_pos / kLmWin + 1u);
        const uint32_t n_groups = (uint32_t)((n_rows + kLmChunk - 1u) / kLmChunk * kLmChunkGroups);
        if ((uint64_t)(n_groups + 1u) * n_windows > (1ull << 24)) return STORM_HIP_OK;

        struct Deleter {
            storm_hip_ctx_t* ctx;
            void operator()(storm_hip_rowlists_t* l) const { storm_hip_rowlists_destroy(ctx, l); }
        };
        std::unique_ptr<storm_hip_rowlists_t, Deleter> owner(new storm_hip_rowlists_t(), Deleter{ctx});
        storm_hip_rowlists_t* l = owner.get();
        l->n_rows = n_rows;
        l->n_elems = n_elems;
        l->n_bits = max_pos + 1u;
        l->n_groups = n_groups;
        l->n_windows = n_windows;

        // ---- rows as global positions (checked: strictly ascending — a repeated position would toggle itself away)
        std::vector<uint32_t> pos(n_elems), row_off(n_rows + 1), rowlen(n_rows);
        {
            uint64_t e = 0;
            for (uint64_t r = 0; r < n_rows; ++r) {
                row_off[r] = (uint32_t)e;
                for (uint64_t b = row_block_offset[r]; b < row_block_offset[r + 1]; ++b) {
                    const uint16_t* v = static_cast<const uint16_t*>(block_ptr[b]);
                    const uint32_t base = block_id[b] * 65536u;
                    for (uint32_t k = 0; k < block_n[b]; ++k) {
                        if (k && v[k] <= v[k - 1]) {
                            set_error("rowlists_create: list of block %llu is not strictly ascending", (unsigned long long)b);
                            return STORM_HIP_EINVAL;
                        }
                        pos[e++] = base + v[k];
                    }
                }
                rowlen[r] = (uint32_t)e - row_off[r];
            }
            row_off[n_rows] = (uint32_t)e;
        }
        STORM_HIP_TRY(hipSetDevice(ctx->device));
        struct Temps {
            uint32_t *pos = nullptr, *row_off = nullptr, *cursor = nullptr;
            ~Temps() { (void)hipFree(pos); (void)hipFree(row_off); (void)hipFree(cursor); }
        } t;
        const size_t n_cells = (size_t)(n_groups + 1u) * n_windows;
        STORM_HIP_TRY(hipMalloc(&t.pos, n_elems * sizeof(uint32_t)));
        STORM_HIP_TRY(hipMalloc(&t.row_off, (n_rows + 1) * sizeof(uint32_t)));
        STORM_HIP_TRY(hipMalloc(&t.cursor, n_cells * sizeof(uint32_t)));
        STORM_HIP_TRY(hipMalloc(&l->d_elems, n_elems * sizeof(uint32_t)));
        STORM_HIP_TRY(hipMalloc(&l->d_off, n_cells * sizeof(uint32_t)));
        STORM_HIP_TRY(hipMalloc(&l->d_rowlen, n_rows * sizeof(uint32_t)));
        STORM_HIP_TRY(hipMemcpyAsync(t.pos, pos.data(), n_elems * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        STORM_HIP_TRY(hipMemcpyAsync(t.row_off, row_off.data(), (n_rows + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        STORM_HIP_TRY(hipMemcpyAsync(l->d_rowlen, rowlen.data(), n_rows * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        STORM_HIP_TRY(hipMemsetAsync(t.cursor, 0, n_cells * sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(lists_count_kernel, dim3((uint32_t)n_rows), dim3(256), 0, ctx->stream, t.pos, t.row_off,
                           (uint32_t)n_rows, n_windows, t.cursor);
        STORM_HIP_TRY(hipGetLastError());
        std::vector<uint32_t> cells(n_cells), offs(n_cells);
        STORM_HIP_TRY(hipMemcpyAsync(cells.data(), t.cursor, n_cells * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        {   // window-major order of the cells: off[g][w] runs over w first ... no: over g inside w
            uint64_t run = 0;
            for (uint32_t w = 0; w < n_windows; ++w)
                for (uint32_t g = 0; g < n_groups; ++g) {
                    offs[(size_t)g * n_windows + w] = (uint32_t)run;
                    run += cells[(size_t)g * n_windows + w];
                }
            if (run != n_elems) {
                set_error("rowlists_create: the device counted %llu of %llu elements", (unsigned long long)run, (unsigned long long)n_elems);
                return STORM_HIP_EHIP;
            }
            // row n_groups: where a window ends = where the next one begins
            for (uint32_t w = 0; w < n_windows; ++w)
                offs[(size_t)n_groups * n_windows + w] = w + 1u < n_windows ? offs[w + 1u] : (uint32_t)n_elems;
        }
        STORM_HIP_TRY(hipMemcpyAsync(l->d_off, offs.data(), n_cells * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        STORM_HIP_TRY(hipMemcpyAsync(t.cursor, l->d_off, n_cells * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(lists_place_kernel, dim3((uint32_t)n_rows), dim3(256), 0, ctx->stream, t.pos, t.row_off,
                           (uint32_t)n_rows, n_windows, t.cursor, l->d_elems);
        STORM_HIP_TRY(hipGetLastError());
        // ---- the tiles: chunk after chunk (the far stream of a chunk is shared by its tiles)
        std::vector<LmItem> items;
        const uint32_t groups_used = (uint32_t)((n_rows + kLmGroup - 1u) / kLmGroup);
        for (uint32_t cj = 0; cj < n_groups / kLmChunkGroups; ++cj)
            for (uint32_t gi = 0; gi < groups_used && gi <= cj * kLmChunkGroups + (kLmChunkGroups - 1u); ++gi)
                items.push_back({gi, cj});
        l->n_items = (uint32_t)items.size();
        STORM_HIP_TRY(hipMalloc(&l->d_items, items.size() * sizeof(LmItem)));
        STORM_HIP_TRY(hipMemcpyAsync(l->d_items, items.data(), items.size() * sizeof(LmItem), hipMemcpyHostToDevice, ctx->stream));
        STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        *out = owner.release();
        return STORM_HIP_OK;
    });
}

// 1: the lists are expected to beat the dense replica's multiply (or the option says so); 0: keep the dense path
int storm_hip_rowlists_worthwhile(storm_hip_ctx_t* ctx, const storm_hip_rowlists_t* l) {
    if (!ctx || ctx->matrix_lists == 0) return 0;
    if (!l) return 1;   // "is the path switched on at all": worth building the lists to find out
    if (ctx->matrix_lists > 0) return 1;
    // density of the dense replica the lists stand for (its rows are padded to 512 bits); crossover measured at the
    // README's STORM_t shape: profiles/r05_j_storm_matrix_lists.jsonl
    const double bits = (double)((l->n_bits + 511u) / 512u * 512u) * (double)l->n_rows;
    return (double)l->n_elems <= bits * (double)ctx->matrix_lists_permille_x10 / 10000.0;
}

int storm_hip_rowlists_pairw_matrix_device(storm_hip_ctx_t* ctx, const storm_hip_rowlists_t* l, int op,
                                           uint32_t* d_out, uint64_t ld) {
    if (!ctx || !l || !d_out) {
        set_error("rowlists_pairw_matrix: NULL argument");
        return STORM_HIP_EINVAL;
    }
    if (op != STORM_HIP_OP_AND && op != STORM_HIP_OP_OR && op != STORM_HIP_OP_XOR) {
        set_error("rowlists_pairw_matrix: op %d", op);
        return STORM_HIP_EINVAL;
    }
    if (ld < l->n_rows) {
        set_error("rowlists_pairw_matrix: ld %llu < %llu rows", (unsigned long long)ld, (unsigned long long)l->n_rows);
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    kernel_time_mark(ctx);
    hipLaunchKernelGGL(lists_matrix_kernel, dim3(l->n_items), dim3(kLmThreads), 0, ctx->stream, l->d_elems, l->d_off,
                       l->n_windows, l->d_rowlen, l->d_items, (uint32_t)l->n_rows, op, d_out, ld);
    kernel_time_mark(ctx);
    STORM_HIP_TRY(hipGetLastError());
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->pass_report[0] = STORM_HIP_RAN_LISTS_MATRIX;
    ctx->pass_report[1] = 0;
    ctx->pass_report[2] = (uint64_t)(l->n_groups) * l->n_elems / 2u;
    ctx->pass_report[3] = kLmGroup;
    return STORM_HIP_OK;
}

uint64_t storm_hip_rowlists_n_elems(const storm_hip_rowlists_t* l) { return l ? l->n_elems : 0; }

}  // extern "C"
