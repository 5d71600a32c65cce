// storm_hip_lists.hip — K5: the per-pair matrix of a LIST-ONLY sparse container, from the lists themselves.
//
// What it replaces: STORM_bitmap_cont_intersect_cardinality (storm.c:790-814) called for every pair of rows — the block-id
// merge of storm.c:75-106 and, for two list blocks, STORM_intersect_vector16_cardinality (storm.c:4-73, kind dispatch
// :618-656). Until round 5 the device path of that matrix was a dense replica of the rows (N x M bits: 655 MB at the
// README's STORM_t shape) multiplied by the tile kernels whatever the density: 6.4 ms at 524 positions per row of 524288,
// where the rows hold 21 MB of positions between them.
//
// Formulation (a hash join per output tile, the table in the LDS):
//   data   : every listed position as ONE 32-bit element, (row % 768) << 13 | position & 8191, ordered by WINDOW of 8192
//            positions first and by GROUP of 64 rows inside a window; off[g][w] = first element of (window w, group g).
//            The elements of a group — or of twelve consecutive groups, a CHUNK of 768 rows — in a window are contiguous.
//   item   : one output tile = group gi (64 "A" rows) x chunk cj (768 "far" rows), all windows. One workgroup of 1024
//            threads, one item.
//   LDS    : table[8192 positions] of 64-bit row masks (64 KiB): bit a of table[p] = "A row a lists position p of the
//            current window"; counters[64][768] of 16 bits (96 KiB) = the tile.
//   step   : for every window in which both sides list something — look every far element up (one ds_read_b64; for
//            every set bit one 16-bit LDS increment: that is one intersecting position of one row pair); then, behind a
//            barrier, the A elements of this window are toggled out of the table and those of the next such window in.
//            Set and clear are both XOR: they commute, so the two may run in any order even when they hit the same bit.
//            (First form: two planes of masks, 256-row chunks, the re-toggling beside the lookups and ONE barrier per step
//            — three times the steps, and the steps are what costs: 1.25 ms at 524 positions per row against this form's
//            figure in DESIGN.md.)
//   output : the tile's strict upper part is written once, row segments of 1 KiB; OR / XOR counts from the row lengths.
//            Nothing is zero-filled and nothing is added to in global memory; entries i >= j are not touched (as the tile
//            kernels leave them).
//   work   : lookups = N / 64 x elements / 2; increments = the matrix's sum. Both fall with the density, where the dense
//            multiply does not: see worthwhile() for the crossover.
// Eligible: every block a list, every row at most 65535 positions (16-bit counters), at most 2^26 elements, at most
// 2^24 (group, window) cells. Anything else keeps the dense replica.
#include "storm_hip_internal.h"

#include <chrono>
#include <cstring>
#include <memory>

using namespace storm;

// (an empty launch at context creation loads this file's code object ahead of the first real call: storm_hip_ctx_create)
namespace storm {
__global__ void warm_lists_kernel() {}
void warm_lists_code(hipStream_t stream) { hipLaunchKernelGGL(warm_lists_kernel, dim3(1), dim3(64), 0, stream); }
}  // namespace storm


namespace {

constexpr int kLmThreads = 1024;
constexpr uint32_t kLmGroup = 64;        // A rows per tile
constexpr uint32_t kLmChunkGroups = 12;  // far rows per tile: 12 groups = 768
constexpr uint32_t kLmChunk = kLmGroup * kLmChunkGroups;
// [r6] A window is 8191 positions, not 8192: entry 0 of the table is nobody's, and an ELEMENT is
//     (row % 768) << 16 | 8 * (position % 8191 + 1)     — its row in the chunk and the BYTE OFFSET of its table entry —
// so that 0 means "no element" (what a buffer load beyond a range returns: no bounds test, no select) and a lookup is one
// v_and + ds_read_b64. The window kernel issues ~25 vector instructions per 64 lookups and the vector ALU is what it waits
// for (profiles/r06_d_lists_matrix_counters.txt): the old encoding (row << 13 | position) cost five more per lookup.
constexpr uint32_t kLmWin = 8191u;
constexpr uint32_t kLmTableBytes = (kLmWin + 1u) * 8u;           // a 64-bit row mask per position, entry 0 unused
constexpr uint32_t kLmRowShift = 16u;
constexpr uint32_t kLmCountBytes = kLmGroup * kLmChunk * 2u;     // 16-bit counters
static_assert(kLmTableBytes + kLmCountBytes <= 160u * 1024u, "LDS of a gfx950 CU");

struct LmItem { uint32_t gi, cj; };

// counts per (group, window) cell, then the elements into their cells (order inside a cell: whatever the atomics give).
// [r6] A row's positions ascend, so what a row holds of one window is one RUN of its list: the thread that finds a run's
// first element finds its end by bisection and books the whole run with ONE atomic (10000 rows x 64 windows instead of 36
// million elements at 3670 positions per row: 4.7 + 7.7 ms of a first call for the two kernels before).
__device__ __forceinline__ uint32_t lists_run_end(const uint32_t* __restrict__ pos, uint32_t e, uint32_t r1, uint32_t w) {
    uint32_t lo = e + 1u, hi = r1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pos[mid] / kLmWin == w) lo = mid + 1u;
        else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(256) void lists_count_kernel(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ row_off,
                                                          uint32_t n_rows, uint32_t n_windows, uint32_t* __restrict__ cells) {
    const uint32_t row = blockIdx.x;
    if (row >= n_rows) return;
    const uint32_t g = row / kLmGroup, r0 = row_off[row], r1 = row_off[row + 1];
    for (uint32_t e = r0 + threadIdx.x; e < r1; e += 256u) {
        const uint32_t w = pos[e] / kLmWin;
        if (e != r0 && pos[e - 1u] / kLmWin == w) continue;
        atomicAdd(&cells[(uint64_t)g * n_windows + w], lists_run_end(pos, e, r1, w) - e);
    }
}
__global__ __launch_bounds__(256) void lists_place_kernel(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ row_off,
                                                          uint32_t n_rows, uint32_t n_windows, uint32_t* __restrict__ cursor,
                                                          uint32_t* __restrict__ elems) {
    const uint32_t row = blockIdx.x;
    if (row >= n_rows) return;
    const uint32_t g = row / kLmGroup, r0 = row_off[row], r1 = row_off[row + 1];
    const uint32_t tag = (row % kLmChunk) << kLmRowShift;
    for (uint32_t e = r0 + threadIdx.x; e < r1; e += 256u) {
        const uint32_t w = pos[e] / kLmWin;
        if (e != r0 && pos[e - 1u] / kLmWin == w) continue;
        const uint32_t end = lists_run_end(pos, e, r1, w);
        const uint32_t at = atomicAdd(&cursor[(uint64_t)g * n_windows + w], end - e);
        for (uint32_t k = e; k < end; ++k) elems[at + (k - e)] = tag | ((pos[k] % kLmWin + 1u) << 3);
    }
}

// [r6] The elements of a (group, window) cell, re-ordered so that the window kernel's lookups meet fewer bank conflicts: 32
// consecutive lanes of a ds_read_b64 are served together, one 8-byte entry per pair of banks, and positions that are random
// collide (0.70 of the kernel's LDS cycles were conflict cycles, profiles/r06_d_lists_matrix_counters.txt). Dealt here: the
// elements are bucketed by the bank pair of their entry and emitted round by round, one of every bucket that still has one —
// any 32 consecutive elements of the first rounds read 32 different bank pairs. One workgroup per cell, in place; a cell
// beyond kDealMax elements stays as the atomics of lists_place_kernel left it (the order inside a cell carries no meaning).
constexpr uint32_t kDealMax = 12288u;
__global__ __launch_bounds__(256) void lists_deal_kernel(uint32_t* __restrict__ elems, const uint32_t* __restrict__ off,
                                                         uint32_t n_windows) {
    __shared__ uint32_t buf[kDealMax];
    __shared__ uint32_t count[32], cursor[32];
    const uint32_t g = blockIdx.x / n_windows, w = blockIdx.x % n_windows;
    const uint32_t b = off[(uint64_t)g * n_windows + w], e = off[(uint64_t)(g + 1u) * n_windows + w];
    const uint32_t n = e - b;
    if (n < 64u || n > kDealMax) return;
    if (threadIdx.x < 32u) count[threadIdx.x] = cursor[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256u) {
        const uint32_t v = elems[b + i];
        buf[i] = v;
        atomicAdd(&count[(v >> 3) & 31u], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256u) {
        const uint32_t v = buf[i], bk = (v >> 3) & 31u;
        const uint32_t r = atomicAdd(&cursor[bk], 1u);   // the element's round
        uint32_t at = 0;                                 // elements of earlier rounds + of this round in buckets before bk
#pragma unroll
        for (uint32_t k = 0; k < 32u; ++k) at += min(count[k], r) + (k < bk && count[k] > r ? 1u : 0u);
        elems[b + at] = v;
    }
}

struct LmWin { uint32_t ab, ae, fb, fe; bool ok; };

__global__ __launch_bounds__(kLmThreads, 1) void lists_matrix_kernel(
    const uint32_t* __restrict__ elems, const uint32_t* __restrict__ off, uint32_t n_windows,
    const uint32_t* __restrict__ rowlen, const LmItem* __restrict__ items, uint32_t n_rows, int op,
    uint32_t* __restrict__ out, uint64_t ld, uint32_t dbg) {
    // (one array, the table first: its entries are addressed by the elements' own 16-bit byte offsets)
    __shared__ __attribute__((aligned(16))) uint32_t lds[(kLmTableBytes + kLmCountBytes) / 4u];
    uint32_t* const table = lds;                          // [1 + position][2 words]
    uint32_t* const cnt = lds + kLmTableBytes / 4u;       // [a][j / 2]: two 16-bit counters per word
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const LmItem it = items[blockIdx.x];
    if (it.gi == 0xffffffffu) return;   // (a filler: see the item order)
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
    // Elements travel in registers: the far elements of a step (the first kFarRegs per thread) are loaded one step before
    // they are looked up, the A elements (kARegs per thread) two steps before they are toggled in, and stay until they have
    // been toggled out; kInvalid beyond a range. History (profiles/r05_j_lists_matrix_ablation.jsonl): loads where they
    // are used, 1024 threads: 1.34 ms at 524 positions per row (a round trip to the L2 per step, every wave waiting);
    // loads a step early: 1.10 — the step loop WITHOUT any element, barrier or store still takes 0.71 ms (195 000 steps of
    // ~130 vector instructions on sixteen waves: the CU's instruction issue), and four waves that carry four times as much
    // each are slower still (3.1 ms: one wave per SIMD covers no latency; eight waves: 1.64). The steps are the cost of this formulation:
    // lists_hash_kernel below has none and takes over where the rows are short.
    constexpr uint32_t kInvalid = 0u;
    // [r6] kFarRegs 8 -> 24, in three groups of 8 that a step only touches while its window holds that many far elements
    // (a test on scalars: every wave takes the same way): at 2096 positions per row a step looks up 24 elements per thread,
    // and two thirds of them used to be loaded inside the step, the thread waiting for every batch of 8
    constexpr uint32_t kFarGroup = 8u;
    constexpr uint32_t kFarRegs = 3u * kFarGroup;
    constexpr uint32_t kARegs = 1u;
    constexpr uint32_t kBatch = 8u;      // loads in flight per thread in the remainder loops
    constexpr uint32_t kT = (uint32_t)kLmThreads;
    auto toggle1 = [&](uint32_t v) {
        if (v != kInvalid) {
            const uint32_t a = (v >> kLmRowShift) & (kLmGroup - 1u);
            atomicXor(&table[((v & 0xffffu) >> 2) + (a >> 5)], 1u << (a & 31u));
        }
    };
    auto count_bits = [&](uint32_t v, uint2 m) {
        const uint32_t j = v >> kLmRowShift;   // row in chunk: < 768
        const uint32_t inc = 1u << (16u * (j & 1u));
        uint32_t* const cj = cnt + (j >> 1);   // counters of far row j: row a's word is kLmChunk / 2 words further per a
        // one loop over the 64 rows' bits (two loops, one per word, ran max(bits of x) + max(bits of y) trips per wave and
        // paid for two loop heads per element)
        for (uint64_t x = (uint64_t)m.x | (uint64_t)m.y << 32; x; x &= x - 1ull)
            atomicAdd(cj + (uint32_t)__builtin_ctzll(x) * (kLmChunk / 2u), inc);
    };
    // the masks of a batch are read together (one LDS round trip per batch, not per element); "no element" reads entry 0,
    // which no position owns and which stays zero
    auto lookup_batch = [&](const uint32_t* v) {
        uint2 m[kFarGroup];
#pragma unroll
        for (uint32_t q = 0; q < kFarGroup; ++q)
            m[q] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(table) + (v[q] & 0xffffu));
#pragma unroll
        for (uint32_t q = 0; q < kFarGroup; ++q)
            if (m[q].x | m[q].y) count_bits(v[q], m[q]);
    };
    // elems[b, e) as a buffer: what is read beyond its end is 0 = no element (matrix_lists_debug & 1: an empty buffer)
    auto range = [&](uint32_t b, uint32_t e) -> __amdgpu_buffer_rsrc_t {
        const uint32_t bytes = (dbg & 1u) || e <= b ? 0u : (e - b) * 4u;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(elems) + __builtin_amdgcn_readfirstlane((int)b), 0,
                                                 __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    auto load_el = [&](__amdgpu_buffer_rsrc_t rs, uint32_t idx) -> uint32_t {
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(idx * 4u), 0, 0);
    };
    // the far elements a step holds in registers: group g only while the window lists more than g * 8 * threads of them
    auto lookup_regs = [&](const uint32_t (&v)[kFarRegs], uint32_t count) {
#pragma unroll
        for (uint32_t g = 0; g < kFarRegs / kFarGroup; ++g)
            if (count > g * kFarGroup * kT) lookup_batch(&v[g * kFarGroup]);
    };
    auto load_regs = [&](uint32_t (&v)[kFarRegs], uint32_t b, uint32_t e, bool ok) {
        const __amdgpu_buffer_rsrc_t rs = range(b, ok ? e : b);
#pragma unroll
        for (uint32_t g = 0; g < kFarRegs / kFarGroup; ++g)
            if (ok && e - b > g * kFarGroup * kT) {
#pragma unroll
                for (uint32_t q = g * kFarGroup; q < (g + 1u) * kFarGroup; ++q) v[q] = load_el(rs, tid + q * kT);
            }
    };
    // the elements of [b, e) from the `skip`-th per thread on, kBatch loads in flight
    auto toggle_rest = [&](uint32_t b, uint32_t e, uint32_t skip) {
        const __amdgpu_buffer_rsrc_t rs = range(b, e);
        for (uint32_t i = skip * kT; i < e - b; i += kBatch * kT) {   // (scalars: the same trips for every wave)
            uint32_t v[kBatch];
#pragma unroll
            for (uint32_t q = 0; q < kBatch; ++q) v[q] = load_el(rs, i + tid + q * kT);
#pragma unroll
            for (uint32_t q = 0; q < kBatch; ++q) toggle1(v[q]);
        }
    };
    // (beyond the registers: batches of 8 per thread, the next batch's loads in flight while this one is looked up)
    auto lookup_rest = [&](uint32_t b, uint32_t e, uint32_t skip) {
        const uint32_t first = skip * kT, n = e - b;
        if (first >= n) return;
        const __amdgpu_buffer_rsrc_t rs = range(b, e);
        uint32_t va[kFarGroup], vb[kFarGroup];
#pragma unroll
        for (uint32_t q = 0; q < kFarGroup; ++q) va[q] = load_el(rs, first + tid + q * kT);
        for (uint32_t base = first;;) {   // (`base` is a scalar: the loop's exits are the same for every wave)
            const uint32_t nb = base + kFarGroup * kT;
            if (nb < n) {
#pragma unroll
                for (uint32_t q = 0; q < kFarGroup; ++q) vb[q] = load_el(rs, nb + tid + q * kT);
            }
            lookup_batch(va);
            if (nb >= n) break;
            const uint32_t nc = nb + kFarGroup * kT;
            if (nc < n) {
#pragma unroll
                for (uint32_t q = 0; q < kFarGroup; ++q) va[q] = load_el(rs, nc + tid + q * kT);
            }
            lookup_batch(vb);
            if (nc >= n) break;
            base = nc;
        }
    };
    // (a barrier for the LDS alone: __syncthreads() also waits for every global load in flight — the loads that were
    //  issued a step early precisely so that nobody waits for them)
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(dbg & 4u)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // Step k multiplies window w[k], which the table holds: its far elements (the first kFarRegs per thread) and the A
    // elements of w[k + 1] were loaded during step k - 1, the A elements of w[k] are still in their registers from when
    // they were toggled in. Two phases: the lookups; then, behind a barrier, w[k] is toggled out and w[k + 1] in (both XOR:
    // any order, same table) while the far elements of w[k + 1] and the A elements of w[k + 2] are already on their way.
    load_windows();
    LmWin w0 = next_window(), w1 = next_window(), w2 = next_window();
    uint32_t a_cur[kARegs], a_next[kARegs], a_next2[kARegs];
    uint32_t f_cur[kFarRegs], f_next[kFarRegs];
#pragma unroll
    for (uint32_t q = 0; q < kARegs; ++q) {
        a_next2[q] = kInvalid;
        a_cur[q] = load_el(range(w0.ab, w0.ok ? w0.ae : w0.ab), tid + q * kT);
        a_next[q] = load_el(range(w1.ab, w1.ok ? w1.ae : w1.ab), tid + q * kT);
    }
#pragma unroll
    for (uint32_t q = 0; q < kFarRegs; ++q) f_cur[q] = f_next[q] = kInvalid;
    load_regs(f_cur, w0.fb, w0.fe, w0.ok);
    lds_barrier();   // the zeroed table
#pragma unroll
    for (uint32_t q = 0; q < kARegs; ++q) toggle1(a_cur[q]);
    if (w0.ok && w0.ae - w0.ab > kARegs * kT) toggle_rest(w0.ab, w0.ae, kARegs);
    lds_barrier();
    // One step. The registers a step LOADS and the ones it CONSUMES swap roles from step to step: the loop below is unrolled
    // twice over the two namings, so that no register is ever copied while its load is in flight (a copy is a use: a version
    // that rotated the names with v_mov at the end of a step waited there for every load it had just issued).
    auto step = [&](uint32_t (&fc)[kFarRegs], uint32_t (&fn)[kFarRegs], uint32_t (&an)[kARegs], uint32_t (&an2)[kARegs]) {
#pragma unroll
        for (uint32_t q = 0; q < kARegs; ++q) an2[q] = load_el(range(w2.ab, w2.ok ? w2.ae : w2.ab), tid + q * kT);
        load_regs(fn, w1.fb, w1.fe, w1.ok);
        lookup_regs(fc, w0.fe - w0.fb);
        if (w0.fe - w0.fb > kFarRegs * kT) lookup_rest(w0.fb, w0.fe, kFarRegs);
        lds_barrier();
#pragma unroll
        for (uint32_t q = 0; q < kARegs; ++q) toggle1(a_cur[q]);
        if (w0.ae - w0.ab > kARegs * kT) toggle_rest(w0.ab, w0.ae, kARegs);
#pragma unroll
        for (uint32_t q = 0; q < kARegs; ++q) toggle1(an[q]);
        if (w1.ok && w1.ae - w1.ab > kARegs * kT) toggle_rest(w1.ab, w1.ae, kARegs);
        lds_barrier();
#pragma unroll
        for (uint32_t q = 0; q < kARegs; ++q) a_cur[q] = an[q];   // (long since loaded)
        w0 = w1; w1 = w2;
        w2 = next_window();
    };
    while (w0.ok && !(dbg & 16u)) {
        step(f_cur, f_next, a_next, a_next2);
        if (!w0.ok) break;
        step(f_next, f_cur, a_next2, a_next);
    }
    // the tile: rows gi * 64 + a, columns cj * 768 + j, strict upper part
    const uint32_t row0 = it.gi * kLmGroup, col0 = it.cj * kLmChunk;
    for (uint32_t idx = tid; idx < kLmGroup * kLmChunk; idx += (uint32_t)kLmThreads) {
        const uint32_t a = idx / kLmChunk, j = idx % kLmChunk;
        const uint32_t row = row0 + a, col = col0 + j;
        if (row < n_rows && col < n_rows && col > row && !(dbg & 8u)) {
            uint32_t c = (cnt[idx >> 1] >> (16u * (j & 1u))) & 0xffffu;
            if (op == STORM_HIP_OP_OR) c = rowlen[row] + rowlen[col] - c;
            else if (op == STORM_HIP_OP_XOR) c = rowlen[row] + rowlen[col] - 2u * c;
            out[(uint64_t)row * ld + col] = c;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// K5h — the same join with a HASH table instead of a direct-indexed one: no windows, no steps.
//   data   : the rows as they are — `pos`: every row's positions, row after row (CSR, `row_off`); `rtag`: row & 2047 per
//            element, so that a flat stream of elements knows its rows.
//   item   : group gi of G rows (G = 64 / 32 / 16 / 8: the largest whose every group lists at most kLhFill positions) x
//            chunk cj of F = 16384 / G rows. One workgroup of 1024 threads.
//   LDS    : open-addressing table of 8192 buckets of four 32-bit entries, position << 6 | row in group (128 KiB;
//            multiplicative hash, overflow into the next bucket; at most a quarter full); counters[G][F] of 16 bits (32 KiB).
//   work   : build the table from the group's elements (LDS compare-and-swap), then stream the chunk's elements —
//            contiguous in memory, eight loads in flight per thread — and read every element's home bucket (one
//            ds_read_b128): an entry with the same position is one intersecting position of one row pair.
//            Lookups = N / G x elements / 2; G falls with the row length, so the work grows with
//            the SQUARE of the density: this is the kernel of the sparse end, the window kernel above that of the middle.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kLhThreads = 1024;
constexpr uint32_t kLhSlots = 32768u;
constexpr uint32_t kLhEmpty = 0xffffffffu;
constexpr uint32_t kLhCounters = 16384u;     // G x F
constexpr uint32_t kLhFill = 8192u;          // elements of a group at most (a quarter of the slots; beyond ~130 positions per row the window kernel is level)
constexpr uint32_t kLhRowBits = 6u;          // row in group: G <= 64
constexpr uint32_t kLhMaxPos = (1u << (32u - kLhRowBits)) - 2u;

constexpr uint32_t kLhBuckets = kLhSlots / 4u;   // of four entries
__device__ __forceinline__ uint32_t lh_hash(uint32_t pos) { return (pos * 0x9E3779B1u) >> 19; }   // home bucket: 13 bits

__global__ __launch_bounds__(kLhThreads, 1) void lists_hash_kernel(
    const uint32_t* __restrict__ pos, const uint16_t* __restrict__ rtag, const uint32_t* __restrict__ row_off,
    const uint32_t* __restrict__ rowlen, const LmItem* __restrict__ items, uint32_t n_rows, uint32_t g_log2, int op,
    uint32_t* __restrict__ out, uint64_t ld, uint32_t dbg) {
    __shared__ __attribute__((aligned(16))) uint32_t table[kLhSlots];
    __shared__ __attribute__((aligned(16))) uint32_t cnt[kLhCounters / 2u];
    const uint32_t tid = threadIdx.x;
    const LmItem it = items[blockIdx.x];
    if (it.gi == 0xffffffffu) return;
    const uint32_t G = 1u << g_log2, f_log2 = 14u - g_log2, F = 1u << f_log2;
    for (uint32_t w = tid * 4u; w < kLhSlots; w += (uint32_t)kLhThreads * 4u)
        *reinterpret_cast<uint4*>(&table[w]) = uint4{kLhEmpty, kLhEmpty, kLhEmpty, kLhEmpty};
    for (uint32_t w = tid * 4u; w < kLhCounters / 2u; w += (uint32_t)kLhThreads * 4u)
        *reinterpret_cast<uint4*>(&cnt[w]) = uint4{0u, 0u, 0u, 0u};
    const uint32_t row0 = it.gi << g_log2, row1 = min(n_rows, row0 + G);
    const uint32_t col0 = it.cj << f_log2, col1 = min(n_rows, col0 + F);
    const uint32_t a_b = row_off[row0], a_e = row_off[row1], f_b = row_off[col0], f_e = row_off[col1];
    __syncthreads();
    // Buckets of four entries (16 bytes): an element goes into the first free slot of its home bucket, or of the next
    // one that has any; entries are never removed, so the used slots of a bucket are a prefix and the first empty slot
    // ends a search. At most a quarter full, a lookup reads ONE bucket in 99 % of the cases. (The first version probed slot
    // by slot: a wave walks until the longest of its 64 chains ends — 0.83 ms at 190 positions per row where this
    // takes 0.3, profiles/r05_j_*.)
    for (uint32_t e = a_b + tid; e < a_e; e += (uint32_t)kLhThreads) {
        const uint32_t p = pos[e];
        const uint32_t entry = (p << kLhRowBits) | ((uint32_t)rtag[e] & (G - 1u));
        uint32_t b = lh_hash(p);
        for (bool placed = false; !placed; b = (b + 1u) & (kLhBuckets - 1u))
#pragma unroll
            for (uint32_t s4 = 0; s4 < 4u && !placed; ++s4)
                placed = atomicCAS(&table[b * 4u + s4], kLhEmpty, entry) == kLhEmpty;
    }
    __syncthreads();
    constexpr uint32_t kBatch = 8u;
    for (uint32_t e = f_b + tid; e < f_e; e += kBatch * (uint32_t)kLhThreads) {
        uint32_t p[kBatch], j[kBatch];
#pragma unroll
        for (uint32_t q = 0; q < kBatch; ++q) {
            const uint32_t i = e + q * (uint32_t)kLhThreads;
            p[q] = (i < f_e && !(dbg & 1u)) ? pos[i] : kLhEmpty;
            j[q] = (i < f_e && !(dbg & 1u)) ? (uint32_t)rtag[i] & (F - 1u) : 0u;
        }
        if (dbg & 2u) {   // (timing: the stream alone)
            uint32_t acc = 0;
#pragma unroll
            for (uint32_t q = 0; q < kBatch; ++q) acc += p[q] + j[q];
            if (acc == 0x12345u) cnt[0] = acc;
            continue;
        }
        uint4 v[kBatch];
#pragma unroll
        for (uint32_t q = 0; q < kBatch; ++q)   // (an element beyond the stream reads bucket 0 and matches nothing: no key is kLhEmpty >> 6)
            v[q] = *reinterpret_cast<const uint4*>(&table[(p[q] == kLhEmpty ? 0u : lh_hash(p[q])) * 4u]);
#pragma unroll
        for (uint32_t q = 0; q < kBatch; ++q) {
            const uint32_t inc = 1u << (16u * (j[q] & 1u));
            const uint32_t key = p[q];
            auto hit = [&](uint32_t entry) {
                if ((entry >> kLhRowBits) == key && entry != kLhEmpty)
                    atomicAdd(&cnt[(((entry & (G - 1u)) << f_log2) + j[q]) >> 1], inc);
            };
            hit(v[q].x); hit(v[q].y); hit(v[q].z); hit(v[q].w);
            if (v[q].w != kLhEmpty && key != kLhEmpty) {   // a full bucket: the search goes on (rare)
                for (uint32_t b = (lh_hash(key) + 1u) & (kLhBuckets - 1u);; b = (b + 1u) & (kLhBuckets - 1u)) {
                    const uint4 u = *reinterpret_cast<const uint4*>(&table[b * 4u]);
                    hit(u.x); hit(u.y); hit(u.z); hit(u.w);
                    if (u.w == kLhEmpty) break;
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t idx = tid; idx < kLhCounters; idx += (uint32_t)kLhThreads) {
        const uint32_t a = idx >> f_log2, j = idx & (F - 1u);
        const uint32_t row = row0 + a, col = col0 + j;
        if (row < n_rows && col < n_rows && col > row) {
            uint32_t c = (cnt[idx >> 1] >> (16u * (j & 1u))) & 0xffffu;
            if (op == STORM_HIP_OP_OR) c = rowlen[row] + rowlen[col] - c;
            else if (op == STORM_HIP_OP_XOR) c = rowlen[row] + rowlen[col] - 2u * c;
            out[(uint64_t)row * ld + col] = c;
        }
    }
}

// the tiles of a (group, chunk) grid in the order the kernels want them: all tiles of a chunk on ONE XCD (workgroup b runs
// on XCD b % 8), one after the other, so that the chunk's far stream stays in that L2; chunks dealt longest first, lists
// that come out shorter filled with empty items
static std::vector<LmItem> tile_order(uint64_t n_rows, uint32_t group_rows, uint32_t chunk_rows) {
    constexpr uint32_t kXcds = 8;
    const uint32_t n_group = (uint32_t)((n_rows + group_rows - 1u) / group_rows);
    const uint32_t n_chunk = (uint32_t)((n_rows + chunk_rows - 1u) / chunk_rows);
    const uint32_t per = chunk_rows / group_rows;
    std::vector<std::vector<LmItem>> per_xcd(kXcds);
    for (uint32_t c = n_chunk; c-- > 0;) {   // (the last chunks have the most tiles)
        size_t best = 0;
        for (size_t x = 1; x < kXcds; ++x)
            if (per_xcd[x].size() < per_xcd[best].size()) best = x;
        for (uint32_t gi = 0; gi < n_group && gi <= c * per + (per - 1u); ++gi) per_xcd[best].push_back({gi, c});
    }
    size_t longest = 0;
    for (const auto& v : per_xcd) longest = std::max(longest, v.size());
    std::vector<LmItem> items;
    for (size_t i = 0; i < longest; ++i)
        for (uint32_t x = 0; x < kXcds; ++x)
            items.push_back(i < per_xcd[x].size() ? per_xcd[x][i] : LmItem{0xffffffffu, 0u});
    while (!items.empty() && items.back().gi == 0xffffffffu) items.pop_back();
    return items;
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
    // K5h: the rows as they are
    uint32_t* d_pos = nullptr;
    uint16_t* d_rtag = nullptr;
    uint32_t* d_row_off = nullptr;
    LmItem* d_hash_items = nullptr;
    uint32_t n_hash_items = 0;
    uint32_t hash_g_log2 = 0;   // 0: no group size fits the table (rows too long): the window kernel only
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
    (void)hipFree(l->d_pos);
    (void)hipFree(l->d_rtag);
    (void)hipFree(l->d_row_off);
    (void)hipFree(l->d_hash_items);
    delete l;
}

namespace {
struct LxBlock {
    uint32_t at;     // where the block's positions go (= where its list lies in the raw upload), in elements
    uint32_t n;      // list length
    uint32_t base;   // block id x 65536: what its 16-bit values count from
    uint32_t row;
};
// raw 16-bit lists -> global positions and row tags; one wave per block; *bad: a list that does not ascend strictly
__global__ __launch_bounds__(256) void lists_expand_kernel(const uint16_t* __restrict__ raw, const LxBlock* __restrict__ blocks,
                                                           uint32_t n_blocks, uint32_t* __restrict__ pos,
                                                           uint16_t* __restrict__ rtag, uint32_t* __restrict__ bad) {
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (b >= n_blocks) return;
    const LxBlock x = blocks[b];
    const uint16_t tag = (uint16_t)(x.row & 2047u);
    bool wrong = false;
    for (uint32_t k = threadIdx.x & 63u; k < x.n; k += 64u) {
        const uint32_t v = raw[x.at + k];
        if (k && raw[x.at + k - 1u] >= v) wrong = true;
        pos[x.at + k] = x.base + v;
        rtag[x.at + k] = tag;
    }
    if (wrong) atomicOr(bad, 1u);
}
}  // namespace

int storm_hip_rowlists_create_blocks(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                     const uint64_t* row_block_offset, const uint32_t* block_id,
                                     const uint8_t* block_kind, const uint32_t* block_n,
                                     const void* const* block_ptr, storm_hip_rowlists_t** out) {
    return storm_hip_rowlists_create_blocks_staged(ctx, n_rows, n_blocks, row_block_offset, block_id, block_kind, block_n, block_ptr,
                                                   nullptr, nullptr, out);
}

// [r6] ... with the lists the caller staged while it filled the container (storm_hip_stage_add_list: token[b] != ~0) gathered
// from the stage instead of carried over the bus now (4.4 of 8.6 ms of the build at 3145 positions per row)
int storm_hip_rowlists_create_blocks_staged(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                            const uint64_t* row_block_offset, const uint32_t* block_id,
                                            const uint8_t* block_kind, const uint32_t* block_n,
                                            const void* const* block_ptr, storm_hip_stage_t* stage, const uint64_t* token,
                                            storm_hip_rowlists_t** out) {
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

        // ---- rows as global positions. [r6] The host walks the BLOCKS only (where a list starts, how long it is, what its
        //      positions count from); the lists go up as they lie — 2 bytes per position through the pinned ring, packed by
        //      four threads while the copy before flies — and one kernel turns them into positions and row tags and checks that
        //      every list ascends strictly (a repeated position would toggle itself away). Until round 5 the host expanded
        //      every position itself and shipped 6 bytes for each out of pageable vectors: 118 ms of a first call at 3670
        //      positions per row where a steady call takes 6.
        auto T0 = std::chrono::steady_clock::now();
        const bool timing = getenv("STORM_HIP_TIMING") != nullptr;
        auto lap = [&](const char* what) {
            if (!timing) return;
            const auto t = std::chrono::steady_clock::now();
            fprintf(stderr, "[rowlists_create] %-34s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count());
            T0 = t;
        };
        std::vector<uint32_t> row_off(n_rows + 1), rowlen(n_rows);
        std::vector<LxBlock> xb;
        std::vector<std::pair<const void*, size_t>> run;   // the lists as they travel now, block after block ...
        std::vector<uint64_t> ltable;                      // ... or, when EVERY list is in the stage: destination element, token, length
        bool all_staged = stage != nullptr && token != nullptr;
        for (uint64_t b = 0; b < n_blocks && all_staged; ++b) all_staged = !block_n[b] || token[b] != ~0ull;
        xb.reserve(n_blocks);
        if (all_staged) ltable.reserve(3 * n_blocks);
        else run.reserve(n_blocks);
        {
            uint64_t e = 0;
            for (uint64_t r = 0; r < n_rows; ++r) {
                row_off[r] = (uint32_t)e;
                for (uint64_t b = row_block_offset[r]; b < row_block_offset[r + 1]; ++b) {
                    if (!block_n[b]) continue;
                    xb.push_back({(uint32_t)e, block_n[b], block_id[b] * 65536u, (uint32_t)r});
                    if (all_staged) ltable.insert(ltable.end(), {e, token[b], (uint64_t)block_n[b]});
                    else run.emplace_back(block_ptr[b], (size_t)block_n[b] * sizeof(uint16_t));
                    e += block_n[b];
                }
                rowlen[r] = (uint32_t)e - row_off[r];
            }
            row_off[n_rows] = (uint32_t)e;
        }
        lap("block walk");
        STORM_HIP_TRY(hipSetDevice(ctx->device));
        struct Temps {
            uint32_t *pos = nullptr, *row_off = nullptr, *cursor = nullptr, *bad = nullptr;   // (pos and row_off: the arena's own, see below)
            uint16_t* raw = nullptr;
            LxBlock* xb = nullptr;
            uint64_t* ltable = nullptr;
            storm_hip_ctx_t* ctx = nullptr;
            ~Temps() {   // (put off: a hipFree waits for the device, ~0.2 ms each — storm_hip_ctx_s::deferred_free)
                for (void* p : {(void*)cursor, (void*)bad, (void*)raw, (void*)xb, (void*)ltable})
                    if (p) ctx->deferred_free.push_back(p);
            }
        } t;
        t.ctx = ctx;
        drain_deferred(ctx, false);
        const size_t n_cells = (size_t)(n_groups + 1u) * n_windows;
        STORM_HIP_TRY(hipMalloc(&l->d_pos, n_elems * sizeof(uint32_t)));
        STORM_HIP_TRY(hipMalloc(&l->d_row_off, (n_rows + 1) * sizeof(uint32_t)));
        STORM_HIP_TRY(hipMalloc(&l->d_rtag, n_elems * sizeof(uint16_t)));
        t.pos = l->d_pos;
        t.row_off = l->d_row_off;
        STORM_HIP_TRY(hipMalloc(&t.raw, (n_elems + 8) * sizeof(uint16_t)));
        STORM_HIP_TRY(hipMalloc(&t.xb, xb.size() * sizeof(LxBlock)));
        STORM_HIP_TRY(hipMalloc(&t.bad, sizeof(uint32_t)));
        STORM_HIP_TRY(hipMalloc(&t.cursor, n_cells * sizeof(uint32_t)));
        STORM_HIP_TRY(hipMalloc(&l->d_elems, n_elems * sizeof(uint32_t)));
        STORM_HIP_TRY(hipMalloc(&l->d_off, n_cells * sizeof(uint32_t)));
        STORM_HIP_TRY(hipMalloc(&l->d_rowlen, n_rows * sizeof(uint32_t)));
        STORM_HIP_TRY(hipMemsetAsync(t.bad, 0, sizeof(uint32_t), ctx->stream));
        lap("allocations");
        {
            // (the block table — 1.3 MB at the README's STORM_t shape — through the pinned ring as well. Copied with
            //  hipMemcpyAsync out of the pageable vector and the vector freed at the end of this function, the FIRST matrix
            //  kernel after the build started 10 - 30 ms late on the card in 6 of 10 fresh processes, always after a pause:
            //  0.88 ms by events, 11 - 29 ms by the host, with hipEventQuery polling as late as hipStreamSynchronize; gone with
            //  glibc's MALLOC_MMAP_THRESHOLD_ / MALLOC_TRIM_THRESHOLD_ raised, gone with this line. An isolated copy +
            //  munmap + launch does not show it (tools/probes/pin_evict.hip); LAB_NOTES "K5 first call, the late kernel")
            Stager stager(ctx);
            if (int rc = stager.init()) return rc;
            if (int rc = stager.send_run(reinterpret_cast<uint8_t*>(t.xb), {{xb.data(), xb.size() * sizeof(LxBlock)}})) return rc;
            if (all_staged) {
                if (int rc = stage_gather_lists(ctx, stage, ltable, t.raw, &t.ltable)) return rc;
            } else if (int rc = stager.send_run(reinterpret_cast<uint8_t*>(t.raw), run)) {
                return rc;
            }
        }
        lap(all_staged ? "lists gathered from the stage" : "lists through the ring");
        hipLaunchKernelGGL(lists_expand_kernel, dim3((uint32_t)((xb.size() + 3) / 4)), dim3(256), 0, ctx->stream, t.raw, t.xb,
                           (uint32_t)xb.size(), t.pos, l->d_rtag, t.bad);
        STORM_HIP_TRY(hipGetLastError());
        STORM_HIP_TRY(hipMemcpyAsync(t.row_off, row_off.data(), (n_rows + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        STORM_HIP_TRY(hipMemcpyAsync(l->d_rowlen, rowlen.data(), n_rows * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        STORM_HIP_TRY(hipMemsetAsync(t.cursor, 0, n_cells * sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(lists_count_kernel, dim3((uint32_t)n_rows), dim3(256), 0, ctx->stream, t.pos, t.row_off,
                           (uint32_t)n_rows, n_windows, t.cursor);
        STORM_HIP_TRY(hipGetLastError());
        std::vector<uint32_t> cells(n_cells), offs(n_cells);
        uint32_t bad = 0;
        STORM_HIP_TRY(hipMemcpyAsync(cells.data(), t.cursor, n_cells * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        STORM_HIP_TRY(hipMemcpyAsync(&bad, t.bad, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream));
        STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        // (a list that is not strictly ascending — a row filled by out-of-order STORM_add calls — is no error: the container is
        //  not eligible, *out stays NULL and the dense replica, which sets bits in any order, takes the call as it did before
        //  this path existed)
        if (bad) return STORM_HIP_OK;
        lap("expand + count on the device");
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
        if (getenv("STORM_HIP_LISTS_NO_DEAL") == nullptr)
            hipLaunchKernelGGL(lists_deal_kernel, dim3(n_groups * n_windows), dim3(256), 0, ctx->stream, l->d_elems, l->d_off, n_windows);
        STORM_HIP_TRY(hipGetLastError());
        lap("offsets, place launched");
        const std::vector<LmItem> items = tile_order(n_rows, kLmGroup, kLmChunk);
        l->n_items = (uint32_t)items.size();
        STORM_HIP_TRY(hipMalloc(&l->d_items, items.size() * sizeof(LmItem)));
        STORM_HIP_TRY(hipMemcpyAsync(l->d_items, items.data(), items.size() * sizeof(LmItem), hipMemcpyHostToDevice, ctx->stream));
        // ---- K5h: the largest group size whose every group fits the hash table
        if (max_pos <= kLhMaxPos) {
            for (uint32_t g_log2 = 6; g_log2 >= 3 && !l->hash_g_log2; --g_log2) {
                const uint64_t G = 1ull << g_log2;
                uint64_t worst = 0;
                for (uint64_t r = 0; r < n_rows; r += G)
                    worst = std::max<uint64_t>(worst, row_off[std::min<uint64_t>(n_rows, r + G)] - row_off[r]);
                if (worst <= kLhFill) l->hash_g_log2 = g_log2;
            }
        }
        if (l->hash_g_log2) {
            const std::vector<LmItem> hitems = tile_order(n_rows, 1u << l->hash_g_log2, kLhCounters >> l->hash_g_log2);
            l->n_hash_items = (uint32_t)hitems.size();
            STORM_HIP_TRY(hipMalloc(&l->d_hash_items, hitems.size() * sizeof(LmItem)));
            STORM_HIP_TRY(hipMemcpyAsync(l->d_hash_items, hitems.data(), hitems.size() * sizeof(LmItem), hipMemcpyHostToDevice, ctx->stream));
        }
        lap("tile lists");
        STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        lap("place on the device");
        *out = owner.release();
        return STORM_HIP_OK;
    });
}

// 1: the lists are expected to beat the dense replica's multiply (or the option says so); 0: keep the dense path
int storm_hip_rowlists_worthwhile_counts(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_elems, uint64_t n_bits) {
    if (!ctx || ctx->matrix_lists == 0) return 0;
    if (ctx->matrix_lists > 0) return 1;
    // density of the dense replica the lists stand for (its rows are padded to 512 bits); crossover measured at the
    // README's STORM_t shape: profiles/r05_j_storm_matrix_lists.jsonl
    const double bits = (double)((n_bits + 511u) / 512u * 512u) * (double)n_rows;
    return (double)n_elems <= bits * (double)ctx->matrix_lists_permille_x10 / 10000.0;
}
int storm_hip_rowlists_worthwhile(storm_hip_ctx_t* ctx, const storm_hip_rowlists_t* l) {
    if (!ctx || ctx->matrix_lists == 0) return 0;
    if (!l) return 1;   // "is the path switched on at all"
    return storm_hip_rowlists_worthwhile_counts(ctx, l->n_rows, l->n_elems, l->n_bits);
}

static int launch_lists(storm_hip_ctx_t* ctx, const storm_hip_rowlists_t* l, int op, uint32_t* d_out, uint64_t ld) {
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
    // which of the two: the hash kernel where the rows are short (its work grows with the square of the row length, the
    // window kernel's with the row length on top of a fixed cost per step); option matrix_lists_kernel forces one
    const bool hash = l->hash_g_log2 != 0 && ctx->matrix_lists_kernel != 1 &&
                      (ctx->matrix_lists_kernel == 2 || l->hash_g_log2 >= (uint32_t)ctx->matrix_lists_hash_min_log2);
    kernel_time_mark(ctx);
    if (hash)
        hipLaunchKernelGGL(lists_hash_kernel, dim3(l->n_hash_items), dim3(kLhThreads), 0, ctx->stream, l->d_pos, l->d_rtag,
                           l->d_row_off, l->d_rowlen, l->d_hash_items, (uint32_t)l->n_rows, l->hash_g_log2, op, d_out, ld, (uint32_t)ctx->matrix_lists_debug);
    else
        hipLaunchKernelGGL(lists_matrix_kernel, dim3(l->n_items), dim3(kLmThreads), 0, ctx->stream, l->d_elems, l->d_off,
                           l->n_windows, l->d_rowlen, l->d_items, (uint32_t)l->n_rows, op, d_out, ld, (uint32_t)ctx->matrix_lists_debug);
    kernel_time_mark(ctx);
    STORM_HIP_TRY(hipGetLastError());
    ctx->pass_report[0] = STORM_HIP_RAN_LISTS_MATRIX;
    ctx->pass_report[1] = 0;
    const uint64_t group_rows = hash ? 1ull << l->hash_g_log2 : kLmGroup;
    ctx->pass_report[2] = (l->n_rows + group_rows - 1u) / group_rows * l->n_elems / 2u;
    ctx->pass_report[3] = group_rows;
    return STORM_HIP_OK;
}

int storm_hip_rowlists_pairw_matrix_device(storm_hip_ctx_t* ctx, const storm_hip_rowlists_t* l, int op,
                                           uint32_t* d_out, uint64_t ld) {
    if (getenv("STORM_HIP_TIMING") != nullptr) {   // the kernel alone by the card's clock beside the host's view of launch + wait
        hipEvent_t e0 = nullptr, e1 = nullptr;
        STORM_HIP_TRY(hipSetDevice(ctx->device));
        STORM_HIP_TRY(hipEventCreate(&e0));
        STORM_HIP_TRY(hipEventCreate(&e1));
        const auto t0 = std::chrono::steady_clock::now();
        STORM_HIP_TRY(hipEventRecord(e0, ctx->stream));
        if (int rc = launch_lists(ctx, l, op, d_out, ld)) return rc;
        STORM_HIP_TRY(hipEventRecord(e1, ctx->stream));
        STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        float ms = 0;
        STORM_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        fprintf(stderr, "[rowlists_matrix] kernel by events %8.3f ms, launch + wait on the host %8.3f ms\n", ms,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return STORM_HIP_OK;
    }
    if (int rc = launch_lists(ctx, l, op, d_out, ld)) return rc;
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STORM_HIP_OK;
}

// The same into HOST memory, whole rows (zeros at i >= j, as storm_hip_pairw_matrix writes them): the matrix is built in
// the context's band buffer and copied out.
int storm_hip_rowlists_pairw_matrix(storm_hip_ctx_t* ctx, const storm_hip_rowlists_t* l, int op, uint32_t* h_out,
                                    uint64_t ld) {
    if (!ctx || !l || !h_out) {
        set_error("rowlists_pairw_matrix: NULL argument");
        return STORM_HIP_EINVAL;
    }
    const uint64_t n = l->n_rows;
    if (ld < n) {
        set_error("rowlists_pairw_matrix: ld %llu < %llu rows", (unsigned long long)ld, (unsigned long long)n);
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    const size_t need = (size_t)n * n * sizeof(uint32_t);
    if (need > ctx->band_capacity) {
        if (ctx->d_band) STORM_HIP_TRY(hipFree(ctx->d_band));
        ctx->d_band = nullptr;
        ctx->band_capacity = 0;
        if (hipMalloc(reinterpret_cast<void**>(&ctx->d_band), need) != hipSuccess) {
            set_error("rowlists_pairw_matrix: hipMalloc of %zu bytes for the output failed", need);
            return STORM_HIP_ENOMEM;
        }
        ctx->band_capacity = need;
    }
    STORM_HIP_TRY(hipMemsetAsync(ctx->d_band, 0, need, ctx->stream));
    if (int rc = launch_lists(ctx, l, op, ctx->d_band, n)) return rc;
    STORM_HIP_TRY(hipMemcpy2DAsync(h_out, ld * sizeof(uint32_t), ctx->d_band, n * sizeof(uint32_t), n * sizeof(uint32_t), n,
                                   hipMemcpyDeviceToHost, ctx->stream));
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STORM_HIP_OK;
}

uint64_t storm_hip_rowlists_n_elems(const storm_hip_rowlists_t* l) { return l ? l->n_elems : 0; }

}  // extern "C"
