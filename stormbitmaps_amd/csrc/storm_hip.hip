// storm_hip.hip — gfx950 (MI355X / CDNA4) kernels and the C-ABI shim for the dense pairwise
// AND+popcount path (include/storm_hip.h).
//
// K1 "pairw_dense": sum_{i<j} popcount(X_i & X_j) over a row-major uint64 bitmap matrix.
//   Replaces the blocked upper-triangle loop of STORM_contig_pairw_intersect_cardinality_blocked
//   (storm.c:1175-1241), STORM_wrapper_diag_blocked (storm.c:222-279) and the libalgebra leaf
//   they call per pair (STORM_compute_func; storm.c:1205,1217,1227,1236).
//
//   Work item  = (segment, k-slice). A segment pairs a 128-row A block with a run of up to
//                seg_rows later B rows; a k-slice is `cps` chunks of 64 words.
//   Workgroup  = 4 waves. Lane l owns word (chunk*64 + l) of every row it touches, so every
//                global access is a fully coalesced 512-byte row slice and no cross-lane data
//                movement is needed until the final reduction.
//   A operand  = 32 rows per wave, kept in 64 VGPRs for the whole chunk (register-stationary).
//   B operand  = streamed: 32-row stages of the B run go global -> LDS (global_load_lds,
//                16 B/lane, double-buffered) once per workgroup and are read by all 4 waves
//                with conflict-free ds_read_b64.
//   Math       = per A row and B word: 2x v_and_b32 + 2x v_bcnt_u32_b32 (accumulating form),
//                i.e. 128 VALU instructions per ds_read_b64 — the kernel is VALU-issue bound
//                by construction; HBM/L2/LDS traffic is a few percent of their peaks.
//   Reduction  = per-lane uint32 -> wave shuffle -> one 64-bit atomic per wave into one of
//                4096 slots -> tiny second kernel folds the slots into the result word.
#include "storm_hip_internal.h"

#include <chrono>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdlib>
#include <cstring>

namespace storm {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
// acc += popcount(a & b) for word pairs: 2x v_and_b32 + 2x v_bcnt_u32_b32 per 64-bit pair.
// Hand-written because both of these matter on gfx950 (measured with tools/ubench_regs,
// profiles/r01_b_ubench_regs.txt):
//  * the bcnt must be the ACCUMULATING form (D = popcount(S0) + S1); left to itself hipcc
//    emits bcnt(x, 0) twice plus a v_add3_u32 — 5 VALU instructions per word pair, not 4;
//  * one `s_nop 0` between each and and its bcnt: a pure and,bcnt,and,bcnt stream issues at
//    ~4.3 cycles per instruction per SIMD, the same stream with the scalar no-op at ~3.05
//    (= 2 for the and + 4 for the half-rate bcnt, i.e. the VALU ceiling): +40 %.
// Four A rows per asm statement so that hipcc's own conservative s_nop between two asm
// statements appears once per 16 VALU instructions instead of after every pair.
#define STORM_PAIR(A, B)                 \
    "v_and_b32 %[t], %[" A "], %[" B "]\n\t" \
    "s_nop 0\n\t"                        \
    "v_bcnt_u32_b32 %[acc], %[t], %[acc]\n\t"

__device__ __forceinline__ uint32_t popc_and4(uint32_t l0, uint32_t h0, uint32_t l1, uint32_t h1,
                                              uint32_t l2, uint32_t h2, uint32_t l3, uint32_t h3,
                                              uint32_t b_lo, uint32_t b_hi, uint32_t acc) {
    uint32_t t;
    asm(STORM_PAIR("l0", "bl") STORM_PAIR("h0", "bh") STORM_PAIR("l1", "bl") STORM_PAIR("h1", "bh")
        STORM_PAIR("l2", "bl") STORM_PAIR("h2", "bh") STORM_PAIR("l3", "bl") STORM_PAIR("h3", "bh")
        : [acc] "+v"(acc), [t] "=&v"(t)
        : [l0] "v"(l0), [h0] "v"(h0), [l1] "v"(l1), [h1] "v"(h1), [l2] "v"(l2), [h2] "v"(h2),
          [l3] "v"(l3), [h3] "v"(h3), [bl] "v"(b_lo), [bh] "v"(b_hi));
    return acc;
}

__device__ __forceinline__ uint32_t popc_and(uint32_t a_lo, uint32_t a_hi, uint32_t b_lo,
                                             uint32_t b_hi, uint32_t acc) {
    uint32_t t;
    asm(STORM_PAIR("l0", "bl") STORM_PAIR("h0", "bh")
        : [acc] "+v"(acc), [t] "=&v"(t)
        : [l0] "v"(a_lo), [h0] "v"(a_hi), [bl] "v"(b_lo), [bh] "v"(b_hi));
    return acc;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kLanes);
    return v;  // valid in lane 0
}

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

// One LDS stage = kStageRows rows x 512 B = 16 KiB = 16 wave-instructions of 1 KiB.
// Wave w issues instructions q*4+w (q = 0..3). Instruction n fills LDS bytes [n*1024, +1024):
// 16-byte piece p = n*64 + lane is row p/32, 16-byte column p%32 of the stage.
__device__ __forceinline__ void stage_b_glds(uint64_t* lds_stage, const uint64_t* __restrict__ X,
                                             uint64_t stride, uint32_t kbase, uint32_t j0,
                                             uint32_t j_last, uint32_t wave, uint32_t lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t n = (uint32_t)q * 4u + wave;
        const uint32_t p = n * 64u + lane;
        uint32_t j = j0 + (p >> 5);
        j = j < j_last ? j : j_last;  // rows past the run re-read the last row; never consumed
        const char* g = reinterpret_cast<const char*>(X + (uint64_t)j * stride + kbase) +
                        (p & 31u) * 16u;
        char* l = reinterpret_cast<char*>(lds_stage) + n * 1024u;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
    }
}

// Same stage through VGPRs (no LDS-DMA): global_load_dwordx4 + ds_write_b128.
__device__ __forceinline__ void stage_b_regs(uint64_t* lds_stage, const uint64_t* __restrict__ X,
                                             uint64_t stride, uint32_t kbase, uint32_t j0,
                                             uint32_t j_last, uint32_t tid) {
    uint4 tmp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t p = (uint32_t)q * 256u + tid;
        uint32_t j = j0 + (p >> 5);
        j = j < j_last ? j : j_last;
        const char* g = reinterpret_cast<const char*>(X + (uint64_t)j * stride + kbase) +
                        (p & 31u) * 16u;
        tmp[q] = *reinterpret_cast<const uint4*>(g);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t p = (uint32_t)q * 256u + tid;
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(lds_stage) + p * 16u) = tmp[q];
    }
}

// All kRowsPerWave A rows against one B word. DIAG: only rows r < rmax (wave-uniform).
template <bool DIAG>
__device__ __forceinline__ void row_step(const uint32_t (&a_lo)[kRowsPerWave],
                                         const uint32_t (&a_hi)[kRowsPerWave], uint64_t b,
                                         int rmax, uint32_t (&acc)[4]) {
    const uint32_t b_lo = (uint32_t)b, b_hi = (uint32_t)(b >> 32);
    if (!DIAG) {
#pragma unroll
        for (int r = 0; r < kRowsPerWave; r += 4)
            acc[(r >> 2) & 3] = popc_and4(a_lo[r], a_hi[r], a_lo[r + 1], a_hi[r + 1], a_lo[r + 2],
                                          a_hi[r + 2], a_lo[r + 3], a_hi[r + 3], b_lo, b_hi,
                                          acc[(r >> 2) & 3]);
    } else {
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r)
            if (r < rmax) acc[r & 3] = popc_and(a_lo[r], a_hi[r], b_lo, b_hi, acc[r & 3]);
    }
}

// A full 32-row stage: groups of 4 B words, the next group's ds_reads issued before the
// current group's 512 VALU instructions so the LDS latency is never exposed.
__device__ __forceinline__ void stage_compute(const uint32_t (&a_lo)[kRowsPerWave],
                                              const uint32_t (&a_hi)[kRowsPerWave],
                                              const uint64_t* __restrict__ col,
                                              uint32_t (&acc)[4]) {
    constexpr int G = 4;
    uint64_t cur[G], nxt[G];
#pragma unroll
    for (int u = 0; u < G; ++u) cur[u] = col[u * kChunkWords];
#pragma unroll
    for (int g = 0; g < kStageRows / G; ++g) {
        if (g + 1 < kStageRows / G) {
#pragma unroll
            for (int u = 0; u < G; ++u) nxt[u] = col[((g + 1) * G + u) * kChunkWords];
        }
#pragma unroll
        for (int u = 0; u < G; ++u) row_step<false>(a_lo, a_hi, cur[u], 0, acc);
#pragma unroll
        for (int u = 0; u < G; ++u) cur[u] = nxt[u];
    }
}

// VARIANT 0: B rows straight from global memory into VGPRs (each wave loads its own copy)
// VARIANT 1: B stages through VGPRs into LDS (single buffer, two barriers per stage)
// VARIANT 2: B stages by global_load_lds, double-buffered, one barrier per stage
template <int VARIANT>
__global__ __launch_bounds__(kThreads, 4) void pairw_dense_kernel(
    const uint64_t* __restrict__ X, uint64_t stride, const Seg* __restrict__ segs,
    uint32_t n_segs, uint32_t n_chunks, uint32_t cps, unsigned long long* __restrict__ slots) {
    __shared__ uint64_t lds[VARIANT == 0 ? 1 : (VARIANT == 1 ? 1 : 2)]
                           [VARIANT == 0 ? 1 : kStageRows * kChunkWords];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t item = blockIdx.x;
    const uint32_t ks = item / n_segs;  // k-slice major: concurrent workgroups share columns
    const Seg seg = segs[item - ks * n_segs];
    const uint32_t a_row = seg.a_row0 + wave * kRowsPerWave;
    const uint32_t j_lo = seg.j_lo, j_hi = seg.j_hi, j_last = seg.j_hi - 1;
    const bool diag = seg.j_lo == seg.a_row0;

    uint32_t acc[4] = {0, 0, 0, 0};
    const uint32_t c_end = min(n_chunks, (ks + 1) * cps);
    for (uint32_t c = ks * cps; c < c_end; ++c) {
        const uint32_t kbase = c * kChunkWords;
        // ---- A block: 32 rows of this wave, one 64-bit word per lane, into VGPRs ----
        uint32_t a_lo[kRowsPerWave], a_hi[kRowsPerWave];
        {
            const uint64_t* ap = X + (uint64_t)a_row * stride + kbase + lane;
#pragma unroll
            for (int r = 0; r < kRowsPerWave; ++r) {
                uint64_t v = ap[(uint64_t)r * stride];
                if (a_row + r >= seg.a_end) v = 0;  // wave-uniform: rows past the block edge
                a_lo[r] = (uint32_t)v;
                a_hi[r] = (uint32_t)(v >> 32);
            }
        }

        if (VARIANT == 0) {
            const uint64_t* bp = X + kbase + lane;
            uint64_t cur[4], nxt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) cur[u] = bp[(uint64_t)min(j_lo + u, j_last) * stride];
            for (uint32_t j = j_lo; j < j_hi; j += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    nxt[u] = bp[(uint64_t)min(j + 4 + u, j_last) * stride];
                if (!diag && j + 4 <= j_hi) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) row_step<false>(a_lo, a_hi, cur[u], 0, acc);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (j + u < j_hi) {
                            const int rmax = diag ? (int)(j + u) - (int)a_row : kRowsPerWave;
                            row_step<true>(a_lo, a_hi, cur[u], rmax, acc);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
            }
        } else {
            const uint32_t n_stages = (j_hi - j_lo + kStageRows - 1) / kStageRows;
            if (VARIANT == 2) stage_b_glds(lds[0], X, stride, kbase, j_lo, j_last, wave, lane);
            for (uint32_t s = 0; s < n_stages; ++s) {
                const uint32_t j0 = j_lo + s * kStageRows;
                const uint64_t* buf;
                if (VARIANT == 2) {
                    // Drain this wave's LDS-DMA, then the barrier makes every wave's share of
                    // stage s visible and proves every wave is done reading the buffer that
                    // stage s+1 is about to overwrite. The wait is explicit: hipcc (ROCm 7.2)
                    // emits only lgkmcnt(0) for __syncthreads() inside this loop, and an
                    // LDS-DMA is tracked by vmcnt.
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (s + 1 < n_stages)
                        stage_b_glds(lds[(s + 1) & 1], X, stride, kbase, j0 + kStageRows, j_last,
                                     wave, lane);
                    buf = lds[s & 1];
                } else {
                    __syncthreads();  // previous stage fully consumed
                    stage_b_regs(lds[0], X, stride, kbase, j0, j_last, tid);
                    __syncthreads();
                    buf = lds[0];
                }
                const uint32_t rows = min((uint32_t)kStageRows, j_hi - j0);
                if (!diag && rows == kStageRows) {
                    stage_compute(a_lo, a_hi, buf + lane, acc);
                } else {
                    for (uint32_t jj = 0; jj < rows; ++jj) {
                        const int rmax = diag ? (int)(j0 + jj) - (int)a_row : kRowsPerWave;
                        row_step<true>(a_lo, a_hi, buf[jj * kChunkWords + lane], rmax, acc);
                    }
                }
            }
            // the next chunk's first stage overwrites lds[0]: wait for every wave to finish
            if (c + 1 < c_end) __syncthreads();
        }
    }

    const uint64_t mine = (uint64_t)acc[0] + acc[1] + acc[2] + acc[3];
    const uint64_t wsum = wave_sum_u64(mine);
    if (lane == 0 && wsum != 0)
        atomicAdd(&slots[(item * kWaves + wave) & (kSlots - 1)], (unsigned long long)wsum);
}

// Folds the slot array into out[0] and re-zeroes the slots for the next launch.
__global__ __launch_bounds__(1024) void fold_slots_kernel(unsigned long long* __restrict__ slots,
                                                          unsigned long long* __restrict__ out) {
    __shared__ unsigned long long part[16];
    unsigned long long v = 0;
    for (int i = threadIdx.x; i < kSlots; i += 1024) {
        v += slots[i];
        slots[i] = 0;
    }
    if (threadIdx.x < kSlotsExtra) slots[kSlots + threadIdx.x] = 0;  // queue heads of the strip kernel
    v = wave_sum_u64(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; ++w) t += part[w];
        out[0] = t;
    }
}

// Rectangle A x B^T (STORM_wrapper_square): same inner loop, segments pair A blocks of `xa`
// with row runs of `xb`; never diagonal.
__global__ __launch_bounds__(kThreads, 4) void square_dense_kernel(
    const uint64_t* __restrict__ XA, uint64_t stride_a, const uint64_t* __restrict__ XB,
    uint64_t stride_b, uint32_t n_rows_b, uint32_t seg_rows, uint32_t segs_per_ablock,
    uint32_t n_segs, uint32_t n_chunks, unsigned long long* __restrict__ slots) {
    __shared__ uint64_t lds[2][kStageRows * kChunkWords];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t item = blockIdx.x;
    const uint32_t c = item / n_segs;
    const uint32_t si = item - c * n_segs;
    const uint32_t a_row = (si / segs_per_ablock) * kABlockRows + wave * kRowsPerWave;
    const uint32_t j_lo = (si % segs_per_ablock) * seg_rows;
    const uint32_t j_hi = min(n_rows_b, j_lo + seg_rows);
    const uint32_t j_last = j_hi - 1;
    const uint32_t kbase = c * kChunkWords;

    uint32_t acc[4] = {0, 0, 0, 0};
    uint32_t a_lo[kRowsPerWave], a_hi[kRowsPerWave];
    {
        const uint64_t* ap = XA + (uint64_t)a_row * stride_a + kbase + lane;
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
            const uint64_t v = ap[(uint64_t)r * stride_a];
            a_lo[r] = (uint32_t)v;
            a_hi[r] = (uint32_t)(v >> 32);
        }
    }
    const uint32_t n_stages = (j_hi - j_lo + kStageRows - 1) / kStageRows;
    stage_b_glds(lds[0], XB, stride_b, kbase, j_lo, j_last, wave, lane);
    for (uint32_t s = 0; s < n_stages; ++s) {
        const uint32_t j0 = j_lo + s * kStageRows;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < n_stages)
            stage_b_glds(lds[(s + 1) & 1], XB, stride_b, kbase, j0 + kStageRows, j_last, wave,
                         lane);
        const uint64_t* buf = lds[s & 1];
        const uint32_t rows = min((uint32_t)kStageRows, j_hi - j0);
        if (rows == kStageRows) {
            stage_compute(a_lo, a_hi, buf + lane, acc);
        } else {
            for (uint32_t jj = 0; jj < rows; ++jj)
                row_step<false>(a_lo, a_hi, buf[jj * kChunkWords + lane], 0, acc);
        }
    }
    const uint64_t wsum = wave_sum_u64((uint64_t)acc[0] + acc[1] + acc[2] + acc[3]);
    if (lane == 0 && wsum != 0)
        atomicAdd(&slots[(item * kWaves + wave) & (kSlots - 1)], (unsigned long long)wsum);
}

// Per-pair counts of one tile (tests / materialised output). One wave per pair.
__global__ __launch_bounds__(kThreads) void tile_counts_kernel(
    const uint64_t* __restrict__ X, uint64_t stride, uint32_t n_words, uint64_t i0, uint64_t j0,
    uint32_t ni, uint32_t nj, uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t pair = (uint64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (pair >= (uint64_t)ni * nj) return;
    const uint64_t* a = X + (i0 + pair / nj) * stride;
    const uint64_t* b = X + (j0 + pair % nj) * stride;
    uint32_t cnt = 0;
    for (uint32_t k = lane; k < n_words; k += kLanes) cnt += (uint32_t)__popcll(a[k] & b[k]);
    const uint64_t s = wave_sum_u64(cnt);
    if (lane == 0) out[pair] = (uint32_t)s;
}

// sum_c C(n_c, 2): verification identity only. One wave per 64-bit word column,
// lane = bit; 4 word columns per workgroup.
__global__ __launch_bounds__(kThreads) void column_identity_kernel(
    const uint64_t* __restrict__ X, uint64_t stride, uint64_t n_rows, uint32_t n_words,
    unsigned long long* __restrict__ slots) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = blockIdx.x * kWaves + (threadIdx.x >> 6);
    uint64_t n = 0;
    if (w < n_words) {
        for (uint64_t i = 0; i < n_rows; ++i) n += (X[i * stride + w] >> lane) & 1ull;
    }
    const uint64_t s = wave_sum_u64(n * (n - (n != 0)) / 2);
    if (lane == 0 && s != 0) atomicAdd(&slots[w & (kSlots - 1)], (unsigned long long)s);
}

// Per-row set-bit counts n_i (one wave per row). The sibling pair counts follow from the
// intersect count by inclusion-exclusion: |a|b| = n_a + n_b - |a&b|, |a^b| = n_a + n_b - 2|a&b|.
__global__ __launch_bounds__(kThreads) void row_counts_kernel(const uint64_t* __restrict__ X,
                                                              uint64_t stride, uint64_t n_rows,
                                                              uint32_t n_words,
                                                              uint32_t* __restrict__ counts) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t r = (uint64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    uint64_t n = 0;
    for (uint32_t k = lane; k < n_words; k += 64u) n += (uint64_t)__popcll(X[r * stride + k]);
    n = wave_sum_u64(n);
    if (lane == 0) counts[r] = (uint32_t)n;
}

// OR sorted position lists (CSR) into rows [row0, ...): one workgroup per row.
__global__ __launch_bounds__(kThreads) void set_bits_kernel(uint64_t* __restrict__ X,
                                                            uint64_t stride, uint64_t row0,
                                                            const uint64_t* __restrict__ offsets,
                                                            const uint32_t* __restrict__ pos) {
    const uint64_t r = blockIdx.x;
    unsigned long long* row = reinterpret_cast<unsigned long long*>(X + (row0 + r) * stride);
    const uint64_t base = offsets[0];
    for (uint64_t p = offsets[r] - base + threadIdx.x; p < offsets[r + 1] - base; p += kThreads) {
        const uint32_t v = pos[p];
        atomicOr(&row[v >> 6], 1ull << (v & 63u));
    }
}

// Synthetic fill, bit-identical to storm_synth_fill_dense (storm_synth.h).
__device__ __forceinline__ uint64_t synth_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(kThreads) void synth_fill_kernel(uint64_t* __restrict__ X,
                                                              uint64_t stride, uint64_t n_rows,
                                                              uint64_t n_bits, uint32_t draws,
                                                              uint64_t seed) {
    const uint64_t total = n_rows * draws;
    for (uint64_t t = (uint64_t)blockIdx.x * kThreads + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * kThreads) {
        const uint64_t row = t / draws;
        const uint64_t z = synth_mix(seed + (t + 1) * 0x9E3779B97F4A7C15ull);
        const uint64_t v = __umul64hi(z, n_bits);
        atomicOr(reinterpret_cast<unsigned long long*>(X + row * stride + (v >> 6)),
                 1ull << (v & 63u));
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int ensure_segments(storm_hip_ctx_t* ctx, uint64_t n_rows, uint32_t shard_rank,
                           uint32_t shard_count) {
    const uint32_t seg_len = (uint32_t)ctx->seg_rows;
    if (ctx->d_segs && ctx->seg_rows_n == n_rows && ctx->seg_shard_rank == shard_rank &&
        ctx->seg_shard_count == shard_count && ctx->seg_len == seg_len)
        return STORM_HIP_OK;

    // Full segments first (A-block major), the short diagonal segments last; the shard takes
    // every shard_count-th entry of each list, so the shards partition the pair space.
    std::vector<Seg> full, diag, mine;
    for (uint64_t a0 = 0; a0 < n_rows; a0 += kABlockRows) {
        const uint32_t a_end = (uint32_t)std::min<uint64_t>(a0 + kABlockRows, n_rows);
        if (a_end - a0 > 1) diag.push_back({(uint32_t)a0, a_end, (uint32_t)a0, a_end});
        for (uint64_t j = a0 + kABlockRows; j < n_rows; j += seg_len)
            full.push_back({(uint32_t)a0, a_end, (uint32_t)j,
                            (uint32_t)std::min<uint64_t>(j + seg_len, n_rows)});
    }
    for (size_t i = shard_rank; i < full.size(); i += shard_count) mine.push_back(full[i]);
    // rotate the diagonal list so that rank r does not always start at the same A block
    for (size_t i = shard_rank; i < diag.size(); i += shard_count) mine.push_back(diag[i]);

    uint64_t row_sum = 0;
    for (const Seg& s : mine) row_sum += s.j_hi - s.j_lo;
    if (mine.size() > ctx->segs_capacity) {
        if (ctx->d_segs) STORM_HIP_TRY(hipFree(ctx->d_segs));
        ctx->d_segs = nullptr;
        ctx->segs_capacity = 0;
        const size_t cap = std::max<size_t>(mine.size(), 1024);
        STORM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_segs), cap * sizeof(Seg)));
        ctx->segs_capacity = cap;
    }
    if (!mine.empty()) {
        // pageable host memory: the copy is complete (staged) when the call returns
        STORM_HIP_TRY(hipMemcpyAsync(ctx->d_segs, mine.data(), mine.size() * sizeof(Seg),
                                     hipMemcpyHostToDevice, ctx->stream));
        STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    ctx->n_segs = (uint32_t)mine.size();
    ctx->seg_row_sum = row_sum;
    ctx->seg_rows_n = n_rows;
    ctx->seg_shard_rank = shard_rank;
    ctx->seg_shard_count = shard_count;
    ctx->seg_len = seg_len;
    return STORM_HIP_OK;
}

void drain_deferred(storm_hip_ctx_t* ctx, bool aged) {
    if (aged && ctx->deferred_age++ == 0) return;   // (the call that deferred them: leave them to the next one)
    for (void* p : ctx->deferred_free) (void)hipFree(p);
    for (void* p : ctx->deferred_host_free) (void)hipHostFree(p);
    ctx->deferred_free.clear();
    ctx->deferred_host_free.clear();
    ctx->deferred_age = 0;
}

static int check_ctx(const storm_hip_ctx_t* ctx) {
    if (!ctx) {
        set_error("NULL context");
        return STORM_HIP_EINVAL;
    }
    return STORM_HIP_OK;
}

}  // namespace storm

using namespace storm;

// ==========================================================================================
// C-ABI
// ==========================================================================================
extern "C" {

const char* storm_hip_last_error(void) { return g_error; }

int storm_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int storm_hip_device_arch(int device, char* buf, size_t buflen) {
    if (!buf || buflen == 0) return STORM_HIP_EINVAL;
    hipDeviceProp_t prop;
    STORM_HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s", prop.gcnArchName);
    return STORM_HIP_OK;
}

int storm_hip_device_pci_bus_id(int device, char* buf, size_t buflen) {
    if (!buf || buflen < 13) return STORM_HIP_EINVAL;
    STORM_HIP_TRY(hipDeviceGetPCIBusId(buf, (int)buflen, device));
    return STORM_HIP_OK;
}

namespace storm {
__global__ void warm_core_kernel() {}
}

int storm_hip_ctx_create(int device, void* stream, storm_hip_ctx_t** out) {
    if (!out) {
        set_error("storm_hip_ctx_create: NULL out");
        return STORM_HIP_EINVAL;
    }
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("no HIP device visible (this library has no CPU fallback)");
        return STORM_HIP_ENODEV;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (%d visible)", device, n);
        return STORM_HIP_EINVAL;
    }
    hipDeviceProp_t prop;
    STORM_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library carries gfx950 code objects only", device,
                  prop.gcnArchName);
        return STORM_HIP_ENODEV;
    }
    STORM_HIP_TRY(hipSetDevice(device));
    storm_hip_ctx_t* ctx = new (std::nothrow) storm_hip_ctx_t();
    if (!ctx) return STORM_HIP_ENOMEM;
    ctx->device = device;
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
    ctx->n_cus = prop.multiProcessorCount;
    if (hipMalloc(reinterpret_cast<void**>(&ctx->d_slots), (kSlots + kSlotsExtra) * sizeof(uint64_t)) !=
            hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&ctx->d_scalar), 64) != hipSuccess) {
        set_error("workspace allocation failed");
        storm_hip_ctx_destroy(ctx);
        return STORM_HIP_ENOMEM;
    }
    if (hipMemset(ctx->d_slots, 0, (kSlots + kSlotsExtra) * sizeof(uint64_t)) != hipSuccess ||
        hipMemset(ctx->d_scalar, 0, 64) != hipSuccess) {
        set_error("workspace memset failed");
        storm_hip_ctx_destroy(ctx);
        return STORM_HIP_EHIP;
    }
    // the code objects of all translation units and the result mailbox now, not inside the first call (it costs the
    // creation a few ms once; STORM_HIP_NO_WARM: the lazy loading of rounds 1 - 5, for tools/bench_cold.py's A/B)
    if (!getenv("STORM_HIP_NO_WARM")) {
        hipLaunchKernelGGL(warm_core_kernel, dim3(1), dim3(64), 0, ctx->stream);
        warm_mfma_code(ctx->stream);
        warm_sparse_code(ctx->stream);
        warm_lists_code(ctx->stream);
        (void)result_target(ctx);
        ctx->mail_armed = false;
        // (the second stream of the raw-buffer wrappers: creating it took 6.7 ms of their first call)
        (void)hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
        // ... and the runtime's path for copies out of pageable host memory: the first such copy of a process (a work list on
        // its way to the device) took 7.5 ms — the whole first-call penalty of a 1024-row matrix (rocprofv3 --hip-trace)
        {
            std::vector<uint64_t> pageable(kSlots, 0);
            (void)hipMemcpyAsync(ctx->d_slots, pageable.data(), pageable.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
            set_error("warm-up launches failed");
            storm_hip_ctx_destroy(ctx);
            return STORM_HIP_EHIP;
        }
    }
    if (const char* v = getenv("STORM_HIP_VARIANT")) ctx->variant = atoi(v);
    if (const char* v = getenv("STORM_HIP_SEG_ROWS")) ctx->seg_rows = atoi(v);
    if (const char* v = getenv("STORM_HIP_CHUNKS_PER_ITEM")) ctx->chunks_per_item = atoi(v);
    *out = ctx;
    return STORM_HIP_OK;
}

int storm_hip_ctx_set_stream(storm_hip_ctx_t* ctx, void* stream) {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
    return STORM_HIP_OK;
}

int storm_hip_ctx_synchronize(storm_hip_ctx_t* ctx) {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STORM_HIP_OK;
}

int storm_hip_ctx_reserve_staging(storm_hip_ctx_t* ctx) {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (ctx->h_stage_ring) return STORM_HIP_OK;
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    if (hipHostMalloc(&ctx->h_stage_ring, (size_t)24 << 20, hipHostMallocDefault) != hipSuccess) {
        ctx->h_stage_ring = nullptr;
        set_error("reserve_staging: hipHostMalloc of the 24 MiB staging ring failed");
        return STORM_HIP_ENOMEM;
    }
    return STORM_HIP_OK;
}

void storm_hip_ctx_destroy(storm_hip_ctx_t* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    drain_deferred(ctx, false);
    if (ctx->d_slots) (void)hipFree(ctx->d_slots);
    if (ctx->d_scalar) (void)hipFree(ctx->d_scalar);
    if (ctx->d_positions) (void)hipFree(ctx->d_positions);
    if (ctx->h_scalar) (void)hipHostFree(ctx->h_scalar);
    if (ctx->h_mail) (void)hipHostFree(ctx->h_mail);
    if (ctx->h_stage_ring) (void)hipHostFree(ctx->h_stage_ring);
    for (hipEvent_t e : ctx->stage_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->d_segs) (void)hipFree(ctx->d_segs);
    release_mfma_state(ctx);
    for (hipEvent_t ev : ctx->kernel_events) (void)hipEventDestroy(ev);
    delete ctx;
}

int storm_hip_option_check(const char* key, int64_t value) {
    // the same validation on a scratch context (storm_hip_ctx_set_option touches no device)
    storm_hip_ctx_s scratch;
    scratch.n_cus = 256;
    return storm_hip_ctx_set_option(&scratch, key, value);
}

int storm_hip_ctx_set_option(storm_hip_ctx_t* ctx, const char* key, int64_t value) {
    if (check_ctx(ctx) || !key) return STORM_HIP_EINVAL;
    if (!strcmp(key, "variant")) {
        if (value < -1 || value > 5) {
            set_error("variant must be -1 (auto) or 0..5");
            return STORM_HIP_EINVAL;
        }
#ifndef STORM_HIP_PROBES
        if (value == 5) {
            set_error("variant 5 (wide 32x32x64 strips) is a form of the tools build (`make probes`), not of the shipped library");
            return STORM_HIP_EINVAL;
        }
#endif
        ctx->variant = (int)value;
    } else if (!strcmp(key, "probe_bundle")) {
        if (value != -1 && value != 1 && value != 4) {
            set_error("probe_bundle must be -1 (auto: 1), 1 (one group of 128 rows per workgroup) or 4 (bundles of four)");
            return STORM_HIP_EINVAL;
        }
        ctx->probe_bundle = (int)value;
    } else if (!strcmp(key, "sparse_probe")) {
        if (value < -1 || value > 1) {
            set_error("sparse_probe must be -1 (auto), 0 (never) or 1 (every eligible column)");
            return STORM_HIP_EINVAL;
        }
        ctx->sparse_probe = (int)value;
    } else if (!strcmp(key, "result_mailbox")) {
        ctx->result_mailbox = value != 0;
    } else if (!strcmp(key, "sync_poll_us")) {
        if (value < 0 || value > 1000000) {
            set_error("sync_poll_us: 0 .. 1000000 microseconds");
            return STORM_HIP_EINVAL;
        }
        ctx->sync_poll_us = (int)value;
    } else if (!strcmp(key, "matrix_lists")) {
        if (value < -1 || value > 1) {
            set_error("matrix_lists must be -1 (by density), 0 (never) or 1 (whenever eligible)");
            return STORM_HIP_EINVAL;
        }
        ctx->matrix_lists = (int)value;
    } else if (!strcmp(key, "matrix_lists_kernel")) {
        if (value < 0 || value > 2) {
            set_error("matrix_lists_kernel: 0 (by the row length), 1 (window kernel) or 2 (hash kernel)");
            return STORM_HIP_EINVAL;
        }
        ctx->matrix_lists_kernel = (int)value;
    } else if (!strcmp(key, "matrix_lists_hash_min_log2")) {
        if (value < 3 || value > 7) {
            set_error("matrix_lists_hash_min_log2: 3 .. 7");
            return STORM_HIP_EINVAL;
        }
        ctx->matrix_lists_hash_min_log2 = (int)value;
    } else if (!strcmp(key, "matrix_lists_debug")) {
        ctx->matrix_lists_debug = (int)value;
    } else if (!strcmp(key, "matrix_lists_density")) {
        if (value < 0 || value > 10000) {
            set_error("matrix_lists_density: 0 .. 10000 (1/10000 of the dense replica's bits)");
            return STORM_HIP_EINVAL;
        }
        ctx->matrix_lists_permille_x10 = (int)value;
    } else if (!strcmp(key, "seg_rows")) {
        if (value < 1 || value > (1 << 20)) {
            set_error("seg_rows out of range");
            return STORM_HIP_EINVAL;
        }
        ctx->seg_rows = (int)value;
    } else if (!strcmp(key, "k2_stages_per_item")) {
        if (value < 1 || value > 65536) {
            set_error("k2_stages_per_item out of range");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_stages_per_item = (int)value;
    } else if (!strcmp(key, "k2_max_run")) {
        if (value < 0 || value > 4096) {
            set_error("k2_max_run out of range (0 = chosen by the list-scheduling estimate, 1..4096)");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_max_run = (int)value;
    } else if (!strcmp(key, "k2_ring")) {
#ifdef STORM_HIP_PROBES
        if ((value < 3 || value > 5) && (value < 11 || value > 18) && value != 26) {
            set_error("k2_ring must be 3, 4 or 5 (10 + bits = timing probes)");
            return STORM_HIP_EINVAL;
        }
#else
        if (value != 4) {
            set_error("k2_ring: this build ships the 4-deep ring only (other depths and the timing "
                      "probes: build with STORM_HIP_PROBES, `make probes`)");
            return STORM_HIP_EINVAL;
        }
#endif
        ctx->k2_ring = (int)value;
    } else if (!strcmp(key, "k2_shadow_budget_mb")) {
        if (value < 0 || value > (1 << 22)) {
            set_error("k2_shadow_budget_mb must be 0 (unbounded) .. 4194304");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_shadow_budget_mb = (int)value;
        memset(ctx->x4_key, 0, sizeof(ctx->x4_key));
    } else if (!strcmp(key, "k2_tile_shape")) {
        if (value != 0 && value != 1 && value != 2 && value != 3 && value != 4 && value != 5 && value != 6 && value != 16 && value != 32) {
            set_error("k2_tile_shape must be 0 (chosen by the matrix), 5 (both operands as FP4 images in the LDS, 16x16x128 MFMAs), 1, 2 (bit operands inflated in registers), 3 / 4 (B as FP4 images in the LDS, 16x16x128 / 32x32x64 MFMAs), 6 (128 x 128 tiles, k-parts whose sums meet inside the launch), 16 or 32 (FP4 shadow)");
            return STORM_HIP_EINVAL;
        }
#ifndef STORM_HIP_PROBES
        if (value == 1 || value == 16) {
            set_error("k2_tile_shape = 1 / 16 (tilebits_kernel, tile16_fp4_kernel) is a form of the tools build (`make probes`), not of the shipped library");
            return STORM_HIP_EINVAL;
        }
#endif
        ctx->k2_tile_shape = (int)value;
    } else if (!strcmp(key, "k2_ring_sync")) {
        ctx->k2_ring_sync = value != 0;
    } else if (!strcmp(key, "k2_wave_below")) {
        if (value < 0 || value > (1 << 30)) {
            set_error("k2_wave_below: 0 (never) .. 2^30 tiles of 256 x 256");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_wave_below = (int)value;
    } else if (!strcmp(key, "k2_part_slots")) {
        if (value < 0 || value > 2) {
            set_error("k2_part_slots must be 0 (by the length of the segments), 1 or 2 per CU");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_part_slots = (int)value;
    } else if (!strcmp(key, "k2_part_min_chunks")) {
        if (value < 1 || value > 4096) {
            set_error("k2_part_min_chunks: 1 .. 4096 chunks of 512 bits");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_part_min_chunks = (int)value;
    } else if (!strcmp(key, "k2_part_narrow")) {
        ctx->k2_part_narrow = value != 0;
    } else if (!strcmp(key, "k2_part_cost_diag")) {
        if (value < 10 || value > 100) {
            set_error("k2_part_cost_diag: 10 .. 100 percent of a full tile's chunk");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_part_cost_diag = (int)value;
    } else if (!strcmp(key, "k2_ring_cost_diag") || !strcmp(key, "k2_ring_cost_ragged")) {
        if (value < 5 || value > 100) {
            set_error("%s is a percentage of a full tile's time, 5..100", key);
            return STORM_HIP_EINVAL;
        }
        (key[13] == 'd' ? ctx->k2_ring_cost_diag : ctx->k2_ring_cost_ragged) = (int)value;
    } else if (!strcmp(key, "k2_tile_cost_diag") || !strcmp(key, "k2_tile_cost_ragged")) {
        if (value < 5 || value > 100) {
            set_error("%s is a percentage of a full tile's time, 5..100", key);
            return STORM_HIP_EINVAL;
        }
        (key[13] == 'd' ? ctx->k2_tile_cost_diag : ctx->k2_tile_cost_ragged) = (int)value;
    } else if (!strcmp(key, "k2_strip_operands")) {
        if (value < 0 || value > 6) {
            set_error("k2_strip_operands must be 0 (by size), 1 (bit operands, one item per workgroup), 2 (bit operands, one stream per workgroup), 3 (the same with a ring per wave), 4 (FP4 shadow), 5 (bit operands, FP4 image built in the LDS) or 6 (the same with 512-row A tiles, two halves behind one image)");
            return STORM_HIP_EINVAL;
        }
#ifndef STORM_HIP_PROBES
        if (value == 1 || value == 3) {
            set_error("k2_strip_operands = 1 (stripbits_kernel) and 3 (bitwave_kernel) are forms of the tools build (`make probes`), not of the shipped library");
            return STORM_HIP_EINVAL;
        }
#endif
        ctx->k2_strip_operands = (int)value;
    } else if (!strcmp(key, "k2_shard_pairs")) {
        ctx->k2_shard_pairs = value != 0;
    } else if (!strcmp(key, "k2_matrix_pad")) {
        ctx->k2_matrix_pad = value < 0 ? -1 : (int)std::min<int64_t>(value, 64);   // chunks of 512 bytes
    } else if (!strcmp(key, "k2_fold_inline")) {
        ctx->k2_fold_inline = value < 0 ? -1 : value != 0;
    } else if (!strcmp(key, "k2_wave_ring")) {
        if (value != 0 && value != 3 && value != 4 && value != 6 && value != 8) {
            set_error("k2_wave_ring must be 0 (by occupancy), 3, 4, 6 or 8");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_wave_ring = (int)value;
    } else if (!strcmp(key, "k2_stream_max_rows")) {
        if (value < 0 || value > (1ll << 31)) {
            set_error("k2_stream_max_rows must be 0 .. 2^31");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_stream_max_rows = (int)value;
    } else if (!strcmp(key, "k2_stream_groups_per_cu")) {
        if (value < 0 || value > 255) {
            set_error("k2_stream_groups_per_cu must be 0 (auto) .. 255");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_stream_groups_per_cu = (int)value;
    } else if (!strcmp(key, "k2_stream_min_piece") || !strcmp(key, "k2_stream_min_run")) {
        if (value < 1 || value > 4096) {
            set_error("%s must be 1..4096 stages", key);
            return STORM_HIP_EINVAL;
        }
        (key[14] == 'p' ? ctx->k2_stream_min_piece : ctx->k2_stream_min_run) = (int)value;
    } else if (!strcmp(key, "k2_stream_w3_1") || !strcmp(key, "k2_stream_w3_2")) {
        if (value < 10 || value > 1000) {
            set_error("%s must be 10..1000 percent of the first workgroup's share", key);
            return STORM_HIP_EINVAL;
        }
        (key[13] == '1' ? ctx->k2_stream_w3_1 : ctx->k2_stream_w3_2) = (int)value;
    } else if (!strcmp(key, "k2_shape")) {
        if (value != 16 && value != 32) {
            set_error("k2_shape must be 16 (16x16x128 MFMA) or 32 (32x32x64)");
            return STORM_HIP_EINVAL;
        }
#ifndef STORM_HIP_PROBES
        if (value == 32) {
            set_error("k2_shape = 32 (strip_fp4_kernel) is a form of the tools build (`make probes`), not of the shipped library");
            return STORM_HIP_EINVAL;
        }
#endif
        ctx->k2_shape = (int)value;
    } else if (!strcmp(key, "keep_shadow")) {
        ctx->keep_shadow = value != 0;
        memset(ctx->x4_key, 0, sizeof(ctx->x4_key));
    } else if (!strcmp(key, "k2_matrix_parts")) {
        ctx->k2_matrix_parts = value != 0;
    } else if (!strcmp(key, "k2_matrix_min_part")) {
        if (value < 4 || value > 4096 || value % 4) {
            set_error("k2_matrix_min_part: a multiple of 4 in 4 .. 4096 (stages of 128 bits)");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_matrix_min_part = (int)value;
    } else if (!strcmp(key, "k2_matrix_split")) {
        ctx->k2_matrix_split = value != 0;
    } else if (!strcmp(key, "k2_pitch_pad")) {
        if (value < -1 || value > 65536 || (value > 0 && value % 128 != 0)) {
            set_error("k2_pitch_pad must be -1 (auto) or a multiple of 128 in 0..65536");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_pitch_pad = (int)value;
    } else if (!strcmp(key, "k2_lds_pad")) {
        if (value < 0 || value > 120 * 1024) {
            set_error("k2_lds_pad must be 0..122880");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_lds_pad = (int)value;
    } else if (!strcmp(key, "k2_persistent")) {
#ifndef STORM_HIP_PROBES
        if (value != 0) {
            set_error("k2_persistent (strip_fp4_kernel with work queues) is a form of the tools build (`make probes`), not of the shipped library");
            return STORM_HIP_EINVAL;
        }
#endif
        ctx->k2_persistent = value != 0;
    } else if (!strcmp(key, "k2_lpt_rounds")) {
        if (value < 0 || value > 63) {
            set_error("k2_lpt_rounds must be 0..63");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_lpt_rounds = (int)value;
    } else if (!strcmp(key, "k2_tail_slices")) {
        if (value < 0 || value > 255) {
            set_error("k2_tail_slices must be 0..255");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_tail_slices = (int)value;
    } else if (!strcmp(key, "k2_tail_run")) {
        if (value < 1 || value > 4096) {
            set_error("k2_tail_run must be 1..4096");
            return STORM_HIP_EINVAL;
        }
        ctx->k2_tail_run = (int)value;
    } else if (!strcmp(key, "k2_debug")) {
#ifdef STORM_HIP_PROBES
        ctx->k2_debug = (int)value;
#else
        if (value != 0) {
            set_error("k2_debug: timing probes (wrong results by design) are not in this build "
                      "(STORM_HIP_PROBES, `make probes`)");
            return STORM_HIP_EINVAL;
        }
#endif
    } else if (!strcmp(key, "time_kernels")) {
        // 1: start a new series of bracketed launches; 0: pause (the series is kept for
        // storm_hip_kernel_time); 2: resume it (bench.py brackets every 4th step at N > 1)
        ctx->time_kernels = value != 0;
        if (value == 1) ctx->kernel_events_used = 0;
    } else if (!strcmp(key, "chunks_per_item")) {
        if (value < 0 || value > 4096) {
            set_error("chunks_per_item out of range");
            return STORM_HIP_EINVAL;
        }
        ctx->chunks_per_item = (int)value;
    } else {
        set_error("unknown option '%s'", key);
        return STORM_HIP_EINVAL;
    }
    return STORM_HIP_OK;
}

int64_t storm_hip_ctx_get_option(storm_hip_ctx_t* ctx, const char* key) {
    if (check_ctx(ctx) || !key) return -1;
    if (!strcmp(key, "variant")) return ctx->variant;
    if (!strcmp(key, "variant_used")) return ctx->variant_used;
    if (!strcmp(key, "seg_rows")) return ctx->seg_rows;
    if (!strcmp(key, "result_mailbox")) return ctx->result_mailbox;
    if (!strcmp(key, "probe_bundle")) return ctx->probe_bundle;
    if (!strcmp(key, "sync_poll_us")) return ctx->sync_poll_us;
    if (!strcmp(key, "matrix_lists")) return ctx->matrix_lists;
    if (!strcmp(key, "matrix_lists_density")) return ctx->matrix_lists_permille_x10;
    if (!strcmp(key, "chunks_per_item")) return ctx->chunks_per_item;
    if (!strcmp(key, "k2_stages_per_item")) return ctx->k2_stages_per_item;
    if (!strcmp(key, "k2_max_run")) return ctx->k2_max_run;
    if (!strcmp(key, "k2_shape")) return ctx->k2_shape;
    if (!strcmp(key, "k2_strip_operands")) return ctx->k2_strip_operands;
    if (!strcmp(key, "k2_operands_used")) return ctx->k2_operands_used;
    if (!strcmp(key, "k2_stream_max_rows")) return ctx->k2_stream_max_rows;
    if (!strcmp(key, "k2_fold_inline")) return ctx->k2_fold_inline;
    if (!strcmp(key, "k2_matrix_pad")) return ctx->k2_matrix_pad;
    if (!strcmp(key, "k2_shard_pairs")) return ctx->k2_shard_pairs;
    if (!strcmp(key, "k2_ring_sync")) return ctx->k2_ring_sync;
    if (!strcmp(key, "k2_tile_shape")) return ctx->k2_tile_shape;
    if (!strcmp(key, "k2_wave_below")) return ctx->k2_wave_below;
    if (!strcmp(key, "k2_part_slots")) return ctx->k2_part_slots;
    if (!strcmp(key, "k2_part_min_chunks")) return ctx->k2_part_min_chunks;
    if (!strcmp(key, "k2_part_cost_diag")) return ctx->k2_part_cost_diag;
    if (!strcmp(key, "k2_part_narrow")) return ctx->k2_part_narrow;
    if (!strcmp(key, "k2_tile_shape_used")) return ctx->k2_tile_shape_eff;
    if (!strcmp(key, "k2_stream_w3_1")) return ctx->k2_stream_w3_1;
    if (!strcmp(key, "k2_stream_w3_2")) return ctx->k2_stream_w3_2;
    if (!strcmp(key, "k2_shadow_budget_mb")) return ctx->k2_shadow_budget_mb;
    if (!strcmp(key, "n_cus")) return ctx->n_cus;
    if (!strcmp(key, "probes_build")) {
#ifdef STORM_HIP_PROBES
        return 1;
#else
        return 0;
#endif
    }
#ifdef STORM_HIP_PROBES
    if (!strcmp(key, "probes_built")) return 1;
#else
    if (!strcmp(key, "probes_built")) return 0;
#endif
    return -1;
}

int storm_hip_kernel_time(storm_hip_ctx_t* ctx, double* sum_ms, uint64_t* launches) {
    if (check_ctx(ctx) || !sum_ms || !launches) return STORM_HIP_EINVAL;
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *sum_ms = 0.0;
    *launches = 0;
    for (size_t i = 0; i + 1 < ctx->kernel_events_used; i += 2) {
        float ms = 0.f;
        STORM_HIP_TRY(hipEventElapsedTime(&ms, ctx->kernel_events[i], ctx->kernel_events[i + 1]));
        *sum_ms += ms;
        ++*launches;
    }
    ctx->kernel_events_used = 0;
    return STORM_HIP_OK;
}

int storm_hip_debug_strip_trace(storm_hip_ctx_t* ctx, uint64_t* out, uint64_t capacity_items,
                                uint64_t* n_items) {
    return guarded("storm_hip_debug_strip_trace", [&]() -> int {
    if (check_ctx(ctx) || !n_items) return STORM_HIP_EINVAL;
#ifndef STORM_HIP_PROBES
    set_error("strip trace: not in this build (STORM_HIP_PROBES, `make probes`)");
    *n_items = 0;
    return STORM_HIP_EINVAL;
#endif
    *n_items = ctx->trace_items;
    if (!out || ctx->trace_items == 0) return STORM_HIP_OK;
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const uint64_t n = std::min<uint64_t>(capacity_items, ctx->trace_items);
    if (ctx->trace_is_stream) {  // bitstream_kernel: 8 words per workgroup, as they are
        STORM_HIP_TRY(hipMemcpy(out, ctx->d_trace, n * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
        return STORM_HIP_OK;
    }
    // per item: 4 trace words, then the item record {a_row0, diag, j0, j1} widened to 64 bit
    std::vector<uint64_t> raw(n * 4);
    std::vector<uint32_t> items(n * 5, 0);
    STORM_HIP_TRY(hipMemcpy(raw.data(), ctx->d_trace, n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (!ctx->trace_is_stream)  // (bitstream_kernel traces workgroups, which have no item record)
        STORM_HIP_TRY(hipMemcpy(items.data(), ctx->d_strip_items, n * 5 * sizeof(uint32_t),
                                hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; ++i) {
        for (int k = 0; k < 4; ++k) out[i * 8 + k] = raw[i * 4 + k];
        out[i * 8 + 4] = items[i * 5 + 0];
        out[i * 8 + 5] = items[i * 5 + 1];
        out[i * 8 + 6] = (uint64_t)items[i * 5 + 3] - items[i * 5 + 2];
        out[i * 8 + 7] = items[i * 5 + 4];
    }
    return STORM_HIP_OK;
    });
}

int storm_hip_last_pass_report(storm_hip_ctx_t* ctx, uint64_t out[4]) {
    if (check_ctx(ctx) || !out) return STORM_HIP_EINVAL;
    memcpy(out, ctx->pass_report, sizeof(ctx->pass_report));
    return STORM_HIP_OK;
}

int storm_hip_last_launch_info(storm_hip_ctx_t* ctx, uint64_t out[4]) {
    if (check_ctx(ctx) || !out) return STORM_HIP_EINVAL;
    memcpy(out, ctx->last_info, sizeof(ctx->last_info));
    return STORM_HIP_OK;
}

// ---- dense matrix ----
int storm_hip_matrix_create(storm_hip_ctx_t* ctx, uint64_t n_rows, uint32_t n_words,
                            storm_hip_matrix_t** out) {
    return guarded("storm_hip_matrix_create", [&]() -> int {
    if (check_ctx(ctx) || !out) return STORM_HIP_EINVAL;
    *out = nullptr;
    if (n_words == 0 || n_rows >= (1ull << 32) - kABlockRows) {
        set_error("matrix shape out of range (rows=%llu words=%u)", (unsigned long long)n_rows,
                  n_words);
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    storm_hip_matrix_t* m = new (std::nothrow) storm_hip_matrix_t();
    if (!m) return STORM_HIP_ENOMEM;
    m->n_rows = n_rows;
    m->n_words = n_words;
    m->generation = next_matrix_generation();
    m->n_rows_pad = std::max<uint64_t>(kRowPad, (n_rows + kRowPad - 1) / kRowPad * kRowPad);
    m->stride_words = ((uint64_t)n_words + kChunkWords - 1) / kChunkWords * kChunkWords;
    m->stride_words += pitch_pad_chunks(ctx->k2_matrix_pad, m->stride_words) * kChunkWords;   // (why: storm_hip_internal.h)
    const size_t bytes = m->n_rows_pad * m->stride_words * sizeof(uint64_t);
    if (hipMalloc(reinterpret_cast<void**>(&m->d), bytes) != hipSuccess) {
        set_error("hipMalloc of %zu bytes for the dense matrix failed", bytes);
        delete m;
        return STORM_HIP_ENOMEM;
    }
    if (hipMemsetAsync(m->d, 0, bytes, ctx->stream) != hipSuccess) {
        set_error("hipMemsetAsync failed");
        (void)hipFree(m->d);
        delete m;
        return STORM_HIP_EHIP;
    }
    *out = m;
    return STORM_HIP_OK;
    });
}

// Changes the logical row count. Growing beyond the allocation reallocates (at least doubling, so
// that a container streamed in batch by batch copies O(n) rows in all) and carries the rows over
// device to device; rows [old, new) are zero until uploaded; shrinking zeroes the dropped rows, so
// "rows >= n_rows are zero" — what the kernels' padding relies on — always holds.
int storm_hip_matrix_resize(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m, uint64_t n_rows) {
    return guarded("storm_hip_matrix_resize", [&]() -> int {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!m || n_rows >= (1ull << 32) - kABlockRows) {
        set_error("matrix_resize: NULL matrix or row count out of range");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t row_bytes = m->stride_words * sizeof(uint64_t);
    if (n_rows > m->n_rows_pad) {
        const uint64_t want = (n_rows + kRowPad - 1) / kRowPad * kRowPad;
        const uint64_t new_pad = std::max<uint64_t>(want, 2 * m->n_rows_pad);
        uint64_t* nd = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&nd), new_pad * row_bytes) != hipSuccess) {
            set_error("matrix_resize: hipMalloc of %llu bytes failed",
                      (unsigned long long)(new_pad * row_bytes));
            return STORM_HIP_ENOMEM;
        }
        if (hipMemcpyAsync(nd, m->d, m->n_rows_pad * row_bytes, hipMemcpyDeviceToDevice,
                           ctx->stream) != hipSuccess ||
            hipMemsetAsync(nd + m->n_rows_pad * m->stride_words, 0,
                           (new_pad - m->n_rows_pad) * row_bytes, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            (void)hipFree(nd);
            set_error("matrix_resize: device copy failed");
            return STORM_HIP_EHIP;
        }
        (void)hipFree(m->d);
        m->d = nd;
        m->n_rows_pad = new_pad;
    } else if (n_rows < m->n_rows) {
        STORM_HIP_TRY(hipMemsetAsync(m->d + n_rows * m->stride_words, 0,
                                     (m->n_rows - n_rows) * row_bytes, ctx->stream));
    }
    m->n_rows = n_rows;
    m->generation = next_matrix_generation();
    return STORM_HIP_OK;
    });
}

static int check_rows(const storm_hip_matrix_t* m, uint64_t row0, uint64_t n_rows) {
    if (!m) {
        set_error("NULL matrix");
        return STORM_HIP_EINVAL;
    }
    if (row0 + n_rows > m->n_rows || row0 + n_rows < row0) {
        set_error("rows [%llu, +%llu) outside the matrix (%llu rows)", (unsigned long long)row0,
                  (unsigned long long)n_rows, (unsigned long long)m->n_rows);
        return STORM_HIP_EINVAL;
    }
    return STORM_HIP_OK;
}

int storm_hip_matrix_upload(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m, uint64_t row0,
                            uint64_t n_rows, const uint64_t* host_rows,
                            uint64_t src_stride_words) {
    if (m) m->generation = next_matrix_generation();  // any cached FP4 shadow is stale now
    if (check_ctx(ctx) || check_rows(m, row0, n_rows)) return STORM_HIP_EINVAL;
    if (n_rows == 0) return STORM_HIP_OK;
    if (!host_rows || src_stride_words < m->n_words) {
        set_error("upload: bad source");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    STORM_HIP_TRY(hipMemcpy2DAsync(m->d + row0 * m->stride_words, m->stride_words * 8, host_rows,
                                   src_stride_words * 8, (size_t)m->n_words * 8, n_rows,
                                   hipMemcpyHostToDevice, ctx->stream));
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STORM_HIP_OK;
}

int storm_hip_matrix_import(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m, uint64_t row0,
                            uint64_t n_rows, const void* device_rows,
                            uint64_t src_stride_words) {
    if (m) m->generation = next_matrix_generation();  // any cached FP4 shadow is stale now
    if (check_ctx(ctx) || check_rows(m, row0, n_rows)) return STORM_HIP_EINVAL;
    if (n_rows == 0) return STORM_HIP_OK;
    if (!device_rows || src_stride_words < m->n_words) {
        set_error("import: bad source");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    STORM_HIP_TRY(hipMemcpy2DAsync(m->d + row0 * m->stride_words, m->stride_words * 8,
                                   device_rows, src_stride_words * 8, (size_t)m->n_words * 8,
                                   n_rows, hipMemcpyDeviceToDevice, ctx->stream));
    return STORM_HIP_OK;
}

int storm_hip_matrix_download(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, uint64_t row0,
                              uint64_t n_rows, uint64_t* host_rows, uint64_t dst_stride_words) {
    if (check_ctx(ctx) || check_rows(m, row0, n_rows)) return STORM_HIP_EINVAL;
    if (n_rows == 0) return STORM_HIP_OK;
    if (!host_rows || dst_stride_words < m->n_words) {
        set_error("download: bad destination");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    STORM_HIP_TRY(hipMemcpy2DAsync(host_rows, dst_stride_words * 8,
                                   m->d + row0 * m->stride_words, m->stride_words * 8,
                                   (size_t)m->n_words * 8, n_rows, hipMemcpyDeviceToHost,
                                   ctx->stream));
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STORM_HIP_OK;
}

int storm_hip_matrix_set_rows_from_positions(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m,
                                             uint64_t row0, uint64_t n_rows,
                                             const uint64_t* offsets,
                                             const uint32_t* positions) {
    return guarded("storm_hip_matrix_set_rows_from_positions", [&]() -> int {
    if (m) m->generation = next_matrix_generation();  // any cached FP4 shadow is stale now
    if (check_ctx(ctx) || check_rows(m, row0, n_rows)) return STORM_HIP_EINVAL;
    if (n_rows == 0) return STORM_HIP_OK;
    if (!offsets || (!positions && offsets[n_rows] != offsets[0])) {
        set_error("set_rows_from_positions: NULL input");
        return STORM_HIP_EINVAL;
    }
    const uint64_t n_pos = offsets[n_rows] - offsets[0];
    const uint64_t n_bits = (uint64_t)m->n_words * 64;
    uint32_t largest = 0;  // (a max reduction vectorises; an early exit per position does not)
    for (uint64_t p = 0; p < n_pos; ++p) largest = std::max(largest, positions[offsets[0] + p]);
    if (n_pos && largest >= n_bits) {
        set_error("position %u outside the %llu-bit rows", largest, (unsigned long long)n_bits);
        return STORM_HIP_EINVAL;
    }
    for (uint64_t r = 0; r < n_rows; ++r)
        if (offsets[r + 1] < offsets[r]) {
            set_error("set_rows_from_positions: offsets decrease at row %llu", (unsigned long long)r);
            return STORM_HIP_EINVAL;
        }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    if (n_pos == 0) return STORM_HIP_OK;
    // staging buffer of the context (grow-only: STORM_contig_add streams batches of 256 rows through here, and
    // a hipMalloc / hipFree pair per batch costs more than its copy)
    const size_t off_bytes = ((n_rows + 1) * sizeof(uint64_t) + 255) / 256 * 256;
    const size_t need = off_bytes + n_pos * sizeof(uint32_t);
    if (need > ctx->positions_capacity) {
        if (ctx->d_positions) STORM_HIP_TRY(hipFree(ctx->d_positions));
        ctx->d_positions = nullptr;
        ctx->positions_capacity = 0;
        const size_t cap = std::max<size_t>(need + need / 2, 1u << 20);
        if (hipMalloc(&ctx->d_positions, cap) != hipSuccess) {
            set_error("hipMalloc for %llu positions failed", (unsigned long long)n_pos);
            return STORM_HIP_ENOMEM;
        }
        ctx->positions_capacity = cap;
    }
    uint64_t* d_off = static_cast<uint64_t*>(ctx->d_positions);
    uint32_t* d_pos = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(ctx->d_positions) + off_bytes);
    int rc = STORM_HIP_OK;
    if (hipMemcpyAsync(d_off, offsets, (n_rows + 1) * sizeof(uint64_t), hipMemcpyHostToDevice,
                       ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d_pos, positions + offsets[0], n_pos * sizeof(uint32_t),
                       hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        set_error("position upload failed");
        rc = STORM_HIP_EHIP;
    } else {
        hipLaunchKernelGGL(set_bits_kernel, dim3((uint32_t)n_rows), dim3(kThreads), 0,
                           ctx->stream, m->d, m->stride_words, row0, d_off, d_pos);
        // (the host buffers are pageable and the caller's: wait for the copies)
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            set_error("set_bits_kernel failed");
            rc = STORM_HIP_EHIP;
        }
    }
    return rc;
    });
}

int storm_hip_matrix_fill_synthetic(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m,
                                    uint64_t n_bits, uint32_t draws, uint64_t seed) {
    return guarded("storm_hip_matrix_fill_synthetic", [&]() -> int {
    if (m) m->generation = next_matrix_generation();  // any cached FP4 shadow is stale now
    if (check_ctx(ctx) || !m) return STORM_HIP_EINVAL;
    if (n_bits == 0 || (n_bits + 63) / 64 != m->n_words) {
        set_error("fill_synthetic: n_bits=%llu does not match %u words per row",
                  (unsigned long long)n_bits, m->n_words);
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    STORM_HIP_TRY(hipMemsetAsync(m->d, 0, m->n_rows_pad * m->stride_words * 8, ctx->stream));
    if (m->n_rows == 0 || draws == 0) return STORM_HIP_OK;
    const uint64_t total = m->n_rows * draws;
    const uint32_t grid =
        (uint32_t)std::min<uint64_t>((total + kThreads - 1) / kThreads, 256u * 64u);
    hipLaunchKernelGGL(synth_fill_kernel, dim3(grid), dim3(kThreads), 0, ctx->stream, m->d,
                       m->stride_words, m->n_rows, n_bits, draws, seed);
    STORM_HIP_TRY(hipGetLastError());
    return STORM_HIP_OK;
    });
}

int storm_hip_matrix_clear(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m) {
    if (m) m->generation = next_matrix_generation();  // any cached FP4 shadow is stale now
    if (check_ctx(ctx) || !m) return STORM_HIP_EINVAL;
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    STORM_HIP_TRY(hipMemsetAsync(m->d, 0, m->n_rows_pad * m->stride_words * 8, ctx->stream));
    return STORM_HIP_OK;
}

void storm_hip_matrix_destroy(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m) {
    if (!m) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (m->d) (void)hipFree(m->d);
    delete m;
}

uint64_t storm_hip_matrix_rows(const storm_hip_matrix_t* m) { return m ? m->n_rows : 0; }
uint32_t storm_hip_matrix_words(const storm_hip_matrix_t* m) { return m ? m->n_words : 0; }
uint64_t storm_hip_matrix_stride_words(const storm_hip_matrix_t* m) {
    return m ? m->stride_words : 0;
}
void* storm_hip_matrix_device_ptr(const storm_hip_matrix_t* m) { return m ? m->d : nullptr; }

// ---- hot path ----
}  // extern "C"

namespace storm {

uint64_t next_matrix_generation() {
    static std::atomic<uint64_t> counter{1};
    return counter.fetch_add(1);
}

void kernel_time_mark(storm_hip_ctx_t* ctx) {
    if (!ctx->time_kernels) return;
    if (ctx->kernel_events_used == ctx->kernel_events.size()) {
        hipEvent_t ev = nullptr;
        // (no system-scope release with the record: only the timestamps are wanted, and the default
        //  flavour's cache write-back costs the stream ~4 us per record)
        if (hipEventCreateWithFlags(&ev, hipEventDisableSystemFence) != hipSuccess) return;
        ctx->kernel_events.push_back(ev);
    }
    (void)hipEventRecord(ctx->kernel_events[ctx->kernel_events_used++], ctx->stream);
}

int launch_fold_slots(storm_hip_ctx_t* ctx, uint64_t* d_total) {
    hipLaunchKernelGGL(fold_slots_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_slots,
                       reinterpret_cast<unsigned long long*>(d_total));
    STORM_HIP_TRY(hipGetLastError());
    return STORM_HIP_OK;
}

int launch_pairw_segments(storm_hip_ctx_t* ctx, const uint64_t* X, uint64_t stride_words,
                          const Seg* d_segs, uint32_t n_segs, uint64_t seg_row_sum,
                          uint64_t* d_total) {
    const uint32_t n_chunks = (uint32_t)(stride_words / kChunkWords);
    uint32_t cps = (uint32_t)ctx->chunks_per_item;
    if (cps == 0) {
        // enough items to keep 256 CUs x 4 workgroups balanced; fewer, longer items once
        // there are plenty (each item re-loads its A block per chunk either way)
        const uint64_t items_at_1 = (uint64_t)n_segs * n_chunks;
        cps = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(16, items_at_1 / 65536));
    }
    // a lane adds at most 32 rows x seg_rows x 64 bits per chunk into its uint32 accumulators
    const uint64_t per_chunk = (uint64_t)kRowsPerWave * (uint64_t)ctx->seg_rows * 64u;
    if (per_chunk >= (1ull << 32)) {
        set_error("seg_rows=%d overflows the 32-bit lane accumulators", ctx->seg_rows);
        return STORM_HIP_EINVAL;
    }
    while (cps > 1 && per_chunk * cps >= (1ull << 32)) --cps;
    cps = std::max(1u, std::min(cps, n_chunks));
    const uint32_t n_kslices = (n_chunks + cps - 1) / cps;
    const uint64_t n_items = (uint64_t)n_segs * n_kslices;
    if (n_items >= (1ull << 31)) {
        set_error("work decomposition has %llu items (> 2^31): raise chunks_per_item",
                  (unsigned long long)n_items);
        return STORM_HIP_EINVAL;
    }
    ctx->last_info[0] = n_items;
    ctx->last_info[1] = cps;
    ctx->last_info[2] = (uint64_t)kABlockRows * seg_row_sum * stride_words;
    ctx->last_info[3] = n_segs;

    if (n_items > 0) {
        const dim3 grid((uint32_t)n_items), block(kThreads);
        kernel_time_mark(ctx);
        switch (ctx->variant < 0 ? 2 : ctx->variant) {
            case 0:
                hipLaunchKernelGGL(pairw_dense_kernel<0>, grid, block, 0, ctx->stream, X,
                                   stride_words, d_segs, n_segs, n_chunks, cps, ctx->d_slots);
                break;
            case 1:
                hipLaunchKernelGGL(pairw_dense_kernel<1>, grid, block, 0, ctx->stream, X,
                                   stride_words, d_segs, n_segs, n_chunks, cps, ctx->d_slots);
                break;
            default:
                hipLaunchKernelGGL(pairw_dense_kernel<2>, grid, block, 0, ctx->stream, X,
                                   stride_words, d_segs, n_segs, n_chunks, cps, ctx->d_slots);
                break;
        }
        kernel_time_mark(ctx);
        STORM_HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(fold_slots_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_slots,
                       reinterpret_cast<unsigned long long*>(d_total));
    STORM_HIP_TRY(hipGetLastError());
    return STORM_HIP_OK;
}

// The context's result word to the host, through a pinned word (a pageable destination makes the runtime stage
// the 8 bytes and costs a call ~10 us more; with G devices driven from one process that is G times).
uint64_t* result_target(storm_hip_ctx_t* ctx) {
    ctx->mail_armed = false;
    if (!ctx->result_mailbox) return reinterpret_cast<uint64_t*>(ctx->d_scalar);
    if (!ctx->h_mail) {
        // (on the context's own device: callers evaluate this as an ARGUMENT, before the entry point they call has set the
        //  device — a worker thread of slot d starts out on device 0; portable: one mapping for every device)
        void* dev = nullptr;
        if (hipSetDevice(ctx->device) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&ctx->h_mail), 64,
                          hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess ||
            hipHostGetDevicePointer(&dev, ctx->h_mail, 0) != hipSuccess) {
            if (ctx->h_mail) (void)hipHostFree(ctx->h_mail);
            ctx->h_mail = nullptr;
            ctx->result_mailbox = 0;   // (no mapped host memory here: the copy + synchronize path)
            return reinterpret_cast<uint64_t*>(ctx->d_scalar);
        }
        ctx->d_mail = static_cast<unsigned long long*>(dev);
    }
    __atomic_store_n(ctx->h_mail, ~0ull, __ATOMIC_RELEASE);   // no total is ~0: that is the API's own failure value
    ctx->mail_armed = true;
    return reinterpret_cast<uint64_t*>(ctx->d_mail);
}

int wait_mailbox(storm_hip_ctx_t* ctx, uint64_t* value) {
    ctx->mail_armed = false;
    // the device's store arrives over the bus while the kernel ends: a few hundred polls; a pass of seconds (c5) falls
    // through to the synchronize after 2 ms of polling
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; ++spin) {
        const unsigned long long v = __atomic_load_n(ctx->h_mail, __ATOMIC_ACQUIRE);
        if (v != ~0ull) {
            *value = v;
            return STORM_HIP_OK;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if ((spin & 1023u) == 1023u &&
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 2.0)
            break;
    }
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const unsigned long long v = __atomic_load_n(ctx->h_mail, __ATOMIC_ACQUIRE);
    if (v == ~0ull) {
        set_error("the pass ended without writing its total (result mailbox)");
        return STORM_HIP_EHIP;
    }
    *value = v;
    return STORM_HIP_OK;
}

int fetch_result_word(storm_hip_ctx_t* ctx, uint64_t* h_total) {
    if (ctx->mail_armed) return wait_mailbox(ctx, h_total);
    if (!ctx->h_scalar &&
        hipHostMalloc(reinterpret_cast<void**>(&ctx->h_scalar), 64, hipHostMallocDefault) != hipSuccess)
        ctx->h_scalar = nullptr;
    unsigned long long* dst = ctx->h_scalar ? ctx->h_scalar : reinterpret_cast<unsigned long long*>(h_total);
    STORM_HIP_TRY(hipMemcpyAsync(dst, ctx->d_scalar, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->h_scalar) *h_total = *ctx->h_scalar;
    return STORM_HIP_OK;
}
}  // namespace storm

extern "C" {

int storm_hip_pairw_dense_launch(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m,
                                 uint32_t shard_rank, uint32_t shard_count, uint64_t* d_total) {
    return guarded("storm_hip_pairw_dense_launch", [&]() -> int {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!m || !d_total) {
        set_error("pairw_dense: NULL matrix or result pointer");
        return STORM_HIP_EINVAL;
    }
    if (shard_count == 0 || shard_rank >= shard_count) {
        set_error("pairw_dense: shard %u of %u is not valid", shard_rank, shard_count);
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    // variant -1 = auto: the matrix-core strips at every size (rows beyond their 32-bit DMA offsets
    // are multiplied k-chunk by k-chunk over a compact shadow). Measured (tools/archive/bench_crossover.py):
    // they beat the popcount kernel from N = 64 up (10-37 us vs its 42-84 us latency floor at
    // N <= 256).
    int variant = ctx->variant;
    if (variant < 0) variant = 4;
    ctx->variant_used = variant;
    memset(ctx->pass_report, 0, sizeof(ctx->pass_report));
    if (variant >= 3) {
        const int saved = ctx->variant;
        ctx->variant = variant;
        const int rc = launch_pairw_mfma(ctx, m, shard_rank, shard_count, d_total);
        ctx->variant = saved;
        return rc;
    }
    if (int rc = ensure_segments(ctx, m->n_rows, shard_rank, shard_count)) return rc;
    ctx->pass_report[0] |= STORM_HIP_RAN_POPCOUNT;
    ctx->pass_report[1] += m->n_rows * (m->n_rows - (m->n_rows != 0)) / 2 * m->n_words / shard_count;
    return launch_pairw_segments(ctx, m->d, m->stride_words, ctx->d_segs, ctx->n_segs,
                                 ctx->seg_row_sum, d_total);
    });
}

int storm_hip_pairw_dense_begin(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m,
                                uint32_t shard_rank, uint32_t shard_count) {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    return storm_hip_pairw_dense_launch(ctx, m, shard_rank, shard_count, result_target(ctx));
}

int storm_hip_pairw_dense_end(storm_hip_ctx_t* ctx, uint64_t* h_total) {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!h_total) {
        set_error("pairw_dense_end: NULL result pointer");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    return fetch_result_word(ctx, h_total);
}

int storm_hip_pairw_dense(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m,
                          uint32_t shard_rank, uint32_t shard_count, uint64_t* h_total) {
    if (int rc = storm_hip_pairw_dense_begin(ctx, m, shard_rank, shard_count)) return rc;
    return storm_hip_pairw_dense_end(ctx, h_total);
}

int storm_hip_pairw_dense_upload(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m, const uint64_t* host_rows,
                                 uint64_t src_stride_words, uint64_t* h_total) {
    return guarded("storm_hip_pairw_dense_upload", [&]() -> int {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!m || !host_rows || !h_total || src_stride_words < m->n_words) {
        set_error("pairw_dense_upload: bad argument");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    m->generation = next_matrix_generation();
    const bool strips = (ctx->variant < 0 || ctx->variant == 4) && strip_operands_of(ctx) == 5 && !ctx->k2_persistent &&
                        ctx->k2_debug == 0 && m->n_rows >= 2048 &&
                        (m->n_rows + 255) / 256 * 256 <= m->n_rows_pad && m->stride_words * 8 * 64ull < (1ull << 32);
    if (!strips) {   // small matrices, forced kernel forms: the copy, then the pass
        if (int rc = storm_hip_matrix_upload(ctx, m, 0, m->n_rows, host_rows, src_stride_words)) return rc;
        return storm_hip_pairw_dense(ctx, m, 0, 1, h_total);
    }
    memset(ctx->pass_report, 0, sizeof(ctx->pass_report));
    ctx->variant_used = 4;
    if (int rc = launch_pairw_bits_upload(ctx, m, host_rows, src_stride_words, result_target(ctx)))
        return rc;
    return fetch_result_word(ctx, h_total);
    });
}

int storm_hip_square_dense(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* a,
                           const storm_hip_matrix_t* b, uint64_t* h_total) {
    return guarded("storm_hip_square_dense", [&]() -> int {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!a || !b || !h_total) {
        set_error("square_dense: NULL argument");
        return STORM_HIP_EINVAL;
    }
    if (a->n_words != b->n_words) {
        set_error("square_dense: row widths differ (%u vs %u words)", a->n_words, b->n_words);
        return STORM_HIP_EINVAL;
    }
    *h_total = 0;
    if (a->n_rows == 0 || b->n_rows == 0) return STORM_HIP_OK;
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    // same choice as the all-pairs path: matrix cores whenever the strips can address the rows,
    // the popcount kernel otherwise (or when forced by `variant`)
    const bool big = a->stride_words * 32ull * 64ull < (1ull << 32);
    if (a->stride_words == b->stride_words && (ctx->variant >= 3 || (ctx->variant < 0 && big))) {
        if (int rc = launch_square_mfma(ctx, a, b, result_target(ctx)))
            return rc;
        ctx->variant_used = 4;
        return fetch_result_word(ctx, h_total);
    }
    ctx->variant_used = 2;
    const uint32_t seg_rows = (uint32_t)ctx->seg_rows;
    const uint32_t a_blocks = (uint32_t)(a->n_rows_pad / kABlockRows);
    const uint32_t spb = (uint32_t)((b->n_rows + seg_rows - 1) / seg_rows);
    const uint32_t n_segs = a_blocks * spb;
    const uint32_t n_chunks = (uint32_t)(a->stride_words / kChunkWords);
    const uint64_t n_items = (uint64_t)n_segs * n_chunks;
    if (n_items >= (1ull << 31)) {
        set_error("square_dense: too many work items");
        return STORM_HIP_EINVAL;
    }
    hipLaunchKernelGGL(square_dense_kernel, dim3((uint32_t)n_items), dim3(kThreads), 0,
                       ctx->stream, a->d, a->stride_words, b->d, b->stride_words,
                       (uint32_t)b->n_rows, seg_rows, spb, n_segs, n_chunks, ctx->d_slots);
    STORM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(fold_slots_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_slots,
                       reinterpret_cast<unsigned long long*>(result_target(ctx)));
    STORM_HIP_TRY(hipGetLastError());
    return fetch_result_word(ctx, h_total);
    });
}

int storm_hip_tile_counts(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, uint64_t i0,
                          uint64_t i1, uint64_t j0, uint64_t j1, uint32_t* h_out) {
    return guarded("storm_hip_tile_counts", [&]() -> int {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!m || !h_out || i1 < i0 || j1 < j0 || i1 > m->n_rows || j1 > m->n_rows) {
        set_error("tile_counts: bad tile");
        return STORM_HIP_EINVAL;
    }
    const uint64_t ni = i1 - i0, nj = j1 - j0, n = ni * nj;
    if (n == 0) return STORM_HIP_OK;
    if (n >= (1ull << 31) || ni >= (1ull << 31) || nj >= (1ull << 31)) {
        set_error("tile_counts: tile too large");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    uint32_t* d_out = nullptr;
    STORM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), n * sizeof(uint32_t)));
    hipLaunchKernelGGL(tile_counts_kernel, dim3((uint32_t)((n + kWaves - 1) / kWaves)),
                       dim3(kThreads), 0, ctx->stream, m->d, m->stride_words, m->n_words, i0, j0,
                       (uint32_t)ni, (uint32_t)nj, d_out);
    int rc = STORM_HIP_OK;
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(h_out, d_out, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream) !=
            hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        set_error("tile_counts kernel/copy failed");
        rc = STORM_HIP_EHIP;
    }
    (void)hipFree(d_out);
    return rc;
    });
}

}  // extern "C"

namespace storm {
int launch_row_counts(storm_hip_ctx_t* ctx, const storm_hip_matrix_s* m, uint32_t* d_counts) {
    if (m->n_rows == 0) return STORM_HIP_OK;
    hipLaunchKernelGGL(row_counts_kernel, dim3((uint32_t)((m->n_rows + kWaves - 1) / kWaves)),
                       dim3(kThreads), 0, ctx->stream, m->d, m->stride_words, m->n_rows,
                       m->n_words, d_counts);
    STORM_HIP_TRY(hipGetLastError());
    return STORM_HIP_OK;
}
}  // namespace storm

extern "C" {

int storm_hip_row_counts(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, uint32_t* h_counts) {
    return guarded("storm_hip_row_counts", [&]() -> int {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!m || !h_counts) {
        set_error("row_counts: NULL argument");
        return STORM_HIP_EINVAL;
    }
    if (m->n_rows == 0) return STORM_HIP_OK;
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    uint32_t* d_counts = nullptr;
    STORM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_counts), m->n_rows * sizeof(uint32_t)));
    int rc = launch_row_counts(ctx, m, d_counts);
    if (rc == STORM_HIP_OK &&
        (hipMemcpyAsync(h_counts, d_counts, m->n_rows * sizeof(uint32_t), hipMemcpyDeviceToHost,
                        ctx->stream) != hipSuccess ||
         hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        set_error("row_counts: HIP failure");
        rc = STORM_HIP_EHIP;
    }
    (void)hipFree(d_counts);
    return rc;
    });
}

int storm_hip_pairw_dense_op(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op,
                             uint64_t* h_total) {
    return guarded("storm_hip_pairw_dense_op", [&]() -> int {
    if (op < STORM_HIP_OP_AND || op > STORM_HIP_OP_XOR) {
        set_error("pairw_dense_op: unknown op %d", op);
        return STORM_HIP_EINVAL;
    }
    if (int rc = storm_hip_pairw_dense(ctx, m, 0, 1, h_total)) return rc;
    if (op == STORM_HIP_OP_AND || m->n_rows < 2) return STORM_HIP_OK;
    std::vector<uint32_t> counts(m->n_rows);
    if (int rc = storm_hip_row_counts(ctx, m, counts.data())) return rc;
    uint64_t set_bits = 0;
    for (uint32_t c : counts) set_bits += c;
    // every row meets N-1 partners: sum_{i<j} (n_i + n_j) = (N-1) * sum_i n_i
    const uint64_t both = (m->n_rows - 1) * set_bits;
    *h_total = both - (op == STORM_HIP_OP_XOR ? 2 * *h_total : *h_total);
    return STORM_HIP_OK;
    });
}

int storm_hip_square_matrix_device(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* a,
                                   const storm_hip_matrix_t* b, int op, uint32_t* d_out, uint64_t ld) {
    return guarded("storm_hip_square_matrix_device", [&]() -> int {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!a || !b || !d_out || ld < b->n_rows || op < STORM_HIP_OP_AND || op > STORM_HIP_OP_XOR ||
        a->n_words != b->n_words) {
        set_error("square_matrix: NULL argument, unknown op, row widths differ or ld < rows of B");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    return launch_square_matrix(ctx, a, b, op, d_out, ld);
    });
}

int storm_hip_square_matrix(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* a,
                            const storm_hip_matrix_t* b, int op, uint32_t* h_out) {
    return guarded("storm_hip_square_matrix", [&]() -> int {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!a || !b || !h_out) {
        set_error("square_matrix: NULL argument");
        return STORM_HIP_EINVAL;
    }
    const uint64_t n = a->n_rows * b->n_rows;
    if (n == 0) return STORM_HIP_OK;
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    uint32_t* d_out = nullptr;
    STORM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), n * sizeof(uint32_t)));
    int rc = storm_hip_square_matrix_device(ctx, a, b, op, d_out, b->n_rows);
    if (rc == STORM_HIP_OK &&
        (hipMemcpyAsync(h_out, d_out, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream) !=
             hipSuccess ||
         hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        set_error("square_matrix: HIP failure");
        rc = STORM_HIP_EHIP;
    }
    (void)hipFree(d_out);
    return rc;
    });
}

int storm_hip_pairw_matrix_device(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op,
                                  uint32_t* d_out, uint64_t ld) {
    return guarded("storm_hip_pairw_matrix_device", [&]() -> int {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!m || !d_out || ld < m->n_rows || op < STORM_HIP_OP_AND || op > STORM_HIP_OP_XOR) {
        set_error("pairw_matrix: NULL argument, unknown op or leading dimension < rows");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    return launch_pairw_matrix(ctx, m, op, d_out, ld);
    });
}

int storm_hip_pairw_matrix_band_device(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op,
                                       uint64_t row0, uint64_t n_band_rows, uint32_t* d_out,
                                       uint64_t ld) {
    return guarded("storm_hip_pairw_matrix_band_device", [&]() -> int {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!m || !d_out || ld < m->n_rows || op < STORM_HIP_OP_AND || op > STORM_HIP_OP_XOR ||
        row0 > m->n_rows || n_band_rows > m->n_rows - row0) {
        set_error("pairw_matrix_band: NULL argument, unknown op, band outside the matrix or ld < rows");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    return launch_pairw_matrix(ctx, m, op, d_out, ld, row0, n_band_rows);
    });
}

// Host-output band: the band is computed into a staging buffer of the context (kept between
// calls) and copied out row by row (ld may exceed n_rows). _begin only enqueues, so that one host
// thread can keep one band per GPU in flight; _end waits.
int storm_hip_pairw_matrix_band_begin(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op,
                                      uint64_t row0, uint64_t n_band_rows, uint32_t* h_out,
                                      uint64_t ld) {
    return guarded("storm_hip_pairw_matrix_band_begin", [&]() -> int {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!m || !h_out || op < STORM_HIP_OP_AND || op > STORM_HIP_OP_XOR || ld < m->n_rows ||
        row0 > m->n_rows || n_band_rows > m->n_rows - row0) {
        set_error("pairw_matrix_band: NULL argument, unknown op, band outside the matrix or ld < rows");
        return STORM_HIP_EINVAL;
    }
    const uint64_t n = m->n_rows;
    if (n == 0 || n_band_rows == 0) return STORM_HIP_OK;
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    const size_t need = (size_t)n_band_rows * n * sizeof(uint32_t);
    if (need > ctx->band_capacity) {
        if (ctx->d_band) STORM_HIP_TRY(hipFree(ctx->d_band));
        ctx->d_band = nullptr;
        ctx->band_capacity = 0;
        if (hipMalloc(reinterpret_cast<void**>(&ctx->d_band), need) != hipSuccess) {
            set_error("pairw_matrix_band: hipMalloc of %zu bytes for the output band failed", need);
            return STORM_HIP_ENOMEM;
        }
        ctx->band_capacity = need;
    }
    STORM_HIP_TRY(hipMemsetAsync(ctx->d_band, 0, need, ctx->stream));
    if (int rc = launch_pairw_matrix(ctx, m, op, ctx->d_band, n, row0, n_band_rows, false)) return rc;
    STORM_HIP_TRY(hipMemcpy2DAsync(h_out, ld * sizeof(uint32_t), ctx->d_band, n * sizeof(uint32_t),
                                   n * sizeof(uint32_t), n_band_rows, hipMemcpyDeviceToHost,
                                   ctx->stream));
    return STORM_HIP_OK;
    });
}

int storm_hip_pairw_matrix_band_end(storm_hip_ctx_t* ctx) {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STORM_HIP_OK;
}

int storm_hip_pairw_matrix_band(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op,
                                uint64_t row0, uint64_t n_band_rows, uint32_t* h_out, uint64_t ld) {
    if (int rc = storm_hip_pairw_matrix_band_begin(ctx, m, op, row0, n_band_rows, h_out, ld)) return rc;
    return storm_hip_pairw_matrix_band_end(ctx);
}

int storm_hip_pairw_matrix(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op,
                           uint32_t* h_out) {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!m || !h_out || op < STORM_HIP_OP_AND || op > STORM_HIP_OP_XOR) {
        set_error("pairw_matrix: NULL argument or unknown op");
        return STORM_HIP_EINVAL;
    }
    return storm_hip_pairw_matrix_band(ctx, m, op, 0, m->n_rows, h_out, m->n_rows);
}

int storm_hip_column_identity(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m,
                              uint64_t* h_total) {
    return guarded("storm_hip_column_identity", [&]() -> int {
    if (check_ctx(ctx)) return STORM_HIP_EINVAL;
    if (!m || !h_total) {
        set_error("column_identity: NULL argument");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(column_identity_kernel, dim3((m->n_words + kWaves - 1) / kWaves),
                       dim3(kThreads), 0, ctx->stream, m->d, m->stride_words, m->n_rows,
                       m->n_words, ctx->d_slots);
    STORM_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(fold_slots_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_slots,
                       reinterpret_cast<unsigned long long*>(result_target(ctx)));
    STORM_HIP_TRY(hipGetLastError());
    return fetch_result_word(ctx, h_total);
    });
}

}  // extern "C"
