// storm_hip_sparse.hip — device path of the STORM_t (sparse, Roaring-like) container.
//
// Replaces the per-row-pair machinery of the reference: block-id merge
// (STORM_intersect_vector32_unsafe, storm.c:75-106), the 4-way kind dispatch
// (STORM_bitmap_intersect_cardinality_func, storm.c:618-656) and the all-pairs loops around
// them (storm.c:877-961).
//
// MI355X formulation. Two rows can only intersect inside a common 65536-bit block column, so
// the pair space is regrouped BY BLOCK COLUMN instead of by row pair:
//   1. the flattened arena (rows -> blocks) is regrouped on the host into column-major order:
//      all blocks with id c, in row order, become consecutive "pool rows" of 1024 words;
//   2. kind dispatch happens once per block, on the device: bitmap-kind blocks are copied into
//      their pool row, list-kind blocks (sorted uint16 positions) are expanded into theirs by
//      a scatter kernel — after this every block pair, whatever its kinds, is an AND+popcount
//      of two 8 KiB pool rows, and absent blocks simply do not exist in the column;
//   3. the dense kernel (storm_hip.hip) runs over one segment table that covers the upper
//      triangle of every column; A blocks are clipped at the column edge (Seg::a_end).
// The block-id merge of the reference therefore costs nothing at pair time, and HBM holds
// 8 KiB per *present* block (config c4: <= 655 MB), not per (row, column) cell.
#include "storm_hip_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <exception>
#include <memory>
#include <thread>

using namespace storm;

// (an empty launch at context creation loads this file's code object ahead of the first real call: storm_hip_ctx_create)
namespace storm {
__global__ void warm_sparse_kernel() {}
void warm_sparse_code(hipStream_t stream) { hipLaunchKernelGGL(warm_sparse_kernel, dim3(1), dim3(64), 0, stream); }
}  // namespace storm


constexpr uint64_t kPoolPitchDefault = 1024;

struct storm_hip_sparse_s {
    uint64_t* d_pool = nullptr;      // pool rows: [pool_rows_ready + 512][pitch] words, 1024 of them used
    uint64_t pitch = kPoolPitchDefault;  // words per pool row (see build_arena: the row pitch and the memory channels)
    uint64_t n_pool_rows = 0;        // rows of the whole layout: columns the list-probe kernel cannot take first, ...
    uint64_t pool_rows_ready = 0;    // ... and only those exist until a dense pass over a probe column is asked for
    struct ProbeRegion { uint32_t e_begin, e_end, pool_row0, octant; };
    std::vector<ProbeRegion> probe_regions;  // (column, octant) element ranges: how ensure_full_pool expands the lists
    std::vector<RowRange> cols;      // pool-row range [r0, r1) of each non-empty column; every r0
                                     // is a multiple of 512 and the gap up to it is zero rows
    uint64_t census[4] = {0, 0, 0, 0};
    // list-probe path (K4): columns whose blocks are all short lists
    std::vector<uint64_t> col_list0; // per entry of `cols`: first pool row of the column's LIST blocks (its bitmap
                                     // blocks come first, the lists on the next multiple of 512 rows)
    std::vector<uint8_t> col_probe;  // per entry of `cols`: 1 = has probe data (its list blocks among themselves)
    std::vector<uint32_t> col_avg_len;  // per entry of `cols`: mean list length (probe columns)
    uint32_t* d_probe_elems = nullptr;  // (row in column) << 16 | position in block, column by column, row order
    uint16_t* d_probe_pos16 = nullptr;  // 2 x the positions alone (byte offsets into a count table), same index ranges, own order
    struct ProbeItemHost { uint32_t a_begin, a_end, n_begin, n_end, b_begin, b_end, a0, col; };
    std::vector<ProbeItemHost> probe_items;  // all eligible columns (family order); filtered per launch
    // [r6] the same work in bundles of kFatGroups groups (probe_lists_fat_kernel): atoms at[k] .. at[k + 1] of the groups
    // of the bundle, the far piece, first = the bundle's near parts are this item's
    struct ProbeFatItemHost { uint32_t at[5]; uint32_t b_begin, b_end, first, col; };
    std::vector<ProbeFatItemHost> probe_fat_items;
    int probe_bundle_launch = 1;   // which list the device holds (1: probe_items, 4: probe_fat_items)
    void* d_probe_items = nullptr;
    size_t probe_items_capacity = 0;
    uint64_t probe_key = ~0ull;
    uint32_t n_probe_launch = 0, n_probe_cols_launch = 0;
    uint64_t probe_lookups_launch = 0;  // positions the launched items stream + their own rows' elements (this shard)
    // segment table cache (per shard)
    Seg* d_segs = nullptr;
    uint32_t n_segs = 0;
    uint64_t seg_row_sum = 0;
    uint32_t seg_rank = 0, seg_count = 0, seg_len = 0;
};

namespace {

constexpr uint32_t kBlockWords = 1024;  // 65536 bits
constexpr uint32_t kMaxBlockId = 65536;  // uint32 positions / 65536 bits per block
constexpr uint32_t kSerialMagic = 0x314d5453u;  // "STM1": second header word of STORM_serialize

// list-kind block -> pool row: one workgroup per block
__global__ __launch_bounds__(kThreads) void expand_lists_kernel(
    uint64_t* __restrict__ pool, uint64_t pitch, const uint32_t* __restrict__ pool_row,
    const uint64_t* __restrict__ list_off, const uint32_t* __restrict__ list_len,
    const uint16_t* __restrict__ lists) {
    const uint32_t b = blockIdx.x;
    unsigned long long* row =
        reinterpret_cast<unsigned long long*>(pool + (uint64_t)pool_row[b] * pitch);
    const uint16_t* l = lists + list_off[b];
    for (uint32_t k = threadIdx.x; k < list_len[b]; k += kThreads) {
        const uint32_t v = l[k];
        atomicOr(&row[v >> 6], 1ull << (v & 63u));
    }
}

// ------------------------------------------------------------------------------------------
// K4 — list probe kernel (probe_lists_kernel): the list blocks of a block column among themselves (the reference's
// "extremely fast when sparse" regime: STORM_intersect_vector16_cardinality, storm.c:4-73, reached through the kind
// dispatch :618-656). The dense path multiplies 8 KiB per present block whatever its density; here the work is
// proportional to the listed positions.
//   data  : per column AND per OCTANT of the block (8192 positions) the listed positions of its rows, twice:
//           `elems`: uint32 (row in column) << 16 | position in octant, in row order (what a group's count
//           table is built from; also the source of ensure_full_pool);
//           `pos16`: the positions alone as byte offsets (2 x position, uint16) — the FAR STREAM: inside every atom
//           (the elements of one group of 128 rows in one octant) dealt by LDS bank (probe_deal_kernel);
//   item  : one group of kProbeRows = 128 consecutive rows of a column x one octant x a chunk of the far stream
//           behind the group (items begin and end on atoms; the chunk grid is the same for all groups of a stream,
//           the items that read one chunk — a family — run on one XCD);
//   LDS   : Cn[position] = how many of the group's rows list the position: a histogram of the group's elements,
//           8192 x 2 B = 16 KiB per workgroup;
//   work  : the group's first item adds C(Cn[p], 2) per position (the pairs inside the group); every item adds
//           Cn[p] per streamed far position p — ONE 2-byte LDS read and one add per lookup, the contribution of that
//           listed position against all 128 rows of the group at once. A lookup therefore stands for 128 of the
//           reference's per-pair list tests; the work still grows with N^2 / 128 x density (with the whole column as
//           one group this would be the column identity sum_c C(n_c, 2), which stays the verification path).
//   roofs : LDS 32 two-byte lookups per clock and CU; VALU ~2 instructions per lookup (unpack the position, add):
//           both ~2e13 lookups/s per chip. Measured: profiles/r03_b_sparse_probe_roofline.txt, r04_* .
// History (rounds 2 / 3, profiles/r03_b_sparse_probe_ab.txt): a table of 16 rows x 65536 positions; then the group
// as a transposed BITMAP (128-bit masks per position, 128 KiB, popcounts per lookup — v_bcnt_u32_b32 is half rate);
// the counts are what those popcounts add up to.
// ------------------------------------------------------------------------------------------
struct ProbeItem {
    uint32_t a_begin, a_end;  // elements of the A rows [a0, a0 + 128) in this octant
    uint32_t n_begin, n_end;  // "near" elements: the A rows' own (row-tagged, masked); first chunk only
    uint32_t b_begin, b_end;  // chunk of the elements of the rows behind the group (positions only)
    uint32_t a0;              // first A row (row index within the column)
};
constexpr int kProbeThreads = 1024;
// (256 rows x 4096 positions — half the passes over the elements, two 16-byte reads per lookup — is slower:
//  4.57 against 3.67 ms at c4's 20971 draws; the LDS reads are the larger half of the time)
constexpr uint32_t kProbeRows = 128;        // A rows per item
[[maybe_unused]] constexpr uint32_t kProbeWords = kProbeRows / 32u;  // (the mask words of the bitmap form)
constexpr uint32_t kProbeOctBits = 13;      // positions per table: 2^13 of the block's 2^16 (table = 2^13 x 16 B)
constexpr uint32_t kProbeOctants = 1u << (16 - kProbeOctBits);

// One workgroup = one item = one group of kProbeRows rows x one octant of the block, against a chunk of the positions
// of the rows behind the group. All it needs of the group is HOW MANY of its rows list each position: Cn, built by an
// LDS histogram of the group's elements (16 KiB: two workgroups of 1024 threads per CU).
//   * pairs inside the group (the group's first item only): C(Cn[p], 2) per position;
//   * pairs with a later row: Cn[p] per listed position p of that row — one 2-byte LDS read and one add.
// (Rounds 2 / 3 kept the group as a transposed BITMAP — 128-bit masks per position, 128 KiB — and took
//  popcount(mask & rows_before) per own element and popcount(mask) per later element; the counts are what those
//  popcounts add up to, and v_bcnt_u32_b32 is half rate: profiles/r03_b_sparse_probe_ab.txt has every step.)
template <int kT>
__global__ __launch_bounds__(kT, 8) void probe_lists_kernel(  // (8 waves per SIMD: two workgroups of 1024 per CU)
    const uint32_t* __restrict__ elems, const uint16_t* __restrict__ pos16,
    const ProbeItem* __restrict__ items, uint32_t item_stride, uint32_t item_first,
    unsigned long long* __restrict__ slots, unsigned long long* __restrict__ out, uint32_t fold_slots) {
    // two 16-bit counts per word: a group has at most kProbeRows = 128 rows
    __shared__ __attribute__((aligned(16))) uint32_t Cn32[(1u << kProbeOctBits) / 2u];
    __shared__ unsigned long long fold_ws[2 * (kT / 64)];
    const ProbeItem it = items[(uint64_t)blockIdx.x * item_stride + item_first];
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t kPosMask = (1u << kProbeOctBits) - 1u;
    static_assert(kProbeRows <= 0xffffu, "16-bit counts");
    for (uint32_t w = tid * 4u; w < (1u << kProbeOctBits) / 2u; w += (uint32_t)kT * 4u)
        *reinterpret_cast<uint4*>(&Cn32[w]) = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
    // The group's histogram, from the 2-byte positions of its atom (pos16 holds every atom in an order of its own —
    // dealt by LDS bank — and a histogram takes any order): 8 positions per 16-byte load, and the 64 adds of an
    // instruction fall on different banks. (Until round 4 it was built from the row-tagged 4-byte elements, one
    // per load in row order: with the far stream switched off a launch at c4's 20971 draws still took 0.23 of its
    // 0.93 ms — 30 us per item for 42 000 adds, profiles/r04_h_*.)
    {
        auto bump = [&](uint32_t p2) { atomicAdd(&Cn32[p2 >> 2], 1u << (8u * (p2 & 2u))); };   // p2 = 2 x position
        const uint32_t h_end = min(it.a_end, (it.a_begin + 7u) & ~7u);
        if (it.a_begin + tid < h_end) bump(pos16[it.a_begin + tid]);
        const uint32_t hb_end = h_end + ((it.a_end - h_end) & ~7u);
        if (hb_end > h_end) {
            const uint32_t h_last = hb_end - 8u;
            for (uint32_t q = h_end + tid * 8u; q < hb_end; q += 2u * (uint32_t)kT * 8u) {
                const uint32_t q1 = q + (uint32_t)kT * 8u;
                const uint4 v0 = *reinterpret_cast<const uint4*>(&pos16[q]);
                const uint4 v1 = *reinterpret_cast<const uint4*>(&pos16[min(q1, h_last)]);
                bump(v0.x & 0xffffu); bump(v0.x >> 16); bump(v0.y & 0xffffu); bump(v0.y >> 16);
                bump(v0.z & 0xffffu); bump(v0.z >> 16); bump(v0.w & 0xffffu); bump(v0.w >> 16);
                if (q1 < hb_end) {
                    bump(v1.x & 0xffffu); bump(v1.x >> 16); bump(v1.y & 0xffffu); bump(v1.y >> 16);
                    bump(v1.z & 0xffffu); bump(v1.z >> 16); bump(v1.w & 0xffffu); bump(v1.w >> 16);
                }
            }
        }
        if (hb_end + tid < it.a_end) bump(pos16[hb_end + tid]);
    }
    __syncthreads();
    uint32_t count = 0;
    if (it.n_end > it.n_begin) {  // the group's own rows among themselves
        // (one lookup per own element — sum_p C(Cn[p], 2) = half the sum of (Cn[p] - 1) over the group's elements — instead of
        //  this scan of all 8192 counters was built in round 6 and is 4 % SLOWER at every c4 load: it costs two barriers more)
        for (uint32_t w = tid; w < (1u << kProbeOctBits) / 2u; w += (uint32_t)kT) {
            const uint32_t c2 = Cn32[w], lo = c2 & 0xffffu, hi = c2 >> 16;
            count += (lo * (lo - 1u) + hi * (hi - 1u)) >> 1;  // both products are even
        }
    }
    // far: head up to a 16-byte boundary, body 8 positions per load, tail. pos16 holds 2 x the position: the byte
    // offset of its count (one instruction per address instead of two).
    const uint8_t* Cb = reinterpret_cast<const uint8_t*>(Cn32);
    auto visit = [&](uint32_t p2) { count += *reinterpret_cast<const uint16_t*>(Cb + (p2 & (2u * kPosMask))); };
    uint32_t e = it.b_begin;
    const uint32_t head_end = min(it.b_end, (it.b_begin + 7u) & ~7u);
    if (e + tid < head_end) visit(pos16[e + tid]);
    e = head_end;
    const uint32_t body_end = e + ((it.b_end - e) & ~7u);
    // Body: FOUR 16-byte pieces per lane and trip, loaded together and looked up whether or not they lie inside the
    // item (the address is clamped to the item's last piece, whose positions are valid ones; a piece outside counts
    // for nothing): behind a branch hipcc sinks the load to its use and waits for it with vmcnt(0). With one piece
    // per trip the stream was latency-bound (16 KiB in flight per CU). Fetching the NEXT trip's pieces under this
    // trip's lookups — inline-asm loads, counted waits that name the registers they release — was built in round 4
    // and is 2 - 15 % SLOWER (0.696 against 0.679 ms at 20971 draws, 0.093 against 0.079 at 524, same box,
    // profiles/r04_h_sparse_probe.txt): eight waves per SIMD already cover the round trip, and the stream runs at
    // 0.8 of what the LDS delivers for conflict-free 2-byte gathers (tools/probes/lds_gather_roof.hip).
    if (body_end > e) {
        constexpr uint32_t kStep = (uint32_t)kT * 8u;
        const uint32_t last = body_end - 8u;
        for (uint32_t q = e + tid * 8u; q < body_end; q += 4u * kStep) {
            uint4 v[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const uint4*>(&pos16[min(q + j * kStep, last)]);
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t p[8] = {v[j].x & 0xffffu, v[j].x >> 16, v[j].y & 0xffffu, v[j].y >> 16,
                                       v[j].z & 0xffffu, v[j].z >> 16, v[j].w & 0xffffu, v[j].w >> 16};
                uint32_t c[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) c[k] = *reinterpret_cast<const uint16_t*>(Cb + p[k]);
                const uint32_t s8 = c[0] + c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[7];
                count += (q + j * kStep < body_end) ? s8 : 0u;
            }
        }
    }
    if (body_end + tid < it.b_end) visit(pos16[body_end + tid]);
    uint64_t mine = count;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if (out == nullptr) {   // another kernel of the pass adds into the slots too: a fold launch follows
        if ((tid & 63u) == 0 && mine != 0)
            atomicAdd(&slots[(blockIdx.x * 16u + (tid >> 6)) & (kSlots - 1)], (unsigned long long)mine);
        return;
    }
    // ---- [r5] a pass of lists only, short launch: the fold inside the launch, as in strip16_bits_kernel — sum and ARRIVAL
    //      of a wave in one fire-and-forget atomic (low 48 / high 16 bits of the slot word), the workgroup dispatched last
    //      polls fold_slots slots (one per thread) until every wave of the grid has arrived, writes the total and leaves
    //      the slots zeroed. At a few hundred positions per row a call is mostly launches: this removes one of two.
    if ((tid & 63u) == 0)
        atomicAdd(&slots[(blockIdx.x * (uint32_t)(kT / 64) + (tid >> 6)) & (fold_slots - 1u)],
                  (unsigned long long)mine + (1ull << 48));
    if (blockIdx.x != gridDim.x - 1u) return;
    const unsigned long long expected = (unsigned long long)gridDim.x * (unsigned long long)(kT / 64);
    unsigned long long total = 0;
    for (;;) {
        unsigned long long cnt = 0, sum = 0;
        for (uint32_t i = tid; i < fold_slots; i += (uint32_t)kT) {
            const unsigned long long v = __hip_atomic_load(&slots[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cnt += v >> 48;
            sum += v & ((1ull << 48) - 1ull);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            cnt += __shfl_down(cnt, o, 64);
            sum += __shfl_down(sum, o, 64);
        }
        __syncthreads();
        if ((tid & 63u) == 0) {
            fold_ws[tid >> 6] = cnt;
            fold_ws[(kT / 64) + (tid >> 6)] = sum;
        }
        __syncthreads();
        cnt = 0;
        total = 0;
#pragma unroll
        for (int w = 0; w < kT / 64; ++w) {
            cnt += fold_ws[w];
            total += fold_ws[(kT / 64) + w];
        }
        if (cnt == expected) break;
    }
    for (uint32_t i = tid; i < fold_slots; i += (uint32_t)kT)
        __hip_atomic_store(&slots[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) out[0] = total;
}

// ------------------------------------------------------------------------------------------
// [r6] K4 with FAT workgroups (probe_lists_fat_kernel): one workgroup = a BUNDLE of kFatGroups = 4 consecutive groups of
// a (column, octant) stream against a chunk of the stream behind the bundle. probe_lists_kernel's items are all fixed cost
// at the sparse end (5056 workgroups at c4's 104 draws per row: the item record, the group's positions and the far
// positions are three dependent trips to memory, then a 16 KiB table zeroed, filled and scanned, for ~8000 lookups):
//   * the far positions are loaded ONCE per bundle and looked up in four tables (a quarter of the loads and of the
//     workgroups, the fixed costs shared by four groups);
//   * the first far pieces are on their way BEFORE the tables are built (their addresses need the item record only);
//   * the pairs inside a group, sum_p C(Cn[p], 2), come from the group's own elements — (Cn[p] - 1) per element, halved
//     once per workgroup (the sum over a group is n (n - 1) per position: even) — instead of a scan of all 8192
//     counters; the pairs between two groups of the bundle from the later group's elements in the earlier group's table.
// Tables: 4 x 16 KiB of 16-bit counts. Same slots / in-launch fold as probe_lists_kernel.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kFatGroups = 4;
struct ProbeFatItem {
    uint32_t at[kFatGroups + 1];   // elements of group k of the bundle in this octant: [at[k], at[k + 1])
    uint32_t b_begin, b_end;       // chunk of the elements of the rows behind the bundle
    uint32_t first;                // 1: the bundle's own pairs (inside and between its groups) belong to this item
};
template <int kT>
__global__ __launch_bounds__(kT, kT / 128) void probe_lists_fat_kernel(   // (two workgroups per CU by LDS: 2 x 64 KiB)
    const uint16_t* __restrict__ pos16, const ProbeFatItem* __restrict__ items, uint32_t item_stride, uint32_t item_first,
    unsigned long long* __restrict__ slots, unsigned long long* __restrict__ out, uint32_t fold_slots) {
    constexpr uint32_t kTableWords = (1u << kProbeOctBits) / 2u;   // two 16-bit counts per word
    __shared__ __attribute__((aligned(16))) uint32_t Cn32[kFatGroups * kTableWords];
    __shared__ unsigned long long fold_ws[2 * (kT / 64)];
    const ProbeFatItem it = items[(uint64_t)blockIdx.x * item_stride + item_first];
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t kOff = 2u * ((1u << kProbeOctBits) - 1u);   // pos16 holds 2 x the position: a byte offset
    // far body: the first trip's pieces leave before anything else (clamped to the item's last piece, as below)
    uint32_t e = it.b_begin;
    const uint32_t head_end = min(it.b_end, (it.b_begin + 7u) & ~7u);
    const uint32_t body0 = head_end;
    const uint32_t body_end = body0 + ((it.b_end - body0) & ~7u);
    constexpr uint32_t kStep = (uint32_t)kT * 8u;
    const bool has_body = body_end > body0;
    const uint32_t last = has_body ? body_end - 8u : 0u;
    uint4 v[2];
    if (has_body) {
#pragma unroll
        for (uint32_t j = 0; j < 2; ++j) v[j] = *reinterpret_cast<const uint4*>(&pos16[min(body0 + tid * 8u + j * kStep, last)]);
    }
    for (uint32_t w = tid * 4u; w < kFatGroups * kTableWords; w += (uint32_t)kT * 4u)
        *reinterpret_cast<uint4*>(&Cn32[w]) = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
    // the four histograms (any order of an atom's positions will do; pos16 holds them dealt by LDS bank)
#pragma unroll
    for (uint32_t k = 0; k < kFatGroups; ++k) {
        auto bump = [&](uint32_t p2) { atomicAdd(&Cn32[k * kTableWords + ((p2 & kOff) >> 2)], 1u << (8u * (p2 & 2u))); };
        const uint32_t a0 = it.at[k], a1 = it.at[k + 1];
        const uint32_t h_end = min(a1, (a0 + 7u) & ~7u);
        if (a0 + tid < h_end) bump(pos16[a0 + tid]);
        const uint32_t hb_end = h_end + ((a1 - h_end) & ~7u);
        for (uint32_t q = h_end + tid * 8u; q < hb_end; q += (uint32_t)kT * 8u) {
            const uint4 v0 = *reinterpret_cast<const uint4*>(&pos16[q]);
            bump(v0.x & 0xffffu); bump(v0.x >> 16); bump(v0.y & 0xffffu); bump(v0.y >> 16);
            bump(v0.z & 0xffffu); bump(v0.z >> 16); bump(v0.w & 0xffffu); bump(v0.w >> 16);
        }
        if (hb_end + tid < a1) bump(pos16[hb_end + tid]);
    }
    __syncthreads();
    const uint8_t* Cb = reinterpret_cast<const uint8_t*>(Cn32);
    auto look = [&](uint32_t k, uint32_t p2) { return (uint32_t)*reinterpret_cast<const uint16_t*>(Cb + k * (kTableWords * 4u) + p2); };
    uint32_t count = 0;
    if (it.first) {
        // the bundle's own pairs: an element of group k against the groups in front of it, and (twice) inside its own
        uint32_t self2 = 0;
        // (in pieces of 8 positions per lane, as the far stream: consecutive positions of an atom share an LDS bank —
        //  they are dealt that way — and lanes that take consecutive positions collide eightfold: 0.34 of the LDS cycles
        //  were bank conflicts against 0.09 in probe_lists_kernel, profiles/r06_*_probe_bundle.txt)
#pragma unroll
        for (uint32_t k = 0; k < kFatGroups; ++k) {
            auto own = [&](uint32_t p2) {
                p2 &= kOff;
                self2 += look(k, p2) - 1u;
#pragma unroll
                for (uint32_t j = 0; j < kFatGroups; ++j)
                    if (j < k) count += look(j, p2);
            };
            const uint32_t a0 = it.at[k], a1 = it.at[k + 1];
            const uint32_t h_end = min(a1, (a0 + 7u) & ~7u);
            if (a0 + tid < h_end) own(pos16[a0 + tid]);
            const uint32_t hb_end = h_end + ((a1 - h_end) & ~7u);
            for (uint32_t q = h_end + tid * 8u; q < hb_end; q += (uint32_t)kT * 8u) {
                const uint4 v0 = *reinterpret_cast<const uint4*>(&pos16[q]);
                own(v0.x & 0xffffu); own(v0.x >> 16); own(v0.y & 0xffffu); own(v0.y >> 16);
                own(v0.z & 0xffffu); own(v0.z >> 16); own(v0.w & 0xffffu); own(v0.w >> 16);
            }
            if (hb_end + tid < a1) own(pos16[hb_end + tid]);
        }
        // sum over the workgroup, halved (exact: n (n - 1) per position)
        uint64_t s2 = self2;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s2 += __shfl_down(s2, o, 64);
        if ((tid & 63u) == 0) fold_ws[tid >> 6] = s2;
        __syncthreads();
        if (tid == 0) {
            unsigned long long t = 0;
#pragma unroll
            for (int w = 0; w < kT / 64; ++w) t += fold_ws[w];
            count += (uint32_t)(t >> 1);
        }
        __syncthreads();   // (fold_ws is used again below)
    }
    // far: every position against the four tables (a missing group's table is empty)
    auto visit4 = [&](uint32_t p2) {
        p2 &= kOff;
        count += look(0, p2) + look(1, p2) + look(2, p2) + look(3, p2);
    };
    if (e + tid < head_end) visit4(pos16[e + tid]);
    if (has_body) {
        for (uint32_t q = body0 + tid * 8u; q < body_end; q += 2u * kStep) {
            uint4 nx[2];
#pragma unroll
            for (uint32_t j = 0; j < 2; ++j) nx[j] = *reinterpret_cast<const uint4*>(&pos16[min(q + (2u + j) * kStep, last)]);
#pragma unroll
            for (uint32_t j = 0; j < 2; ++j) {
                const uint32_t p[8] = {v[j].x & 0xffffu, v[j].x >> 16, v[j].y & 0xffffu, v[j].y >> 16,
                                       v[j].z & 0xffffu, v[j].z >> 16, v[j].w & 0xffffu, v[j].w >> 16};
                uint32_t s8 = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t p2 = p[k] & kOff;
                    s8 += look(0, p2) + look(1, p2) + look(2, p2) + look(3, p2);
                }
                count += (q + j * kStep < body_end) ? s8 : 0u;
            }
            v[0] = nx[0];
            v[1] = nx[1];
        }
    }
    if (body_end + tid < it.b_end) visit4(pos16[body_end + tid]);
    uint64_t mine = count;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if (out == nullptr) {   // another kernel of the pass adds into the slots too: a fold launch follows
        if ((tid & 63u) == 0 && mine != 0)
            atomicAdd(&slots[(blockIdx.x * 16u + (tid >> 6)) & (kSlots - 1)], (unsigned long long)mine);
        return;
    }
    // the fold inside the launch, as in probe_lists_kernel
    if ((tid & 63u) == 0)
        atomicAdd(&slots[(blockIdx.x * (uint32_t)(kT / 64) + (tid >> 6)) & (fold_slots - 1u)],
                  (unsigned long long)mine + (1ull << 48));
    if (blockIdx.x != gridDim.x - 1u) return;
    const unsigned long long expected = (unsigned long long)gridDim.x * (unsigned long long)(kT / 64);
    unsigned long long total = 0;
    for (;;) {
        unsigned long long cnt = 0, sum = 0;
        for (uint32_t i = tid; i < fold_slots; i += (uint32_t)kT) {
            const unsigned long long x = __hip_atomic_load(&slots[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cnt += x >> 48;
            sum += x & ((1ull << 48) - 1ull);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            cnt += __shfl_down(cnt, o, 64);
            sum += __shfl_down(sum, o, 64);
        }
        __syncthreads();
        if ((tid & 63u) == 0) {
            fold_ws[tid >> 6] = cnt;
            fold_ws[(kT / 64) + (tid >> 6)] = sum;
        }
        __syncthreads();
        cnt = 0;
        total = 0;
#pragma unroll
        for (int w = 0; w < kT / 64; ++w) {
            cnt += fold_ws[w];
            total += fold_ws[(kT / 64) + w];
        }
        if (cnt == expected) break;
    }
    for (uint32_t i = tid; i < fold_slots; i += (uint32_t)kT)
        __hip_atomic_store(&slots[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) out[0] = total;
}

// Pool rows of a probe column from its probe elements (ensure_full_pool): one workgroup per (column, octant)
// region {first element, end, first pool row of the column, octant}
__global__ __launch_bounds__(kThreads) void expand_probe_kernel(uint64_t* __restrict__ pool, uint64_t pitch,
                                                                const uint32_t* __restrict__ elems,
                                                                const uint32_t* __restrict__ regions) {
    const uint32_t e0 = regions[blockIdx.x * 4u + 0], e1 = regions[blockIdx.x * 4u + 1];
    const uint32_t row0 = regions[blockIdx.x * 4u + 2], oct = regions[blockIdx.x * 4u + 3];
    for (uint32_t e = e0 + threadIdx.x; e < e1; e += kThreads) {
        const uint32_t v = elems[e];
        const uint32_t pos = (oct << kProbeOctBits) | (v & ((1u << kProbeOctBits) - 1u));
        unsigned long long* row = reinterpret_cast<unsigned long long*>(pool + (uint64_t)(row0 + (v >> 16)) * pitch);
        atomicOr(&row[pos >> 6], 1ull << (pos & 63u));
    }
}


// ---- arena construction on the device (round 4) ------------------------------------------------------------
// A first all-pairs call on a fresh STORM_t used to cost 0.6 - 0.9 s at c4's 20971 draws per row: the host laid
// out 210 M list elements (0.35 s on four threads), after flattening the container (0.1 s) and in front of 1.7 GB of
// uploads from pageable memory. Now the host only WALKS THE BLOCK HEADERS (where every run of a list begins and
// where it goes: 8 binary searches per block) and ships the raw lists and bitmaps through a small pinned ring;
// the elements are laid out here.

// One workgroup per list block of a probe column: element k of the list goes to run_dst[octant] + (k - start of
// the octant's run) as (row in column) << 16 | position in octant, and as a byte offset (2 x position) for the far
// stream. meta[b] = {list offset (uint16 units), length, tag = row in column << 16, pad}; run_end / run_dst: 8 per block.
// Where the runs of a sorted list end: run_end[8 b + o] = the first index of list b whose value is >= (o + 1) * 8192
// (one thread per list and octant; the host did these 640 000 binary searches over cold lists in 65 of the 89 ms a
// first call took at c4's 20971 draws per row — the lists have to travel to the device anyway).
__global__ __launch_bounds__(kThreads) void probe_run_end_kernel(const uint16_t* __restrict__ lists,
                                                                 const uint64_t* __restrict__ list_off,
                                                                 const uint32_t* __restrict__ list_len, uint32_t n_lists,
                                                                 uint32_t* __restrict__ run_end) {
    const uint32_t idx = blockIdx.x * kThreads + threadIdx.x;
    const uint32_t b = idx / kProbeOctants, o = idx % kProbeOctants;
    if (b >= n_lists) return;
    const uint16_t* l = lists + list_off[b];
    const uint32_t lim = (o + 1u) << kProbeOctBits;
    uint32_t lo = 0, hi = list_len[b];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint32_t)l[mid] < lim) lo = mid + 1u;
        else hi = mid;
    }
    run_end[idx] = lo;
}

// A list that is not strictly ascending sets *bad (the probe kernel counts every listed element: none may repeat).
__global__ __launch_bounds__(kThreads) void probe_fill_kernel(const uint16_t* __restrict__ lists,
                                                              const uint64_t* __restrict__ list_off,
                                                              const uint32_t* __restrict__ list_len,
                                                              const uint32_t* __restrict__ tags,
                                                              const uint32_t* __restrict__ run_end,
                                                              const uint32_t* __restrict__ run_dst,
                                                              uint32_t* __restrict__ elems, uint16_t* __restrict__ pos16,
                                                              uint32_t* __restrict__ bad) {
    const uint32_t b = blockIdx.x;
    const uint16_t* l = lists + list_off[b];
    const uint32_t n = list_len[b], tag = tags[b];
    uint32_t end[kProbeOctants], dst[kProbeOctants];
#pragma unroll
    for (uint32_t o = 0; o < kProbeOctants; ++o) {
        end[o] = run_end[(uint64_t)b * kProbeOctants + o];
        dst[o] = run_dst[(uint64_t)b * kProbeOctants + o];
    }
    bool ok = true;
    for (uint32_t k = threadIdx.x; k < n; k += kThreads) {
        const uint32_t v = l[k];
        if (k && l[k - 1] >= v) ok = false;
        uint32_t o = 0, from = 0;
#pragma unroll
        for (uint32_t q = 0; q + 1 < kProbeOctants; ++q)
            if (k >= end[q]) { o = q + 1; from = end[q]; }
        // (the host's run ends are lower bounds of the octant limits: a list that is not ascending may disagree
        //  with them, and is refused below)
        const uint32_t at = dst[o] + (k - from);
        const uint32_t pos = v & ((1u << kProbeOctBits) - 1u);
        elems[at] = tag | pos;
        pos16[at] = (uint16_t)(pos << 1);
        if ((v >> kProbeOctBits) != o) ok = false;
    }
    if (!ok) atomicOr(bad, 1u);
}

// The far stream in an order of its own (see build_arena): inside every atom — the elements of one group of
// kProbeRows rows in one octant — the positions are dealt by LDS bank, eight of bank 0, eight of bank 1, ... round
// after round. One workgroup per atom {first element, end}: bank counts, then every element's slot from its bank,
// its rank inside the bank (an LDS counter: any order will do, the positions of an atom may stand in any order)
// and the counts: all of the earlier rounds, the banks in front of it in its own round. Atoms of fewer than 256
// elements keep their row order.
__global__ __launch_bounds__(kThreads) void probe_deal_kernel(const uint32_t* __restrict__ atoms,
                                                              const uint16_t* __restrict__ src, uint16_t* __restrict__ dst) {
    __shared__ uint32_t cnt[32], cur[32];
    const uint32_t s0 = atoms[blockIdx.x * 2u], s1 = atoms[blockIdx.x * 2u + 1u];
    const uint32_t n = s1 - s0;
    if (n < 256u) {
        for (uint32_t j = s0 + threadIdx.x; j < s1; j += kThreads) dst[j] = src[j];
        return;
    }
    if (threadIdx.x < 32) cnt[threadIdx.x] = cur[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t j = s0 + threadIdx.x; j < s1; j += kThreads) atomicAdd(&cnt[(src[j] >> 2) & 31u], 1u);
    __syncthreads();
    for (uint32_t j = s0 + threadIdx.x; j < s1; j += kThreads) {
        const uint16_t v = src[j];
        const uint32_t bank = (v >> 2) & 31u;
        const uint32_t k = atomicAdd(&cur[bank], 1u);
        const uint32_t round8 = (k >> 3) * 8u;
        uint32_t before = k & 7u;
#pragma unroll 8
        for (uint32_t q = 0; q < 32; ++q) {
            const uint32_t c = cnt[q];
            before += min(c, round8);                                       // all earlier rounds
            before += q < bank ? min(8u, c > round8 ? c - round8 : 0u) : 0u;  // this round, the banks in front
        }
        dst[s0 + before] = v;
    }
}

// (Piece / Stager: the pinned staging ring, shared with the row lists of storm_hip_lists.hip: storm_hip_internal.h)

template <typename T>
int upload(storm_hip_ctx_t* ctx, T** d, const T* h, size_t n) {
    *d = nullptr;
    if (n == 0) return STORM_HIP_OK;
    STORM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(d), n * sizeof(T)));
    return upload_bytes(ctx, *d, h, n * sizeof(T));
}

}  // namespace

// ---- [r6] block stage: bitmap blocks travel to the device WHILE the container is being built ---------------------
// A STORM_t of dense rows is 8 KiB per block (655 MB at BASELINE c4's 50 % load): built from pointers at the first
// all-pairs call (storm_hip_sparse_create_blocks) those bytes cross the bus inside that call — 91 ms, the one call the
// reference's harness times (benchmark.cpp:605-613). STORM_add hands every bitmap block it finishes to a stage instead:
// 8 KiB into a pinned ring, 4 MiB at a time on its way into 64 MiB device chunks, a token back. The arena build then
// gathers the pool rows from the chunks with one kernel (storm_hip_sparse_create_blocks_staged) — the regrouping by
// block column needs every row, so the final layout cannot be streamed into.
struct storm_hip_stage_s {
    static constexpr size_t kBlockBytes = 8192, kChunkBlocks = 8192, kBufBlocks = 512;
    std::vector<uint8_t*> chunks;   // 64 MiB each
    uint64_t n_blocks = 0;          // tokens handed out
    uint8_t* h_ring = nullptr;      // two buffers of kBufBlocks blocks, pinned
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    int cur = 0;
    uint32_t fill = 0;              // blocks in the current buffer
    uint64_t base = 0;              // token of the current buffer's first block
    const uint8_t** d_chunk_table = nullptr;   // the chunks' addresses on the device (rebuilt per gather)
    // the LIST blocks: a byte stream of its own ("list space": position p = chunk p / 64 MiB, offset p % 64 MiB; a list never
    // straddles a chunk), through two pinned buffers of kListBuf bytes behind the bitmaps' ring in the same allocation
    static constexpr size_t kListBuf = 4u << 20, kListChunk = 64u << 20;
    std::vector<uint8_t*> lchunks;
    uint8_t* h_lring = nullptr;
    hipEvent_t lev[2] = {nullptr, nullptr};
    bool lused[2] = {false, false};
    int lcur = 0;
    uint32_t lfill = 0;             // bytes in the current buffer
    uint64_t lbase = 0;             // list-space position of the current buffer's first byte
    const uint8_t** d_lchunk_table = nullptr;
};

namespace {

int stage_send(storm_hip_ctx_t* ctx, storm_hip_stage_t* st) {
    if (st->fill == 0) return STORM_HIP_OK;
    const uint8_t* buf = st->h_ring + (size_t)st->cur * storm_hip_stage_s::kBufBlocks * storm_hip_stage_s::kBlockBytes;
    uint64_t t = st->base, left = st->fill;
    while (left) {   // (a buffer that was sent before it was full may run across a chunk boundary)
        const uint64_t chunk = t / storm_hip_stage_s::kChunkBlocks, off = t % storm_hip_stage_s::kChunkBlocks;
        while (st->chunks.size() <= chunk) {
            uint8_t* c = nullptr;
            if (hipMalloc(reinterpret_cast<void**>(&c), storm_hip_stage_s::kChunkBlocks * storm_hip_stage_s::kBlockBytes) != hipSuccess) {
                set_error("block stage: hipMalloc of a 64 MiB chunk failed");
                return STORM_HIP_ENOMEM;
            }
            st->chunks.push_back(c);
        }
        const uint64_t n = std::min<uint64_t>(left, storm_hip_stage_s::kChunkBlocks - off);
        STORM_HIP_TRY(hipMemcpyAsync(st->chunks[chunk] + off * storm_hip_stage_s::kBlockBytes,
                                     buf + (t - st->base) * storm_hip_stage_s::kBlockBytes, n * storm_hip_stage_s::kBlockBytes,
                                     hipMemcpyHostToDevice, ctx->stream));
        t += n;
        left -= n;
    }
    STORM_HIP_TRY(hipEventRecord(st->ev[st->cur], ctx->stream));
    st->used[st->cur] = true;
    st->cur ^= 1;
    st->base += st->fill;
    st->fill = 0;
    if (st->used[st->cur]) STORM_HIP_TRY(hipEventSynchronize(st->ev[st->cur]));   // the buffer about to be filled has left
    return STORM_HIP_OK;
}

// the current list buffer -> its chunk (a buffer never runs across a chunk boundary: storm_hip_stage_add_list)
int stage_send_lists(storm_hip_ctx_t* ctx, storm_hip_stage_t* st) {
    if (st->lfill == 0) return STORM_HIP_OK;
    const uint64_t chunk = st->lbase / storm_hip_stage_s::kListChunk, off = st->lbase % storm_hip_stage_s::kListChunk;
    while (st->lchunks.size() <= chunk) {
        uint8_t* c = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&c), storm_hip_stage_s::kListChunk) != hipSuccess) {
            set_error("block stage: hipMalloc of a 64 MiB list chunk failed");
            return STORM_HIP_ENOMEM;
        }
        st->lchunks.push_back(c);
    }
    STORM_HIP_TRY(hipMemcpyAsync(st->lchunks[chunk] + off, st->h_lring + (size_t)st->lcur * storm_hip_stage_s::kListBuf, st->lfill,
                                 hipMemcpyHostToDevice, ctx->stream));
    STORM_HIP_TRY(hipEventRecord(st->lev[st->lcur], ctx->stream));
    st->lused[st->lcur] = true;
    st->lcur ^= 1;
    st->lbase += st->lfill;
    st->lfill = 0;
    if (st->lused[st->lcur]) STORM_HIP_TRY(hipEventSynchronize(st->lev[st->lcur]));
    return STORM_HIP_OK;
}

// lists[dst ..) <- the staged list at list-space position `token`: one workgroup per block, 2 bytes per thread and trip
__global__ __launch_bounds__(256) void gather_staged_lists_kernel(const uint8_t* const* __restrict__ lchunks,
                                                                  const uint64_t* __restrict__ table, uint16_t* __restrict__ lists) {
    const uint64_t dst = table[3 * blockIdx.x], token = table[3 * blockIdx.x + 1];
    const uint32_t n = (uint32_t)table[3 * blockIdx.x + 2];
    const uint16_t* src = reinterpret_cast<const uint16_t*>(lchunks[token / storm_hip_stage_s::kListChunk] +
                                                            token % storm_hip_stage_s::kListChunk);
    for (uint32_t i = threadIdx.x; i < n; i += 256u) lists[dst + i] = src[i];
}

// pool row `dst` <- staged block `token`: one workgroup per block, 32 bytes per thread
__global__ __launch_bounds__(256) void gather_staged_kernel(const uint8_t* const* __restrict__ chunks,
                                                            const uint64_t* __restrict__ table, uint64_t* __restrict__ pool,
                                                            uint64_t pitch_words) {
    const uint64_t dst = table[2 * blockIdx.x], token = table[2 * blockIdx.x + 1];
    const uint4* src = reinterpret_cast<const uint4*>(chunks[token / storm_hip_stage_s::kChunkBlocks] +
                                                      (token % storm_hip_stage_s::kChunkBlocks) * storm_hip_stage_s::kBlockBytes);
    uint4* out = reinterpret_cast<uint4*>(pool + dst * pitch_words);
    out[threadIdx.x] = src[threadIdx.x];
    out[threadIdx.x + 256] = src[threadIdx.x + 256];
}

}  // namespace

namespace storm {
// lists[dst ..) <- the staged lists of `ltable` (three words per block: destination element, list token, length): what is
// still in the stage's ring goes first; the table's device copy is handed back for the caller to free. Waits for the stream
// (the tables are pageable). Tokens are checked against what the stage holds.
int stage_gather_lists(storm_hip_ctx_t* ctx, storm_hip_stage_t* stage, const std::vector<uint64_t>& ltable, uint16_t* d_lists,
                       uint64_t** d_table) {
    *d_table = nullptr;
    if (ltable.empty()) return STORM_HIP_OK;
    const uint64_t list_space = stage->lbase + stage->lfill;
    for (size_t k = 0; k < ltable.size(); k += 3)
        if ((ltable[k + 1] & 1u) || ltable[k + 1] + ltable[k + 2] * 2u > list_space) {
            set_error("block stage: a list token lies outside the stage");
            return STORM_HIP_EINVAL;
        }
    if (int rc0 = stage_send_lists(ctx, stage)) return rc0;
    if (stage->d_lchunk_table) ctx->deferred_free.push_back(stage->d_lchunk_table);
    stage->d_lchunk_table = nullptr;
    STORM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&stage->d_lchunk_table), stage->lchunks.size() * sizeof(uint8_t*)));
    STORM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(d_table), ltable.size() * sizeof(uint64_t)));
    STORM_HIP_TRY(hipMemcpyAsync(stage->d_lchunk_table, stage->lchunks.data(), stage->lchunks.size() * sizeof(uint8_t*),
                                 hipMemcpyHostToDevice, ctx->stream));
    if (int rc0 = upload_bytes(ctx, *d_table, ltable.data(), ltable.size() * sizeof(uint64_t))) return rc0;
    hipLaunchKernelGGL(gather_staged_lists_kernel, dim3((uint32_t)(ltable.size() / 3)), dim3(256), 0, ctx->stream,
                       stage->d_lchunk_table, *d_table, d_lists);
    STORM_HIP_TRY(hipGetLastError());
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STORM_HIP_OK;
}
}  // namespace storm

int storm_hip_stage_create(storm_hip_ctx_t* ctx, storm_hip_stage_t** out) {
    return guarded("storm_hip_stage_create", [&]() -> int {
        if (!ctx || !out) {
            set_error("stage_create: NULL context or output");
            return STORM_HIP_EINVAL;
        }
        *out = nullptr;
        STORM_HIP_TRY(hipSetDevice(ctx->device));
        std::unique_ptr<storm_hip_stage_t> st(new storm_hip_stage_t());
        const size_t ring_bytes = 2 * storm_hip_stage_s::kBufBlocks * storm_hip_stage_s::kBlockBytes;
        if (hipHostMalloc(reinterpret_cast<void**>(&st->h_ring), ring_bytes + 2 * storm_hip_stage_s::kListBuf,
                          hipHostMallocDefault) != hipSuccess) {
            set_error("stage_create: hipHostMalloc of the 16 MiB of rings failed");
            return STORM_HIP_ENOMEM;
        }
        st->h_lring = st->h_ring + ring_bytes;
        (void)storm_hip_ctx_reserve_staging(ctx);   // the arena builder's ring, too, while nobody is waiting
        for (int i = 0; i < 2; ++i)
            if (hipEventCreateWithFlags(&st->ev[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&st->lev[i], hipEventDisableTiming) != hipSuccess) {
                storm_hip_stage_destroy(ctx, st.release());
                return STORM_HIP_EHIP;
            }
        *out = st.release();
        return STORM_HIP_OK;
    });
}

int storm_hip_stage_add(storm_hip_ctx_t* ctx, storm_hip_stage_t* st, const uint64_t* words, uint64_t* token) {
    return guarded("storm_hip_stage_add", [&]() -> int {
        if (!ctx || !st || !words || !token) {
            set_error("stage_add: NULL argument");
            return STORM_HIP_EINVAL;
        }
        memcpy(st->h_ring + ((size_t)st->cur * storm_hip_stage_s::kBufBlocks + st->fill) * storm_hip_stage_s::kBlockBytes, words,
               storm_hip_stage_s::kBlockBytes);
        *token = st->n_blocks++;
        if (++st->fill == storm_hip_stage_s::kBufBlocks) {
            STORM_HIP_TRY(hipSetDevice(ctx->device));
            return stage_send(ctx, st);
        }
        return STORM_HIP_OK;
    });
}

int storm_hip_stage_add_list(storm_hip_ctx_t* ctx, storm_hip_stage_t* st, const uint16_t* list, uint32_t n, uint64_t* token) {
    return guarded("storm_hip_stage_add_list", [&]() -> int {
        if (!ctx || !st || !list || !token || n == 0 || n > 65536u) {
            set_error("stage_add_list: NULL argument or a list of %u positions", n);
            return STORM_HIP_EINVAL;
        }
        const uint32_t bytes = n * 2u;
        const uint64_t at = st->lbase + st->lfill;
        const bool over_chunk = at % storm_hip_stage_s::kListChunk + bytes > storm_hip_stage_s::kListChunk;
        if (over_chunk || st->lfill + bytes > storm_hip_stage_s::kListBuf) {
            STORM_HIP_TRY(hipSetDevice(ctx->device));
            if (int rc = stage_send_lists(ctx, st)) return rc;
            if (over_chunk) st->lbase = (st->lbase / storm_hip_stage_s::kListChunk + 1u) * storm_hip_stage_s::kListChunk;
        }
        memcpy(st->h_lring + (size_t)st->lcur * storm_hip_stage_s::kListBuf + st->lfill, list, bytes);
        *token = st->lbase + st->lfill;
        st->lfill += bytes;
        return STORM_HIP_OK;
    });
}

uint64_t storm_hip_stage_count(const storm_hip_stage_t* st) { return st ? st->n_blocks : 0; }

void storm_hip_stage_destroy(storm_hip_ctx_t* ctx, storm_hip_stage_t* st) {
    if (!st) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);   // copies out of the ring, gathers out of the chunks
    }
    // (put off: ten 64 MiB chunks and the ring cost 3 ms to release, inside the first all-pairs call)
    for (uint8_t* c : st->chunks) {
        if (ctx) ctx->deferred_free.push_back(c);
        else (void)hipFree(c);
    }
    for (uint8_t* c : st->lchunks) {
        if (ctx) ctx->deferred_free.push_back(c);
        else (void)hipFree(c);
    }
    for (const uint8_t** t : {st->d_chunk_table, st->d_lchunk_table})
        if (t) {
            if (ctx) ctx->deferred_free.push_back(t);
            else (void)hipFree(t);
        }
    if (st->h_ring) {
        if (ctx) ctx->deferred_host_free.push_back(st->h_ring);
        else (void)hipHostFree(st->h_ring);
    }
    for (hipEvent_t e : st->ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : st->lev)
        if (e) (void)hipEventDestroy(e);
    delete st;
}

// Builds the device arena from the flat block description of storm_hip.h. `bitmaps_in_stream`:
// the bitmap blocks' words are not in `bitmap_pool` but inside `list_pool` itself (then a byte
// stream viewed as uint16, block_data_offset of a bitmap block in uint16 units): the whole stream
// is uploaded once and both block kinds are unpacked from it on the device.
static int build_arena(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                       const uint64_t* row_block_offset, const uint32_t* block_id,
                       const uint8_t* block_kind, const uint32_t* block_n,
                       const void* const* block_ptr, storm_hip_sparse_t** out,
                       storm_hip_stage_t* stage = nullptr, const uint64_t* stage_token = nullptr) {
    if (!ctx || !out) {
        set_error("sparse_create: NULL context or output");
        return STORM_HIP_EINVAL;
    }
    *out = nullptr;
    if (n_blocks > 0 && (!row_block_offset || !block_id || !block_kind || !block_ptr || !block_n)) {
        set_error("sparse_create: NULL descriptor array");
        return STORM_HIP_EINVAL;
    }
    if (n_blocks >= (1ull << 32) - kABlockRows) {
        set_error("sparse_create: too many blocks");
        return STORM_HIP_EINVAL;
    }
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (getenv("STORM_HIP_TIMING")) {
            auto t = std::chrono::steady_clock::now();
            fprintf(stderr, "[build_arena] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count());
            T0 = t;
        }
    };
    // ---- validate + count blocks per column ----
    if (n_blocks > 0 && n_rows == 0) {
        set_error("sparse_create: %llu blocks but no rows", (unsigned long long)n_blocks);
        return STORM_HIP_EINVAL;
    }
    if (n_rows > 0 && (!row_block_offset || row_block_offset[0] != 0 ||
                       row_block_offset[n_rows] != n_blocks)) {
        set_error("sparse_create: row_block_offset must run from 0 to n_blocks = %llu",
                  (unsigned long long)n_blocks);
        return STORM_HIP_EINVAL;
    }
    uint32_t max_id = 0;
    for (uint64_t r = 0; r < n_rows; ++r) {
        if (row_block_offset[r] > row_block_offset[r + 1] ||
            row_block_offset[r + 1] > n_blocks) {
            set_error("sparse_create: row_block_offset is not a CSR over %llu blocks",
                      (unsigned long long)n_blocks);
            return STORM_HIP_EINVAL;
        }
        for (uint64_t b = row_block_offset[r]; b < row_block_offset[r + 1]; ++b) {
            if (b > row_block_offset[r] && block_id[b] <= block_id[b - 1]) {
                set_error("sparse_create: block ids of row %llu are not ascending",
                          (unsigned long long)r);
                return STORM_HIP_EINVAL;
            }
            if (block_id[b] >= kMaxBlockId) {  // positions are uint32: ids stop at 2^32 / 65536
                set_error("sparse_create: block id %u out of range", block_id[b]);
                return STORM_HIP_EINVAL;
            }
            if (block_kind[b] > 1) {
                set_error("sparse_create: block kind %u", block_kind[b]);
                return STORM_HIP_EINVAL;
            }
            if (block_kind[b] == 0 ? (block_n[b] > 65536u || (block_n[b] && (!block_ptr[b] || ((uintptr_t)block_ptr[b] & 1))))
                                   : !block_ptr[b]) {
                set_error("sparse_create: block %llu has no data (or a list that is too long or not 2-byte aligned)",
                          (unsigned long long)b);
                return STORM_HIP_EINVAL;
            }
            max_id = std::max(max_id, block_id[b]);
        }
    }
    std::vector<uint64_t> per_col((size_t)max_id + 2, 0), n_list_col((size_t)max_id + 2, 0);
    for (uint64_t b = 0; b < n_blocks; ++b) {
        per_col[block_id[b]]++;
        if (block_kind[b] == 0) n_list_col[block_id[b]]++;
    }
    // owned here until it is handed to the caller: a std::vector below may throw (guarded() turns that into
    // ENOMEM) and every early return must release the arena and its device buffers
    struct ArenaDeleter {
        storm_hip_ctx_t* ctx;
        void operator()(storm_hip_sparse_t* a) const { storm_hip_sparse_destroy(ctx, a); }
    };
    std::unique_ptr<storm_hip_sparse_t, ArenaDeleter> owner(new (std::nothrow) storm_hip_sparse_t(), ArenaDeleter{ctx});
    storm_hip_sparse_t* s = owner.get();
    if (!s) return STORM_HIP_ENOMEM;
    // Which columns the list-probe kernel (K4) can take: all blocks lists, 2 .. 65535 rows, element offsets
    // within 32 bits (8 octants x up to 7 elements of alignment each). Their pool rows come LAST in the
    // layout and are not materialised at all unless a dense pass over them is asked for (sparse_probe = 0,
    // the popcount variants): a column of short lists costs its listed positions, not 8 KiB per block.
    std::vector<uint64_t> col_elems((size_t)max_id + 2, 0);
    for (uint64_t b = 0; b < n_blocks; ++b)
        if (block_kind[b] == 0) col_elems[block_id[b]] += block_n[b];
    std::vector<uint8_t> probe_c((size_t)max_id + 2, 0);
    {
        uint64_t total = 0;
        for (uint32_t c = 0; c <= max_id; ++c) {
            const uint64_t n_l = n_list_col[c];
            if (n_l >= 2 && n_l <= 65535 && col_elems[c] > 0 && total + col_elems[c] + 64 < (1ull << 32) - (1u << 20)) {  // (the probe kernel's element indices run up to 12 x 8192 past an item's end)
                probe_c[c] = 1;
                total += col_elems[c] + 64;
            }
        }
    }
    // Layout of a column: its bitmap blocks, then — on the next multiple of 512 rows — its list blocks (the kind
    // dispatch of storm.c:618-656, once per block: list x list pairs go to the probe kernel, every pair with a
    // bitmap block to the matrix cores, which then need the bitmap rows alone as A rows). Columns made of lists
    // only come last: they need no pool rows at all.
    std::vector<uint64_t> start((size_t)max_id + 2, 0), list0((size_t)max_id + 2, 0), col_end((size_t)max_id + 2, 0);
    uint64_t run = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t c = 0; c <= max_id; ++c) {
            const bool lists_only = probe_c[c] && n_list_col[c] == per_col[c];
            if (per_col[c] && (int)lists_only == pass) {
                const uint64_t n_b = per_col[c] - n_list_col[c];
                start[c] = run;
                list0[c] = n_b && n_list_col[c] ? (run + n_b + 511) / 512 * 512 : run + n_b;
                col_end[c] = list0[c] + n_list_col[c];
                run = (col_end[c] + 511) / 512 * 512;  // next column starts on a 512-row A tile
            }
        }
        if (pass == 0) s->pool_rows_ready = run;
    }
    for (uint32_t c = 0; c <= max_id; ++c)
        if (per_col[c]) {
            s->cols.push_back({start[c], col_end[c]});
            s->col_list0.push_back(list0[c]);
            const uint64_t nl = n_list_col[c], nb = per_col[c] - nl;
            s->census[0] += nl * (nl - (nl != 0)) / 2;
            s->census[1] += nl * nb;
            s->census[2] += nb * (nb - (nb != 0)) / 2;
            s->census[3] += 1;
        }
    s->n_pool_rows = run;
    if (s->n_pool_rows >= (1ull << 32) - 512) {
        set_error("sparse_create: block pool too large");
        return STORM_HIP_EINVAL;
    }

    lap("validate");
    // ---- pool row of every block (rows are visited in order => row order inside a column)
    std::vector<uint32_t> list_row, dense_row, list_len;
    std::vector<uint64_t> list_blk, dense_blk;   // the blocks behind those rows
    {
        std::vector<uint64_t> next_bitmap(start), next_list(list0);
        for (uint64_t b = 0; b < n_blocks; ++b) {
            const uint32_t pr = (uint32_t)(block_kind[b] == 0 ? next_list[block_id[b]]++ : next_bitmap[block_id[b]]++);
            if (block_kind[b] == 0) {
                if (block_n[b] && pr < s->pool_rows_ready) {
                    list_row.push_back(pr);
                    list_blk.push_back(b);
                    list_len.push_back(block_n[b]);
                }
            } else {
                dense_row.push_back(pr);
                dense_blk.push_back(b);
            }
        }
    }

    lap("pool rows");
    // ---- probe data (K4): columns whose blocks are all lists and that have at most 65535 rows. Per column
    //      and octant (8192 positions of the block) the listed positions in row order.
    // The host only decides WHERE every run of a list goes (a walk over the block records); the elements are laid
    // out by the device from the raw lists (probe_fill_kernel, probe_deal_kernel).
    size_t n_probe_elems = 0;
    constexpr uint32_t kNoBlock = 0xffffffffu;
    std::vector<uint64_t> probe_blocks;  // the list blocks of probe columns, in row order
    std::vector<uint32_t> run_end;       // per probe block and octant: end of the octant's run inside the list
    std::vector<uint32_t> run_dst;       // ... and where the run starts in the element arrays
    std::vector<uint32_t> block_local;   // the block's row inside its column's list rows
    std::vector<uint32_t> atoms;         // {first element, end} of every atom of the far stream
    // device temporaries of the build: released on every way out
    struct DevTemps {
        uint32_t *lrow = nullptr, *llen = nullptr, *tags = nullptr, *rend = nullptr, *rdst = nullptr;
        uint32_t *atoms = nullptr, *bad = nullptr, *pllen = nullptr;
        uint64_t *loff = nullptr, *ploff = nullptr;
        uint16_t *lists = nullptr, *pos_tmp = nullptr;
        uint64_t* ltable = nullptr;
        storm_hip_ctx_t* ctx = nullptr;
        ~DevTemps() {
            // (the build has waited for its last kernel; a hipFree waits for the device once more and costs ~0.2 ms: put off)
            for (void* p : {(void*)lrow, (void*)loff, (void*)llen, (void*)lists, (void*)tags, (void*)rend, (void*)rdst,
                            (void*)atoms, (void*)bad, (void*)pllen, (void*)ploff, (void*)pos_tmp, (void*)ltable})
                if (p) {
                    if (ctx) ctx->deferred_free.push_back(p);
                    else (void)hipFree(p);
                }
        }
    } dt;
    dt.ctx = ctx;
    drain_deferred(ctx, false);   // (whatever an earlier build left behind)
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    Stager stager(ctx);
    if (int rc0 = stager.init()) return rc0;
    // ---- the raw lists the device needs — of the probe columns (element layout) and of the list blocks that own a
    //      pool row (expanded there) — go up FIRST, block after block: the device then finds where the octants' runs
    //      end inside every list of a probe column, which is all the host's layout below needs to know of them
    for (uint64_t b = 0; b < n_blocks; ++b)
        if (block_kind[b] == 0 && probe_c[block_id[b]]) probe_blocks.push_back(b);
    std::vector<uint64_t> dev_off(n_blocks + 1, ~0ull);
    uint64_t n_list_elems = 0;
    const uint16_t* lists_view = nullptr;   // where the kernels below find the raw lists: dt.lists, or the stage's one chunk
    {
        std::vector<uint8_t> wanted(n_blocks, 0);
        for (uint64_t b : probe_blocks) wanted[b] = 1;
        for (uint64_t b : list_blk) wanted[b] = 1;
        // [r6] lists the caller staged while it built the container (storm_hip_stage_add_list: a token that is not ~0) are
        // gathered from the stage's chunks by one kernel; the others travel now, through the pinned ring. Both kinds lie in
        // dt.lists in block order: first the run from the host, behind it the staged ones.
        std::vector<std::pair<const void*, size_t>> run;
        std::vector<uint64_t> staged;   // blocks whose list is in the stage
        const uint64_t list_space = stage ? stage->lbase + stage->lfill : 0;
        for (uint64_t b = 0; b < n_blocks; ++b)
            if (wanted[b] && block_n[b]) {
                if (stage && stage_token && stage_token[b] != ~0ull) {
                    if ((stage_token[b] & 1u) || stage_token[b] + (uint64_t)block_n[b] * 2u > list_space) {
                        set_error("sparse_create: block %llu carries a list token outside the stage", (unsigned long long)b);
                        return STORM_HIP_EINVAL;
                    }
                    staged.push_back(b);
                    continue;
                }
                dev_off[b] = n_list_elems;
                n_list_elems += block_n[b];
                run.emplace_back(block_ptr[b], (size_t)block_n[b] * sizeof(uint16_t));
            }
        // ... unless EVERY wanted list is staged and the stage's lists fit one chunk (up to 32 M positions: every sparse
        // load of c4): the builder then reads them where they lie — dev_off = the token, no table, no gather, no copy
        if (run.empty() && !staged.empty() && list_space <= storm_hip_stage_s::kListChunk) {
            if (int rc0 = stage_send_lists(ctx, stage)) return rc0;   // what is still in the ring
            for (uint64_t b : staged) dev_off[b] = stage_token[b] / 2u;
            lists_view = reinterpret_cast<const uint16_t*>(stage->lchunks[0]);
            staged.clear();
        }
        std::vector<uint64_t> ltable;
        ltable.reserve(3 * staged.size());
        for (uint64_t b : staged) {
            dev_off[b] = n_list_elems;
            ltable.insert(ltable.end(), {n_list_elems, stage_token[b], (uint64_t)block_n[b]});
            n_list_elems += block_n[b];
        }
        if (n_list_elems) {
            if (hipMalloc(reinterpret_cast<void**>(&dt.lists), n_list_elems * sizeof(uint16_t)) != hipSuccess) {
                set_error("sparse_create: hipMalloc of %llu bytes for the lists failed",
                          (unsigned long long)(n_list_elems * sizeof(uint16_t)));
                return STORM_HIP_ENOMEM;
            }
            if (int rc0 = stager.send_run(reinterpret_cast<uint8_t*>(dt.lists), run)) return rc0;
        }
        if (!staged.empty())
            if (int rc0 = stage_gather_lists(ctx, stage, ltable, dt.lists, &dt.ltable)) return rc0;
        if (!lists_view) lists_view = dt.lists;
    }
    lap("lists -> device");
    std::vector<uint64_t> ploff;
    std::vector<uint32_t> pllen;
    if (!probe_blocks.empty()) {
        ploff.reserve(probe_blocks.size());
        pllen.reserve(probe_blocks.size());
        for (uint64_t b : probe_blocks) {
            ploff.push_back(block_n[b] ? dev_off[b] : 0);
            pllen.push_back(block_n[b]);
        }
        run_end.assign(probe_blocks.size() * kProbeOctants, 0);
        if (int rc0 = upload(ctx, &dt.ploff, ploff.data(), ploff.size())) return rc0;
        if (int rc0 = upload(ctx, &dt.pllen, pllen.data(), pllen.size())) return rc0;
        STORM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dt.rend), run_end.size() * sizeof(uint32_t)));
        const uint32_t n_threads = (uint32_t)run_end.size();
        hipLaunchKernelGGL(probe_run_end_kernel, dim3((n_threads + kThreads - 1) / kThreads), dim3(kThreads), 0, ctx->stream,
                           lists_view, dt.ploff, dt.pllen, (uint32_t)probe_blocks.size(), dt.rend);
        STORM_HIP_TRY(hipGetLastError());
        STORM_HIP_TRY(hipMemcpyAsync(run_end.data(), dt.rend, run_end.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        // a list that is not ascending may give ends that run backwards: refused here, before they are laid out
        // (ascending ends that disagree with the values are caught by probe_fill_kernel)
        for (size_t pb = 0; pb < probe_blocks.size(); ++pb) {
            uint32_t from = 0;
            for (uint32_t o = 0; o < kProbeOctants; ++o) {
                const uint32_t end = run_end[pb * kProbeOctants + o];
                if (end < from || end > pllen[pb]) {
                    set_error("sparse_create: a list block is not strictly ascending");
                    return STORM_HIP_EINVAL;
                }
                from = end;
            }
        }
        lap("run ends on the device");
    }
    {
        std::vector<int64_t> col_entry((size_t)max_id + 2, -1);  // column id -> index into s->cols
        s->col_probe.assign(s->cols.size(), 0);
        s->col_avg_len.assign(s->cols.size(), 0);
        uint64_t total = 0;
        size_t entry = 0;
        for (uint32_t c = 0; c <= max_id; ++c) {
            if (!per_col[c]) continue;
            col_entry[c] = (int64_t)entry;
            if (probe_c[c]) {
                s->col_probe[entry] = 1;
                s->col_avg_len[entry] = (uint32_t)(col_elems[c] / n_list_col[c]);
                total += col_elems[c] + 64;
            }
            ++entry;
        }
        if (total > 0) {
            // count per (probe column, octant), then lay the octants out one after the other, each on a
            // 16-byte boundary of the uint16 position array
            const size_t n_e = s->cols.size();
            std::vector<uint64_t> oct_count(n_e * kProbeOctants, 0), oct_base(n_e * kProbeOctants, 0);
            run_dst.assign(probe_blocks.size() * kProbeOctants, 0);
            block_local.assign(probe_blocks.size(), kNoBlock);
            for (size_t pb = 0; pb < probe_blocks.size(); ++pb) {
                const size_t e = (size_t)col_entry[block_id[probe_blocks[pb]]];
                uint32_t from = 0;
                for (uint32_t o = 0; o < kProbeOctants; ++o) {
                    oct_count[e * kProbeOctants + o] += run_end[pb * kProbeOctants + o] - from;
                    from = run_end[pb * kProbeOctants + o];
                }
            }
            uint64_t at = 0;
            for (size_t i = 0; i < oct_count.size(); ++i) {
                at = (at + 7) & ~7ull;
                oct_base[i] = at;
                at += oct_count[i];
            }
            n_probe_elems = (size_t)at + 8;
            for (size_t i = 0; i < oct_count.size(); ++i)
                if (oct_count[i])
                    s->probe_regions.push_back({(uint32_t)oct_base[i], (uint32_t)(oct_base[i] + oct_count[i]),
                                                (uint32_t)s->col_list0[i / kProbeOctants], (uint32_t)(i % kProbeOctants)});
            // rows are visited in order: element offset of every row, per octant
            std::vector<uint64_t> cursor(oct_base);
            std::vector<uint64_t> next((size_t)max_id + 2, 0);  // list blocks of the column seen so far
            std::vector<std::vector<uint32_t>> row_start(n_e * kProbeOctants);
            for (size_t e = 0; e < n_e; ++e)
                if (s->col_probe[e])
                    for (uint32_t o = 0; o < kProbeOctants; ++o)
                        row_start[e * kProbeOctants + o].reserve((size_t)(s->cols[e].r1 - s->col_list0[e]) + 1);
            {
                size_t pb = 0;
                for (uint64_t b = 0; b < n_blocks; ++b) {
                    if (block_kind[b] != 0) continue;
                    const uint32_t c = block_id[b];
                    const uint64_t local = next[c]++;
                    const int64_t e = col_entry[c];
                    if (e < 0 || !s->col_probe[(size_t)e]) continue;
                    block_local[pb] = (uint32_t)local;
                    uint32_t from = 0;
                    for (uint32_t o = 0; o < kProbeOctants; ++o) {
                        const size_t i = (size_t)e * kProbeOctants + o;
                        row_start[i].push_back((uint32_t)cursor[i]);
                        run_dst[pb * kProbeOctants + o] = (uint32_t)cursor[i];
                        cursor[i] += run_end[pb * kProbeOctants + o] - from;
                        from = run_end[pb * kProbeOctants + o];
                    }
                    ++pb;
                }
            }
            lap("layout (prefix sums, row starts)");
            // atoms of the far stream (the elements of one group of kProbeRows rows in one octant): the device deals
            // the positions of every atom by LDS bank (probe_deal_kernel; why: see there)
            for (size_t i = 0; i < row_start.size(); ++i) {
                const std::vector<uint32_t>& rs = row_start[i];
                const uint32_t end = (uint32_t)(oct_base[i] + oct_count[i]);
                for (size_t k = 0; k < rs.size(); k += kProbeRows) {
                    atoms.push_back(rs[k]);
                    atoms.push_back(k + kProbeRows < rs.size() ? rs[k + kProbeRows] : end);
                }
            }
            // far work of all groups -> positions per item: about 4096 items over all probe columns, between
            // 2^15 and 2^21 positions each (an item zeroes and scatters its 128 KiB table first)
            uint64_t far_work = 0;
            for (size_t i = 0; i < row_start.size(); ++i) {
                const std::vector<uint32_t>& rs = row_start[i];
                if (rs.empty()) continue;
                const uint32_t n_c = (uint32_t)rs.size();
                const uint64_t end = oct_base[i] + oct_count[i];
                for (uint32_t a0 = 0; a0 < n_c; a0 += kProbeRows) {
                    const uint32_t a1 = std::min(a0 + kProbeRows, n_c);
                    far_work += end - (a1 < n_c ? rs[a1] : end);
                }
            }
            // Chunks of the far stream are the SAME for every group of a (column, octant) stream: a fixed grid of runs
            // of atoms, ~far_work / 2048 positions each and at most 2^23. A group needs the rest of the chunk its own
            // atom lies in and every later chunk. The items that read one chunk are a FAMILY; a family goes to one XCD,
            // its items one after the other, so that the XCD's 32 CUs work through the same positions at the same time
            // and HBM delivers them once (the kernel fetched 16 GB per launch at c4's 20971 draws — every group
            // streaming its own 4 MiB chunks through an L2 that hit 9 % of the time — and ran at the HBM rate with the
            // LDS half idle). Every item pays for its group's histogram, so fewer, longer items win as long as the
            // 512 workgroup slots stay filled: / 4096 and 2^21 until round 4; / 2048 is 8 - 18 % faster at every c4
            // load and 2^23 another 7 % at 30000 draws (profiles/r04_h_sparse_probe.txt).
            const uint64_t kProbeChunk =
                std::min<uint64_t>(1u << 23, std::max<uint64_t>(1u << 15, far_work / 2048)) & ~7ull;
            struct Family {
                uint64_t work = 0;
                std::vector<storm_hip_sparse_s::ProbeItemHost> items;
            };
            std::vector<Family> families;
            for (size_t i = 0; i < row_start.size(); ++i) {
                std::vector<uint32_t>& rs = row_start[i];
                if (rs.empty()) continue;
                const uint32_t n_c = (uint32_t)rs.size();
                rs.push_back((uint32_t)(oct_base[i] + oct_count[i]));  // end of the octant
                const uint32_t e = (uint32_t)(i / kProbeOctants);
                // atoms t = 0 .. n_atoms - 1: rows [128 t, 128 t + 128); chunk grid over atoms 1 .. (atom 0 is nobody's far part)
                const uint32_t n_atoms = (n_c + kProbeRows - 1) / kProbeRows;
                auto atom_start = [&](uint32_t t) { return rs[std::min(t * (uint32_t)kProbeRows, n_c)]; };
                std::vector<uint32_t> chunk_first;  // first atom of every chunk, + n_atoms
                for (uint32_t t = 1; t < n_atoms;) {
                    chunk_first.push_back(t);
                    const uint64_t from = atom_start(t);
                    ++t;
                    while (t < n_atoms && atom_start(t) - from < kProbeChunk) ++t;
                }
                chunk_first.push_back(n_atoms);
                const size_t fam0 = families.size();
                families.resize(fam0 + chunk_first.size());  // one per chunk (+ one for groups without a far part)
                for (uint32_t g = 0; g < n_atoms; ++g) {
                    const uint32_t a0 = g * kProbeRows, a1 = std::min(a0 + (uint32_t)kProbeRows, n_c);
                    if (rs[a1] == rs[a0]) continue;  // no listed position of the A rows in this octant
                    // first item of the group: its own rows among themselves (n range set) + the rest of the chunk the
                    // next atom lies in; one item per later chunk
                    if (g + 1 >= n_atoms) {  // the last group has nobody behind it
                        families[fam0 + chunk_first.size() - 1].items.push_back({rs[a0], rs[a1], rs[a0], rs[a1], rs[a1], rs[a1], a0, e});
                        families[fam0 + chunk_first.size() - 1].work += 8192;
                        continue;
                    }
                    size_t c = 0;
                    while (c + 1 < chunk_first.size() && chunk_first[c + 1] <= g + 1) ++c;
                    bool first = true;
                    for (; c + 1 < chunk_first.size(); ++c) {
                        const uint32_t b0 = first ? rs[a1] : atom_start(chunk_first[c]);
                        const uint32_t b1 = atom_start(chunk_first[c + 1]);
                        const uint32_t n0 = first ? rs[a0] : 0u, n1 = first ? rs[a1] : 0u;
                        if (n1 > n0 || b1 > b0) {
                            families[fam0 + c].items.push_back({rs[a0], rs[a1], n0, n1, std::min(b0, b1), b1, a0, e});
                            families[fam0 + c].work += (uint64_t)(b1 - std::min(b0, b1)) + 8192;
                        }
                        first = false;
                    }
                }
            }
            // families to XCDs: heaviest first onto the lightest queue; block b of the launch runs on XCD b % 8
            // (observed; speed only), so the launch order takes one item of every queue in turn
            std::vector<size_t> order(families.size());
            for (size_t f = 0; f < order.size(); ++f) order[f] = f;
            std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return families[x].work > families[y].work; });
            std::vector<std::vector<storm_hip_sparse_s::ProbeItemHost>> queue(8);
            uint64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t f : order) {
                if (families[f].items.empty()) continue;
                int q = 0;
                for (int x = 1; x < 8; ++x)
                    if (load[x] < load[q]) q = x;
                queue[q].insert(queue[q].end(), families[f].items.begin(), families[f].items.end());
                load[q] += families[f].work;
            }
            size_t longest = 0;
            for (int x = 0; x < 8; ++x) longest = std::max(longest, queue[x].size());
            for (size_t pos = 0; pos < longest; ++pos)
                for (int x = 0; x < 8; ++x)
                    if (pos < queue[x].size()) s->probe_items.push_back(queue[x][pos]);
            // [r6] the same work in bundles of kFatGroups groups (probe_lists_fat_kernel): the same chunk grid (a bundle
            // needs the rest of the chunk its successor atom lies in and every later chunk), the same families and queues
            {
                struct FatFamily {
                    uint64_t work = 0;
                    std::vector<storm_hip_sparse_s::ProbeFatItemHost> items;
                };
                std::vector<FatFamily> fat;
                for (size_t i = 0; i < row_start.size(); ++i) {
                    const std::vector<uint32_t>& rs = row_start[i];   // (ends with the end of the octant since the loop above)
                    if (rs.size() < 2) continue;
                    const uint32_t n_c = (uint32_t)rs.size() - 1u;
                    const uint32_t e = (uint32_t)(i / kProbeOctants);
                    const uint32_t n_atoms = (n_c + kProbeRows - 1) / kProbeRows;
                    auto atom_start = [&](uint32_t t) { return rs[std::min(t * (uint32_t)kProbeRows, n_c)]; };
                    std::vector<uint32_t> chunk_first;
                    for (uint32_t t = 1; t < n_atoms;) {
                        chunk_first.push_back(t);
                        const uint64_t from = atom_start(t);
                        ++t;
                        while (t < n_atoms && atom_start(t) - from < kProbeChunk) ++t;
                    }
                    chunk_first.push_back(n_atoms);
                    const size_t fam0 = fat.size();
                    fat.resize(fam0 + chunk_first.size());
                    for (uint32_t g0 = 0; g0 < n_atoms; g0 += kFatGroups) {
                        const uint32_t g1 = std::min(g0 + kFatGroups, n_atoms);
                        storm_hip_sparse_s::ProbeFatItemHost it{};
                        for (uint32_t k = 0; k <= kFatGroups; ++k) it.at[k] = atom_start(std::min(g0 + k, g1));
                        it.col = e;
                        const uint32_t own = it.at[kFatGroups] - it.at[0];
                        if (own == 0) continue;   // no listed position of the bundle's rows in this octant
                        if (g1 >= n_atoms) {      // the last bundle has nobody behind it
                            it.b_begin = it.b_end = it.at[kFatGroups];
                            it.first = 1;
                            fat[fam0 + chunk_first.size() - 1].items.push_back(it);
                            fat[fam0 + chunk_first.size() - 1].work += 8192u * (g1 - g0);
                            continue;
                        }
                        size_t c = 0;
                        while (c + 1 < chunk_first.size() && chunk_first[c + 1] <= g1) ++c;
                        bool first = true;
                        for (; c + 1 < chunk_first.size(); ++c) {
                            const uint32_t b1 = atom_start(chunk_first[c + 1]);
                            const uint32_t b0 = std::min(first ? it.at[kFatGroups] : atom_start(chunk_first[c]), b1);
                            if (first || b1 > b0) {
                                it.b_begin = b0;
                                it.b_end = b1;
                                it.first = first ? 1u : 0u;
                                fat[fam0 + c].items.push_back(it);
                                fat[fam0 + c].work += (uint64_t)(b1 - b0) * (g1 - g0) + 8192u * (g1 - g0);
                            }
                            first = false;
                        }
                    }
                }
                std::vector<size_t> forder(fat.size());
                for (size_t f = 0; f < forder.size(); ++f) forder[f] = f;
                std::stable_sort(forder.begin(), forder.end(), [&](size_t x, size_t y) { return fat[x].work > fat[y].work; });
                std::vector<std::vector<storm_hip_sparse_s::ProbeFatItemHost>> fqueue(8);
                uint64_t fload[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (size_t f : forder) {
                    if (fat[f].items.empty()) continue;
                    int q = 0;
                    for (int x = 1; x < 8; ++x)
                        if (fload[x] < fload[q]) q = x;
                    fqueue[q].insert(fqueue[q].end(), fat[f].items.begin(), fat[f].items.end());
                    fload[q] += fat[f].work;
                }
                size_t flongest = 0;
                for (int x = 0; x < 8; ++x) flongest = std::max(flongest, fqueue[x].size());
                for (size_t pos = 0; pos < flongest; ++pos)
                    for (int x = 0; x < 8; ++x)
                        if (pos < fqueue[x].size()) s->probe_fat_items.push_back(fqueue[x][pos]);
            }
        }
    }

    lap("items");
    // ---- device side: pool rows, raw lists through the pinned ring, element layout by kernels ----
    int rc = STORM_HIP_OK;
    do {
        // Pool rows are 8 KiB of bits; their PITCH gets a 512-byte chunk more (K2b fetches 64-byte pieces of 64
        // consecutive rows: at a power-of-two pitch they fall into a handful of memory channels — the dense matrix's
        // finding, pitch_pad_chunks; option k2_matrix_pad; K2b over the c4 pool 5.95 -> 5.82 ms).
        // The pad words stay zero: the kernels that take the pitch for the row length multiply zeros there.
        s->pitch = kBlockWords + pitch_pad_chunks(ctx->k2_matrix_pad, kBlockWords) * kChunkWords;
        const size_t row_bytes = s->pitch * sizeof(uint64_t);
        const size_t pool_bytes = (s->pool_rows_ready + 512) * row_bytes;
        if (hipMalloc(reinterpret_cast<void**>(&s->d_pool), pool_bytes) != hipSuccess) {
            set_error("sparse_create: hipMalloc of %zu bytes for the block pool failed",
                      pool_bytes);
            rc = STORM_HIP_ENOMEM;
            break;
        }
        if (hipMemsetAsync(s->d_pool, 0, pool_bytes, ctx->stream) != hipSuccess) {
            rc = STORM_HIP_EHIP;
            break;
        }
        // bitmap-kind blocks: straight into their pool rows — in pool-row order the blocks of a column are
        // consecutive rows, so a run of them is ONE contiguous destination (no staging copy on the device, no
        // placement kernel; the words may sit at any alignment on the host: a serialized stream)
        // [r6] ... unless every one of them already lies on the device (the block stage filled while the rows were
        // added): one kernel gathers the pool rows from the stage's chunks
        bool gathered = false;
        if (!dense_row.empty() && stage && stage_token) {
            bool all = true;
            for (uint64_t b : dense_blk) all = all && stage_token[b] < stage->n_blocks;
            if (all) {
                if ((rc = stage_send(ctx, stage)) != STORM_HIP_OK) break;   // the blocks still in the ring
                std::vector<uint64_t> table(2 * dense_row.size());
                for (size_t k = 0; k < dense_row.size(); ++k) table[2 * k] = dense_row[k], table[2 * k + 1] = stage_token[dense_blk[k]];
                uint64_t* d_table = nullptr;
                if (stage->d_chunk_table) (void)hipFree(stage->d_chunk_table);
                stage->d_chunk_table = nullptr;
                if (hipMalloc(reinterpret_cast<void**>(&stage->d_chunk_table), stage->chunks.size() * sizeof(uint8_t*)) != hipSuccess ||
                    hipMalloc(reinterpret_cast<void**>(&d_table), table.size() * sizeof(uint64_t)) != hipSuccess) {
                    set_error("sparse_create: hipMalloc of the gather table failed");
                    rc = STORM_HIP_ENOMEM;
                    break;
                }
                if (hipMemcpyAsync(stage->d_chunk_table, stage->chunks.data(), stage->chunks.size() * sizeof(uint8_t*),
                                   hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                    upload_bytes(ctx, d_table, table.data(), table.size() * sizeof(uint64_t)) != STORM_HIP_OK) {
                    (void)hipFree(d_table);
                    rc = STORM_HIP_EHIP;
                    break;
                }
                hipLaunchKernelGGL(gather_staged_kernel, dim3((uint32_t)dense_row.size()), dim3(256), 0, ctx->stream,
                                   stage->d_chunk_table, d_table, s->d_pool, (uint64_t)s->pitch);
                const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;   // `table` is pageable
                (void)hipFree(d_table);
                if (!ok) { rc = STORM_HIP_EHIP; break; }
                gathered = true;
            }
        }
        if (!dense_row.empty() && !gathered) {
            std::vector<uint32_t> order(dense_row.size());
            for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return dense_row[x] < dense_row[y]; });
            std::vector<const void*> run;
            size_t i = 0;
            while (i < order.size() && rc == STORM_HIP_OK) {
                const uint32_t row0 = dense_row[order[i]];
                run.clear();
                size_t j = i;
                while (j < order.size() && dense_row[order[j]] == row0 + (j - i)) {
                    run.push_back(block_ptr[dense_blk[order[j]]]);
                    ++j;
                }
                rc = stager.send_rows(reinterpret_cast<uint8_t*>(s->d_pool) + (size_t)row0 * row_bytes, run,
                                      (size_t)kBlockWords * sizeof(uint64_t), row_bytes);
                i = j;
            }
            if (rc != STORM_HIP_OK) break;
        }
        lap("bitmaps -> pool rows");
        // list-kind blocks that own a pool row (mixed columns, columns the probe kernel cannot take): expanded there
        std::vector<uint64_t> loff;
        if (!list_row.empty()) {
            loff.reserve(list_blk.size());
            for (uint64_t b : list_blk) loff.push_back(dev_off[b]);
            if ((rc = upload(ctx, &dt.lrow, list_row.data(), list_row.size())) ||
                (rc = upload(ctx, &dt.loff, loff.data(), loff.size())) ||
                (rc = upload(ctx, &dt.llen, list_len.data(), list_len.size())))
                break;
            hipLaunchKernelGGL(expand_lists_kernel, dim3((uint32_t)list_row.size()),
                               dim3(kThreads), 0, ctx->stream, s->d_pool, s->pitch, dt.lrow, dt.loff, dt.llen,
                               lists_view);
            if (hipGetLastError() != hipSuccess) { rc = STORM_HIP_EHIP; break; }
        }
        // probe columns: element layout on the device
        std::vector<uint32_t> tags;
        uint32_t bad = 0;
        if (n_probe_elems) {
            if (hipMalloc(reinterpret_cast<void**>(&s->d_probe_elems), n_probe_elems * sizeof(uint32_t)) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&s->d_probe_pos16), n_probe_elems * sizeof(uint16_t)) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&dt.pos_tmp), n_probe_elems * sizeof(uint16_t)) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&dt.bad), sizeof(uint32_t)) != hipSuccess) {
                set_error("sparse_create: hipMalloc of the probe element arrays (%zu elements) failed", n_probe_elems);
                rc = STORM_HIP_ENOMEM;
                break;
            }
            // (the alignment gaps behind every octant read as zero)
            if (hipMemsetAsync(s->d_probe_elems, 0, n_probe_elems * sizeof(uint32_t), ctx->stream) != hipSuccess ||
                hipMemsetAsync(s->d_probe_pos16, 0, n_probe_elems * sizeof(uint16_t), ctx->stream) != hipSuccess ||
                hipMemsetAsync(dt.bad, 0, sizeof(uint32_t), ctx->stream) != hipSuccess) {
                rc = STORM_HIP_EHIP;
                break;
            }
            tags.reserve(probe_blocks.size());
            for (size_t pb = 0; pb < probe_blocks.size(); ++pb) tags.push_back(block_local[pb] << 16);
            // (the lists' offsets, their lengths and the run ends are on the device since the start)
            if ((rc = upload(ctx, &dt.tags, tags.data(), tags.size())) ||
                (rc = upload(ctx, &dt.rdst, run_dst.data(), run_dst.size())) ||
                (rc = upload(ctx, &dt.atoms, atoms.data(), atoms.size())))
                break;
            if (!probe_blocks.empty())
                hipLaunchKernelGGL(probe_fill_kernel, dim3((uint32_t)probe_blocks.size()), dim3(kThreads), 0, ctx->stream,
                                   lists_view, dt.ploff, dt.pllen, dt.tags, dt.rend, dt.rdst, s->d_probe_elems, dt.pos_tmp, dt.bad);
            if (!atoms.empty())
                hipLaunchKernelGGL(probe_deal_kernel, dim3((uint32_t)(atoms.size() / 2)), dim3(kThreads), 0, ctx->stream,
                                   dt.atoms, dt.pos_tmp, s->d_probe_pos16);
            if (hipGetLastError() != hipSuccess ||
                hipMemcpyAsync(&bad, dt.bad, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
                rc = STORM_HIP_EHIP;
                break;
            }
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = STORM_HIP_EHIP; break; }
        if (bad) {
            set_error("sparse_create: a list block is not strictly ascending");
            rc = STORM_HIP_EINVAL;
        }
        lap("element layout on the device");
    } while (0);
    if (rc == STORM_HIP_EHIP) set_error("sparse_create: HIP failure: %s",
                                        hipGetErrorString(hipGetLastError()));
    if (rc != STORM_HIP_OK) return rc;
    *out = owner.release();
    return STORM_HIP_OK;
}

// The pool rows of the probe columns too (a dense pass over them was asked for): a larger pool, the rows that
// exist copied over, the lists of the probe columns expanded from their probe elements.
static int ensure_full_pool(storm_hip_ctx_t* ctx, storm_hip_sparse_t* s) {
    if (s->pool_rows_ready >= s->n_pool_rows) return STORM_HIP_OK;
    const size_t old_bytes = (size_t)s->pool_rows_ready * s->pitch * sizeof(uint64_t);
    const size_t new_bytes = (size_t)(s->n_pool_rows + 512) * s->pitch * sizeof(uint64_t);
    uint64_t* np = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&np), new_bytes) != hipSuccess) {
        set_error("sparse: hipMalloc of %zu bytes for the pool rows of the list columns failed", new_bytes);
        return STORM_HIP_ENOMEM;
    }
    uint32_t* d_regions = nullptr;
    int rc = STORM_HIP_OK;
    do {
        if (hipMemsetAsync(reinterpret_cast<uint8_t*>(np) + old_bytes, 0, new_bytes - old_bytes, ctx->stream) != hipSuccess ||
            (old_bytes && hipMemcpyAsync(np, s->d_pool, old_bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)) {
            rc = STORM_HIP_EHIP;
            break;
        }
        if (!s->probe_regions.empty()) {
            static_assert(sizeof(storm_hip_sparse_s::ProbeRegion) == 16, "four uint32 per region");
            if ((rc = upload(ctx, &d_regions, reinterpret_cast<const uint32_t*>(s->probe_regions.data()),
                             s->probe_regions.size() * 4)))
                break;
            hipLaunchKernelGGL(expand_probe_kernel, dim3((uint32_t)s->probe_regions.size()), dim3(kThreads), 0,
                               ctx->stream, np, s->pitch, s->d_probe_elems, d_regions);
            if (hipGetLastError() != hipSuccess) { rc = STORM_HIP_EHIP; break; }
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = STORM_HIP_EHIP;
    } while (0);
    (void)hipFree(d_regions);
    if (rc != STORM_HIP_OK) {
        (void)hipFree(np);
        if (rc == STORM_HIP_EHIP) set_error("sparse: building the pool rows of the list columns failed");
        return rc;
    }
    (void)hipFree(s->d_pool);
    s->d_pool = np;
    s->pool_rows_ready = s->n_pool_rows;
    return STORM_HIP_OK;
}

// The rows of a STORM_t as a dense bit matrix on the device (what the per-pair output runs on: the tile kernels
// write popcount(row_i OP row_j) for every pair, and a row pair of the reference — the block-id merge and the 4-way
// kind dispatch of storm.c:790-814, :618-656 — is exactly that over the rows' bits). Blocks travel as they lie in
// the containers: packed into the pinned ring with a table of pieces, unpacked by one kernel per chunk (bitmap
// blocks copied to word 1024 * id of their row, list blocks OR-ed in bit by bit).
struct DensePiece {
    uint32_t src;       // byte offset inside the chunk (16-byte aligned)
    uint32_t n;         // list length, or kDenseBitmap
    uint64_t dst_word;  // first word of the block inside the matrix
};
constexpr uint32_t kDenseBitmap = 0xffffffffu;

__global__ __launch_bounds__(kThreads) void densify_kernel(const uint8_t* __restrict__ chunk,
                                                           const DensePiece* __restrict__ pieces, uint32_t n_pieces,
                                                           uint64_t* __restrict__ matrix) {
    for (uint32_t p = blockIdx.x; p < n_pieces; p += gridDim.x) {
        const DensePiece pc = pieces[p];
        if (pc.n == kDenseBitmap) {
            const uint4* src = reinterpret_cast<const uint4*>(chunk + pc.src);
            uint4* dst = reinterpret_cast<uint4*>(matrix + pc.dst_word);   // rows and blocks are multiples of 16 bytes
            for (uint32_t i = threadIdx.x; i < kBlockWords / 2; i += kThreads) dst[i] = src[i];
        } else {
            const uint16_t* list = reinterpret_cast<const uint16_t*>(chunk + pc.src);
            uint32_t* dst = reinterpret_cast<uint32_t*>(matrix + pc.dst_word);
            for (uint32_t i = threadIdx.x; i < pc.n; i += kThreads) {
                const uint32_t v = list[i];
                atomicOr(dst + (v >> 5), 1u << (v & 31u));
            }
        }
    }
}

extern "C" {

int storm_hip_sparse_create(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                            const uint64_t* row_block_offset, const uint32_t* block_id,
                            const uint8_t* block_kind, const uint64_t* block_data_offset,
                            const uint32_t* block_n, const uint16_t* list_pool,
                            uint64_t list_pool_len, const uint64_t* bitmap_pool,
                            uint64_t bitmap_pool_words, storm_hip_sparse_t** out) {
    try {
        if (n_blocks > 0 && (!block_kind || !block_data_offset || !block_n)) {
            set_error("sparse_create: NULL descriptor array");
            return STORM_HIP_EINVAL;
        }
        // the flat pools as per-block pointers (storm_hip_sparse_create_blocks does the work)
        std::vector<const void*> ptr(n_blocks, nullptr);
        for (uint64_t b = 0; b < n_blocks; ++b) {
            if (block_kind[b] == 0) {
                if (block_data_offset[b] + block_n[b] > list_pool_len || (block_n[b] && !list_pool)) {
                    set_error("sparse_create: list block outside the list pool");
                    return STORM_HIP_EINVAL;
                }
                ptr[b] = list_pool ? list_pool + block_data_offset[b] : nullptr;
            } else {
                if (block_data_offset[b] + kBlockWords > bitmap_pool_words || !bitmap_pool) {
                    set_error("sparse_create: bitmap block outside the bitmap pool");
                    return STORM_HIP_EINVAL;
                }
                ptr[b] = bitmap_pool + block_data_offset[b];
            }
        }
        return build_arena(ctx, n_rows, n_blocks, row_block_offset, block_id, block_kind, block_n, ptr.data(), out);
    } catch (const std::exception& e) {
        set_error("sparse_create: %s", e.what());
        return STORM_HIP_ENOMEM;
    }
}

// The same from per-block pointers into the caller's own containers: block_ptr[b] = the block's sorted uint16 list
// (block_n[b] entries, kind 0) or its 1024 words (kind 1, any alignment). Nothing is flattened on the host: the
// library walks the block headers, ships the raw lists and bitmaps through a pinned ring and lays the elements
// out on the device.
int storm_hip_sparse_create_blocks(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                   const uint64_t* row_block_offset, const uint32_t* block_id,
                                   const uint8_t* block_kind, const uint32_t* block_n,
                                   const void* const* block_ptr, storm_hip_sparse_t** out) {
    try {
        return build_arena(ctx, n_rows, n_blocks, row_block_offset, block_id, block_kind, block_n, block_ptr, out);
    } catch (const std::exception& e) {
        set_error("sparse_create_blocks: %s", e.what());
        return STORM_HIP_ENOMEM;
    }
}

// [r6] The same with the bitmap blocks already on the device: token[b] = what storm_hip_stage_add returned for block b
// (any value >= the stage's count, e.g. ~0: not staged). When every bitmap block is staged the pool rows are gathered from
// the stage and nothing of them crosses the bus; otherwise this is storm_hip_sparse_create_blocks (block_ptr must be
// valid either way).
int storm_hip_sparse_create_blocks_staged(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                          const uint64_t* row_block_offset, const uint32_t* block_id,
                                          const uint8_t* block_kind, const uint32_t* block_n, const void* const* block_ptr,
                                          storm_hip_stage_t* stage, const uint64_t* token, storm_hip_sparse_t** out) {
    try {
        return build_arena(ctx, n_rows, n_blocks, row_block_offset, block_id, block_kind, block_n, block_ptr, out, stage, token);
    } catch (const std::exception& e) {
        set_error("sparse_create_blocks_staged: %s", e.what());
        return STORM_HIP_ENOMEM;
    }
}

int storm_hip_matrix_create_from_blocks(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                                   const uint64_t* row_block_offset, const uint32_t* block_id,
                                                   const uint8_t* block_kind, const uint32_t* block_n,
                                                   const void* const* block_ptr, storm_hip_matrix_t** out) {
    return guarded("storm_hip_matrix_create_from_blocks", [&]() -> int {
        if (!ctx || !out) {
            set_error("matrix_create_from_blocks: NULL context or output");
            return STORM_HIP_EINVAL;
        }
        *out = nullptr;
        if (n_rows == 0 || !row_block_offset || row_block_offset[0] != 0 || row_block_offset[n_rows] != n_blocks ||
            (n_blocks > 0 && (!block_id || !block_kind || !block_n || !block_ptr))) {
            set_error("matrix_create_from_blocks: no rows, NULL descriptor array or a CSR that does not end at n_blocks");
            return STORM_HIP_EINVAL;
        }
        uint32_t max_id = 0;
        for (uint64_t r = 0; r < n_rows; ++r) {
            if (row_block_offset[r] > row_block_offset[r + 1] || row_block_offset[r + 1] > n_blocks) {
                set_error("matrix_create_from_blocks: row_block_offset is not a CSR over %llu blocks",
                          (unsigned long long)n_blocks);
                return STORM_HIP_EINVAL;
            }
            for (uint64_t b = row_block_offset[r]; b < row_block_offset[r + 1]; ++b) {
                if ((b > row_block_offset[r] && block_id[b] <= block_id[b - 1]) || block_kind[b] > 1 ||
                    (block_kind[b] == 0 ? (block_n[b] > 65536u || (block_n[b] && (!block_ptr[b] || ((uintptr_t)block_ptr[b] & 1))))
                                        : !block_ptr[b])) {
                    set_error("matrix_create_from_blocks: block %llu of row %llu: ids not ascending, unknown kind or no data",
                              (unsigned long long)b, (unsigned long long)r);
                    return STORM_HIP_EINVAL;
                }
                max_id = std::max(max_id, block_id[b]);
            }
        }
        // the tile kernels address a row with 32-bit DMA offsets: 2^25 bits (storm_hip_pairw_matrix)
        if (n_blocks > 0 && max_id >= (1u << 25) / 65536u) {
            set_error("matrix_create_from_blocks: block id %u: rows of the dense form stop at 2^25 bits", max_id);
            return STORM_HIP_EINVAL;
        }
        storm_hip_matrix_t* m = nullptr;
        if (int rc = storm_hip_matrix_create(ctx, n_rows, (max_id + 1u) * kBlockWords, &m)) return rc;
        m->sparse_origin = true;   // (the output kernel is chosen by this: storm_hip_internal.h, k2_tile_shape)
        struct MatrixDeleter {
            storm_hip_ctx_t* ctx;
            void operator()(storm_hip_matrix_t* x) const { storm_hip_matrix_destroy(ctx, x); }
        };
        std::unique_ptr<storm_hip_matrix_t, MatrixDeleter> owner(m, MatrixDeleter{ctx});
        Stager stager(ctx);
        if (int rc = stager.init()) return rc;
        // chunk = [data | piece table]: up to 7 MiB of blocks and 64 Ki pieces per buffer of the ring
        constexpr size_t kData = Stager::kBuf - (1u << 20), kMaxPieces = (1u << 20) / sizeof(DensePiece);
        uint8_t* d_chunks = nullptr;
        STORM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_chunks), Stager::kBuf * Stager::kBufs));
        std::unique_ptr<uint8_t, void (*)(uint8_t*)> chunks_owner(d_chunks, [](uint8_t* p) { (void)hipFree(p); });
        std::vector<DensePiece> table;
        std::vector<Piece> pieces;
        table.reserve(kMaxPieces);
        pieces.reserve(kMaxPieces + 1);
        size_t fill = 0;
        int slot = 0;
        auto flush = [&]() -> int {
            if (table.empty()) return STORM_HIP_OK;
            pieces.push_back({table.data(), table.size() * sizeof(DensePiece), fill});  // (fill is 16-byte aligned)
            uint8_t* d_base = d_chunks + (size_t)slot * Stager::kBuf;
            if (int rc = stager.send(d_base, pieces.data(), pieces.size(), fill + table.size() * sizeof(DensePiece))) return rc;
            const uint32_t n = (uint32_t)table.size();
            densify_kernel<<<std::min<uint32_t>(n, 4096u), kThreads, 0, ctx->stream>>>(
                d_base, reinterpret_cast<const DensePiece*>(d_base + fill), n, m->d);
            STORM_HIP_TRY(hipGetLastError());
            slot = (slot + 1) % Stager::kBufs;
            table.clear();
            pieces.clear();
            fill = 0;
            return STORM_HIP_OK;
        };
        for (uint64_t r = 0; r < n_rows; ++r)
            for (uint64_t b = row_block_offset[r]; b < row_block_offset[r + 1]; ++b) {
                const bool bitmap = block_kind[b] != 0;
                if (!bitmap && block_n[b] == 0) continue;
                const size_t bytes = bitmap ? kBlockWords * sizeof(uint64_t) : (size_t)block_n[b] * sizeof(uint16_t);
                if (fill + bytes > kData || table.size() == kMaxPieces)
                    if (int rc = flush()) return rc;
                table.push_back({(uint32_t)fill, bitmap ? kDenseBitmap : block_n[b],
                                 r * m->stride_words + (uint64_t)block_id[b] * kBlockWords});
                pieces.push_back({block_ptr[b], bytes, fill});
                fill = (fill + bytes + 15) & ~(size_t)15;
            }
        if (int rc = flush()) return rc;
        STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));  // the ring and the chunk buffers go away here
        *out = owner.release();
        return STORM_HIP_OK;
    });
}

// Arena straight from a serialized STORM_t (byte layout: STORM_serialize in storm.h / storm_host.c;
// sizes as reference storm.c:372-394, :963-973). Only the headers are walked on the host
// (O(blocks)); the payload bytes are uploaded as they are and unpacked by the device.
int storm_hip_sparse_create_serialized(storm_hip_ctx_t* ctx, const void* buf, uint64_t n_bytes,
                                       storm_hip_sparse_t** out) {
    if (!ctx || !out || !buf) {
        set_error("sparse_create_serialized: NULL argument");
        return STORM_HIP_EINVAL;
    }
    *out = nullptr;
    try {
        const uint8_t* p = static_cast<const uint8_t*>(buf);
        auto u32_at = [&](uint64_t off) { uint32_t v; memcpy(&v, p + off, 4); return v; };
        const char* bad = "sparse_create_serialized: truncated or malformed stream";
        if (n_bytes < 8 || (n_bytes & 1) || u32_at(4) != kSerialMagic) { set_error("%s", bad); return STORM_HIP_EINVAL; }
        const uint64_t n_rows = u32_at(0);
        // (same validity rules as STORM_deserialize, storm_host.c) a row costs at least its 12 header bytes:
        // the row count is bounded by the stream before it sizes anything
        if (n_rows > (n_bytes - 8) / 12) { set_error("%s", bad); return STORM_HIP_EINVAL; }
        std::vector<uint64_t> row_off(n_rows + 1, 0);
        std::vector<const void*> ptrs;
        std::vector<uint32_t> ids, lens;
        std::vector<uint8_t> kinds;
        uint64_t at = 8;
        for (uint64_t r = 0; r < n_rows; ++r) {
            if (at + 12 > n_bytes) { set_error("%s", bad); return STORM_HIP_EINVAL; }
            const uint32_t nb = u32_at(at);
            at += 12;
            if (at + 4ull * nb > n_bytes) { set_error("%s", bad); return STORM_HIP_EINVAL; }
            const uint64_t ids_at = at;  // block_ids[] (repeated in the block headers)
            at += 4ull * nb;
            for (uint32_t b = 0; b < nb; ++b) {
                if (at + 16 > n_bytes) { set_error("%s", bad); return STORM_HIP_EINVAL; }
                const uint32_t n_bitmap = u32_at(at), n_bits_set = u32_at(at + 4);
                const uint32_t w2 = u32_at(at + 8), id = u32_at(at + 12);
                const uint32_t n_scalar = w2 & 0x7fffffffu, has_list = w2 >> 31;
                at += 16;
                const uint64_t words = 8ull * n_bitmap, list = has_list ? 2ull * n_scalar : 0;
                if ((n_bitmap != 0 && n_bitmap != kBlockWords) || n_scalar > 65536u ||
                    id != u32_at(ids_at + 4ull * b) || (b && id <= u32_at(ids_at + 4ull * (b - 1))) ||
                    at + words + list > n_bytes || (!n_bitmap && has_list && n_bits_set != n_scalar)) {
                    set_error("%s", bad);
                    return STORM_HIP_EINVAL;
                }
                if (!n_bitmap && has_list) {  // a list is strictly ascending: the probe kernel counts every element
                    const uint8_t* l = p + at + words;
                    uint16_t prev = 0, cur = 0;
                    for (uint32_t k = 0; k < n_scalar; ++k, prev = cur) {
                        memcpy(&cur, l + 2ull * k, 2);
                        if (k && cur <= prev) { set_error("%s", bad); return STORM_HIP_EINVAL; }
                    }
                }
                ids.push_back(id);
                if (n_bitmap) {  // bitmap kind (storm.c:745-749: a block is one kind or the other)
                    kinds.push_back(1); ptrs.push_back(p + at); lens.push_back(0);
                } else {
                    kinds.push_back(0); ptrs.push_back(p + at + words); lens.push_back(has_list ? n_scalar : 0);
                }
                at += words + list;
            }
            row_off[r + 1] = ids.size();
        }
        if (at != n_bytes) { set_error("%s", bad); return STORM_HIP_EINVAL; }
        // (the stream is 2-byte aligned, so is every list in it; the bitmap words are copied bytewise)
        if ((uintptr_t)p & 1) { set_error("sparse_create_serialized: the stream must be 2-byte aligned"); return STORM_HIP_EINVAL; }
        return build_arena(ctx, n_rows, ids.size(), row_off.data(), ids.data(), kinds.data(), lens.data(),
                           ptrs.data(), out);
    } catch (const std::exception& e) {
        set_error("sparse_create_serialized: %s", e.what());
        return STORM_HIP_ENOMEM;
    }
}

void storm_hip_sparse_destroy(storm_hip_ctx_t* ctx, storm_hip_sparse_t* s) {
    if (!s) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (s->d_pool) (void)hipFree(s->d_pool);
    if (s->d_segs) (void)hipFree(s->d_segs);
    if (s->d_probe_elems) (void)hipFree(s->d_probe_elems);
    if (s->d_probe_pos16) (void)hipFree(s->d_probe_pos16);
    if (s->d_probe_items) (void)hipFree(s->d_probe_items);
    delete s;
}

int storm_hip_pairw_sparse_end(storm_hip_ctx_t* ctx, uint64_t* h_total) {
    if (!ctx || !h_total) {
        set_error("pairw_sparse_end: NULL argument");
        return STORM_HIP_EINVAL;
    }
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    return fetch_result_word(ctx, h_total);
}

int storm_hip_pairw_sparse(storm_hip_ctx_t* ctx, const storm_hip_sparse_t* cs,
                           uint32_t shard_rank, uint32_t shard_count, uint64_t* h_total) {
    if (!h_total) {
        set_error("pairw_sparse: NULL argument");
        return STORM_HIP_EINVAL;
    }
    if (int rc = storm_hip_pairw_sparse_begin(ctx, cs, shard_rank, shard_count)) return rc;
    return storm_hip_pairw_sparse_end(ctx, h_total);
}

// Launches this shard's share into the context's result word; _end fetches it.
int storm_hip_pairw_sparse_begin(storm_hip_ctx_t* ctx, const storm_hip_sparse_t* cs,
                                 uint32_t shard_rank, uint32_t shard_count) {
    return guarded("storm_hip_pairw_sparse_begin", [&]() -> int {
    if (!ctx || !cs) {
        set_error("pairw_sparse: NULL argument");
        return STORM_HIP_EINVAL;
    }
    if (shard_count == 0 || shard_rank >= shard_count) {
        set_error("pairw_sparse: shard %u of %u is not valid", shard_rank, shard_count);
        return STORM_HIP_EINVAL;
    }
    storm_hip_sparse_t* s = const_cast<storm_hip_sparse_t*>(cs);
    STORM_HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->deferred_free.empty() || !ctx->deferred_host_free.empty()) drain_deferred(ctx, true);
    uint64_t* const d_result = result_target(ctx);   // the mailbox, or ctx->d_scalar
    memcpy(ctx->sparse_census, s->census, sizeof(s->census));
    memset(ctx->pass_report, 0, sizeof(ctx->pass_report));
    // Matrix-core path when the columns are big enough to fill the chip (same rule as the dense
    // container): every column is an independent all-pairs problem over its pool rows.
    uint64_t widest = 0;
    for (const RowRange& c : s->cols) widest = std::max(widest, c.r1 - c.r0);
    int variant = ctx->variant;
    if (variant < 0) variant = widest >= 64 ? 4 : 2;
    ctx->variant_used = variant;
    if (variant >= 3) {
        // K4: columns of short lists go to the probe kernel ("sparse_probe": -1 = when the mean list
        // has at most 1000 positions — measured at c4 (profiles/r02_q_sparse_probe.jsonl): 47x faster
        // than the dense path at 13 positions per list, 15x at 65, 3.0x at 393, 1.7x at 655, 1.1x at 1012 —,
        // 1 = every eligible column, 0 = never); what it counts lands in the same slots the strips'
        // fold sums up
        std::vector<uint8_t> use_probe(s->cols.size(), 0);
        if (ctx->sparse_probe != 0 && s->d_probe_elems)
            for (size_t e = 0; e < s->cols.size(); ++e)
                use_probe[e] = s->col_probe[e];
        {
            // [r6] one group per workgroup (probe_lists_kernel) unless the option asks for bundles of four
            // (probe_bundle 4: probe_lists_fat_kernel — measured 5 - 25 % slower at every c4 load); both lists hold the same work
            const int bundle = ctx->probe_bundle == 4 ? (int)kFatGroups : 1;
            uint64_t key = 1469598103934665603ull ^ ((uint64_t)shard_rank << 32 | shard_count) ^ ((uint64_t)bundle << 56);
            for (uint8_t u : use_probe) key = (key ^ u) * 1099511628211ull;
            if (key != s->probe_key) {
                static_assert(sizeof(ProbeFatItem) >= sizeof(ProbeItem), "one device buffer for either list");
                std::vector<ProbeItem> mine;
                std::vector<ProbeFatItem> fat;
                uint32_t cols_used = 0;
                for (size_t e = 0; e < use_probe.size(); ++e) cols_used += use_probe[e];
                if (bundle == 1) {
                    for (const auto& pi : s->probe_items)
                        if (use_probe[pi.col])
                            mine.push_back({pi.a_begin, pi.a_end, pi.n_begin, pi.n_end, pi.b_begin, pi.b_end, pi.a0});
                } else {
                    for (const auto& pi : s->probe_fat_items)
                        if (use_probe[pi.col])
                            fat.push_back({{pi.at[0], pi.at[1], pi.at[2], pi.at[3], pi.at[4]}, pi.b_begin, pi.b_end, pi.first});
                }
                const size_t n_mine = bundle == 1 ? mine.size() : fat.size();
                if (n_mine > s->probe_items_capacity) {
                    if (s->d_probe_items) STORM_HIP_TRY(hipFree(s->d_probe_items));
                    s->d_probe_items = nullptr;
                    s->probe_items_capacity = 0;
                    STORM_HIP_TRY(hipMalloc(&s->d_probe_items, std::max<size_t>(n_mine, 1024) * sizeof(ProbeFatItem)));
                    s->probe_items_capacity = std::max<size_t>(n_mine, 1024);
                }
                if (n_mine) {
                    STORM_HIP_TRY(hipMemcpyAsync(s->d_probe_items, bundle == 1 ? (const void*)mine.data() : (const void*)fat.data(),
                                                 n_mine * (bundle == 1 ? sizeof(ProbeItem) : sizeof(ProbeFatItem)),
                                                 hipMemcpyHostToDevice, ctx->stream));
                    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
                }
                // shard r of G takes items r, r + G, ...: the grid covers ceil((n - r) / G) of them
                s->n_probe_launch = n_mine > shard_rank ? (uint32_t)((n_mine - shard_rank + shard_count - 1) / shard_count) : 0u;
                s->n_probe_cols_launch = cols_used;
                // lookups of this shard's items: a streamed far position against every group of the item + the own rows'
                // elements (against their own group and, in a bundle, the groups in front of it)
                s->probe_lookups_launch = 0;
                if (bundle == 1) {
                    for (size_t k = shard_rank; k < mine.size(); k += shard_count)
                        s->probe_lookups_launch += (uint64_t)(mine[k].b_end - mine[k].b_begin) + (mine[k].n_end - mine[k].n_begin);
                } else {
                    for (size_t k = shard_rank; k < fat.size(); k += shard_count) {
                        uint32_t groups = 0;
                        for (uint32_t g = 0; g < kFatGroups; ++g) {
                            const uint32_t len = fat[k].at[g + 1] - fat[k].at[g];
                            groups += len != 0;
                            if (fat[k].first) s->probe_lookups_launch += (uint64_t)len * (g + 1u);
                        }
                        s->probe_lookups_launch += (uint64_t)(fat[k].b_end - fat[k].b_begin) * groups;
                    }
                }
                s->probe_bundle_launch = bundle;
                s->probe_key = key;
            }
        }
        bool probe_folds = false;
        std::vector<RowRange> ranges;
        uint64_t rows_needed = 0;
        for (size_t e = 0; e < s->cols.size(); ++e) {
            RowRange rg = s->cols[e];
            if (use_probe[e]) {
                // the list blocks pair with each other in the probe kernel: the matrix cores take the pairs with a
                // bitmap block — the column's bitmap rows as A rows, against each other and the lists behind them
                if (s->col_list0[e] == rg.r0) continue;  // no bitmap block
                rg.a_end = s->col_list0[e];
            }
            if (rg.r1 - rg.r0 > 1) {
                ranges.push_back(rg);
                rows_needed = std::max(rows_needed, rg.r1);
            }
        }
        if (s->n_probe_launch > 0) {
            ctx->pass_report[0] |= STORM_HIP_RAN_LIST_PROBE;
            ctx->pass_report[2] += s->probe_lookups_launch;
            ctx->pass_report[3] = kProbeRows;
            // Threads per workgroup by the length of the items (c4, ms per call at 104 / 524 / 1048 / 2097 / 5242 / 10485 /
            // 20971 draws per row, one box, profiles/r04_h_sparse_probe.txt):
            //     256 threads, 8 workgroups per CU   0.034 0.047 0.063 0.095 0.193 0.495 1.554
            //     512, 4                             0.039 0.051 0.064 0.096 0.183 0.341 0.917
            //    1024, 2                             0.055 0.065 0.078 0.106 0.189 0.328 0.664
            // Short items are all fixed cost — the item record, the group's positions and the far positions are three
            // dependent trips to memory, then the histogram — and what covers that is workgroups per CU; long ones want
            // the workgroups of a CU in step on the same chunk of the stream (its L2 lines are read once per XCD).
            const uint64_t per_item = s->probe_lookups_launch / s->n_probe_launch;
            const int threads = s->probe_bundle_launch != 1 ? (per_item < 1500000u ? 512 : kProbeThreads)
                                : per_item < 400000u        ? 256
                                : per_item < 1500000u       ? 512
                                                            : kProbeThreads;
            // lists only (no pool rows to multiply) and a short launch: the probe kernel folds inside the launch
            // (option k2_fold_inline as for the strips; the slot words' 48-bit sums hold any total below 2^47 / 4096 x 256)
            probe_folds = ranges.empty() && ctx->k2_fold_inline != 0 && s->n_probe_launch <= 16384u &&
                          s->probe_lookups_launch < (1ull << 38);
            unsigned long long* fold_out = probe_folds ? reinterpret_cast<unsigned long long*>(d_result) : nullptr;
#define STORM_PROBE_LAUNCH(T)                                                                                          \
    hipLaunchKernelGGL(probe_lists_kernel<T>, dim3(s->n_probe_launch), dim3(T), 0, ctx->stream, s->d_probe_elems,       \
                       s->d_probe_pos16, static_cast<const ProbeItem*>(s->d_probe_items), shard_count, shard_rank, ctx->d_slots, \
                       fold_out, 256u)
#define STORM_PROBE_FAT_LAUNCH(T)                                                                                      \
    hipLaunchKernelGGL(probe_lists_fat_kernel<T>, dim3(s->n_probe_launch), dim3(T), 0, ctx->stream, s->d_probe_pos16,     \
                       static_cast<const ProbeFatItem*>(s->d_probe_items), shard_count, shard_rank, ctx->d_slots, fold_out, 256u)
            if (s->probe_bundle_launch != 1) {
                if (threads == 512) STORM_PROBE_FAT_LAUNCH(512);
                else STORM_PROBE_FAT_LAUNCH(kProbeThreads);
            } else if (threads == 256) STORM_PROBE_LAUNCH(256);
            else if (threads == 512) STORM_PROBE_LAUNCH(512);
            else STORM_PROBE_LAUNCH(kProbeThreads);
#undef STORM_PROBE_FAT_LAUNCH
#undef STORM_PROBE_LAUNCH
            STORM_HIP_TRY(hipGetLastError());
        }
        if (probe_folds) {   // the total is in d_scalar already: nothing to multiply, nothing to fold
            ctx->last_info[0] = 0;
            ctx->last_info[3] = s->n_probe_cols_launch;
            ctx->k2_operands_used = 5;
            return STORM_HIP_OK;
        }
        if (rows_needed > s->pool_rows_ready)
            if (int rc = ensure_full_pool(ctx, s)) return rc;
        // (only the rows the dense pass multiplies are expanded: the probe columns lie behind them)
        const uint64_t rows_dst = (rows_needed + 511) / 512 * 512;
        // The strips on bit operands (K2b) multiply the pool rows as they are: no FP4 shadow of the pool (2.6 GB
        // at c4), no expansion pass. The rows behind a column's last one up to the next multiple of 512 are zero
        // in the pool (columns start on multiples of 512 and nothing writes between them).
        if (variant == 4 && strip_operands_of(ctx) == 5 && ctx->k2_debug == 0 && !ctx->k2_persistent &&
            rows_dst <= s->pool_rows_ready + 512) {
            if (int rc = launch_pairw_bits_ranges(ctx, reinterpret_cast<const uint8_t*>(s->d_pool), s->pitch * 8ull,
                                                  ranges, kBlockWords / 4u, shard_rank, shard_count,
                                                  d_result, s->n_probe_launch > 0))
                return rc;
            ctx->last_info[3] = s->n_probe_cols_launch;
            return STORM_HIP_OK;
        }
        if (int rc = launch_pairw_mfma_ranges(ctx, s->d_pool, s->pitch, s->pool_rows_ready + 512,
                                              std::max<uint64_t>(rows_dst, 512), ranges,
                                              shard_rank, shard_count, variant == 5 ? 2 : variant == 4 ? 1 : 0,
                                              d_result))
            return rc;
        ctx->last_info[3] = s->n_probe_cols_launch;  // block columns counted by the list-probe kernel
        return STORM_HIP_OK;
    }
    if (int rc = ensure_full_pool(ctx, s)) return rc;  // the popcount kernel walks every column's pool rows
    const uint32_t seg_len = (uint32_t)ctx->seg_rows;
    if (!s->d_segs || s->seg_rank != shard_rank || s->seg_count != shard_count ||
        s->seg_len != seg_len) {
        // upper triangle of every block column; shard = every shard_count-th segment
        std::vector<Seg> full, diag, mine;
        for (const RowRange& col : s->cols) {
            const uint64_t lo = col.r0, hi = col.r1;
            for (uint64_t a0 = lo; a0 < hi; a0 += kABlockRows) {
                const uint32_t a_end = (uint32_t)std::min<uint64_t>(a0 + kABlockRows, hi);
                if (a_end - a0 > 1) diag.push_back({(uint32_t)a0, a_end, (uint32_t)a0, a_end});
                for (uint64_t j = a0 + kABlockRows; j < hi; j += seg_len)
                    full.push_back({(uint32_t)a0, a_end, (uint32_t)j,
                                    (uint32_t)std::min<uint64_t>(j + seg_len, hi)});
            }
        }
        for (size_t i = shard_rank; i < full.size(); i += shard_count) mine.push_back(full[i]);
        for (size_t i = shard_rank; i < diag.size(); i += shard_count) mine.push_back(diag[i]);
        if (s->d_segs) STORM_HIP_TRY(hipFree(s->d_segs));
        s->d_segs = nullptr;
        s->seg_row_sum = 0;
        for (const Seg& g : mine) s->seg_row_sum += g.j_hi - g.j_lo;
        if (!mine.empty()) {
            STORM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_segs),
                                    mine.size() * sizeof(Seg)));
            STORM_HIP_TRY(hipMemcpyAsync(s->d_segs, mine.data(), mine.size() * sizeof(Seg),
                                         hipMemcpyHostToDevice, ctx->stream));
            STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        s->n_segs = (uint32_t)mine.size();
        s->seg_rank = shard_rank;
        s->seg_count = shard_count;
        s->seg_len = seg_len;
    }
    if (int rc = launch_pairw_segments(ctx, s->d_pool, s->pitch, s->d_segs, s->n_segs,
                                       s->seg_row_sum,
                                       d_result))
        return rc;
    return STORM_HIP_OK;
    });
}

int storm_hip_sparse_last_census(storm_hip_ctx_t* ctx, uint64_t out[4]) {
    if (!ctx || !out) return STORM_HIP_EINVAL;
    memcpy(out, ctx->sparse_census, sizeof(ctx->sparse_census));
    return STORM_HIP_OK;
}

}  // extern "C"
