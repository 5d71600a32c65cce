// tile_ring.hip — prototype + structure probe of the round-5 materialised-output kernel (K2r), standalone.
//
// One 256 x 256 output tile per workgroup over all of k, v_mfma_f32_16x16x128_f8f6f4 (the shape that draws the least
// power per FLOP: profiles/r05_a_clock_power.jsonl), BOTH operands read as fragments from FP4 images that the workgroup
// builds once in the LDS (K2b's image, swizzle and rotate-and-mask codes: every product 1.0, no block scales):
//   workgroup : 8 waves, two per SIMD; wave (wa, wb) owns A rows 128 wa .. + 127 against B rows 64 wb .. + 63:
//               8 x 4 blocks of 16 x 16 = 128 accumulator registers; per k-step (128 bits, one class of a 512-bit
//               chunk) 32 MFMAs against 12 fragment reads (0.375 per MFMA).
//   stage     : one class of one chunk = one k-step: images of 256 A rows + 256 B rows x 64 B = 32 KiB; ring of 4
//               (128 KiB). Stage s multiplies image s FROM REGISTERS (its fragments were read during stage s - 1),
//               reads the fragments of image s + 1 (A: progressively, fa[m] is reloaded right behind the last MFMA of
//               row m and has a whole stage to arrive; B: into the other of two register sets) and writes the wave's
//               share of image s + 3 (4 of the 32 pieces of 16 rows x 64 B: rotate, mask, ds_write_b128).
//   bits      : straight from global memory into registers (global_load_dwordx4, issued five stages ahead of their
//               first use; plain loads cost the issuing wave next to nothing beside MFMAs), two chunks in registers.
//   sync      : ONE s_barrier per stage; what it orders is two stages old by then (an image written in stage s is first
//               read in stage s + 2), so no wave waits for an LDS operation at the barrier — only for the other waves.
//   lgkmcnt   : LDS operations of a wave complete in order and every stage issues the same 16 of them, so "fa[m] of the
//               previous stage has landed" is lgkmcnt(15) in front of every row.
// Modes (argv): check = compare sampled entries against a CPU popcount; time = ms per launch over the triangle of an
// N x M matrix, in-kernel clock stamps; variants by template: barrier / no barrier (upper bound, wrong results).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/tile_ring tools/probes/tile_ring.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <utility>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kTile = 256;
constexpr uint32_t kRowBytes = 128;                 // two k-steps of 64 B per image row
constexpr uint32_t kPairBytes = kTile * kRowBytes;  // 32 KiB: image pair (two ring slots) of one operand
constexpr uint32_t kOperandBytes = 2 * kPairBytes;  // 64 KiB: the four ring slots of one operand
constexpr uint32_t kMask = 0x22222222u;

__device__ unsigned long long g_clock[4];

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int ROT>
__device__ __forceinline__ v4i inflate(v4i w) {
    v4i e;
    if constexpr (ROT == 0) {
        e = w & (int)kMask;
    } else {
        e.x = (int)(__builtin_amdgcn_alignbit((uint32_t)w.x, (uint32_t)w.x, ROT) & kMask);
        e.y = (int)(__builtin_amdgcn_alignbit((uint32_t)w.y, (uint32_t)w.y, ROT) & kMask);
        e.z = (int)(__builtin_amdgcn_alignbit((uint32_t)w.z, (uint32_t)w.z, ROT) & kMask);
        e.w = (int)(__builtin_amdgcn_alignbit((uint32_t)w.w, (uint32_t)w.w, ROT) & kMask);
    }
    return e;
}
template <int C>
__device__ __forceinline__ int inflate1(int w) {
    constexpr int ROT = (C + 31) % 32;
    if constexpr (ROT == 0) return w & (int)kMask;
    else return (int)(__builtin_amdgcn_alignbit((uint32_t)w, (uint32_t)w, ROT) & kMask);
}
// class c of a dword = bits 4 n + c: rotate right by c - 1 (mod 32) puts them on bit 1 of every nibble = E2M1 1.0
template <int C>
__device__ __forceinline__ v4i inflate_class(v4i w) {
    return inflate<(C + 31) % 32>(w);
}

template <int OFF>
__device__ __forceinline__ void lds_read128(v4i& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_write128(uint32_t addr, const v4i& e) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(e), "n"(OFF) : "memory");
}
__device__ __forceinline__ void keep(const v4i& v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void mfma_agpr(v4f& acc, const v4i& a, const v4i& b) {
    asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+a"(acc) : "v"(a), "v"(b));
}

struct TileItem {
    uint32_t a_row0, b_row0;
};

#define LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// kAblate (timing and power probes; results wrong by construction): bit 0 = no s_barrier, bit 1 = one-operation inflation (the
// piece ANDed with a class mask where it stands: one-hot nibbles, wrong values — the cost of a pre-permuted bit layout), bit 2 = no image stores, bit 3 = no fragment reads (the registers keep the prologue's fragments),
// bit 4 = no loads of the bits beyond the prologue.
// Correct variants: bit 5 = a barrier in front of the EVEN stages only (ring of 4, images three stages ahead: a wave may run one
// stage ahead of another), bit 6 = accumulators in AGPRs (MFMAs as inline asm)
template <int kAblate, int kSched>
__global__ __launch_bounds__(512, 2) void tile_ring_kernel(const uint8_t* __restrict__ X, uint64_t pitch,
                                                           uint32_t n_chunks, const TileItem* __restrict__ items,
                                                           uint32_t* __restrict__ out, uint64_t ld, int stamp) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * kOperandBytes];
    const uint64_t ck_t0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wa = wave & 1u, wb = wave >> 1;
    const TileItem it = items[blockIdx.x];

    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&lds[0];
    // fragment (block b, half h) of an operand's slot pair: row 16 b + (lane & 15), 16-byte slot (4 h + (lane >> 4)) ^ swizzle
    const uint32_t fswz = ((lane & 15u) >> 1) & 7u;
    const uint32_t fr0 = lds_base + (lane & 15u) * kRowBytes + (((lane >> 4) ^ fswz) * 16u);
    const uint32_t fa_addr[2] = {fr0 + wa * 8u * 2048u, (fr0 ^ 64u) + wa * 8u * 2048u};
    const uint32_t fb_addr[2] = {fr0 + kOperandBytes + wb * 4u * 2048u, (fr0 ^ 64u) + kOperandBytes + wb * 4u * 2048u};
    // the wave's pieces: rows 32 wave + 16 p + pr of the A tile (p = 0, 1) and of the B tile (p = 2, 3); lanes 4 g .. 4 g + 3
    // hold the four 16-byte quarters of one row, the two rows of an 8-lane group are 8 apart (conflict-free stores)
    const uint32_t pr = (lane >> 3) + 8u * ((lane >> 2) & 1u);
    const uint32_t wr0 = lds_base + (wave * 32u + pr) * kRowBytes + (((lane & 3u) ^ ((pr >> 1) & 7u)) * 16u);
    const uint32_t wr_addr[2][2] = {{wr0, wr0 ^ 64u}, {wr0 + kOperandBytes, (wr0 ^ 64u) + kOperandBytes}};
    const uint32_t voff = (wave * 32u + pr) * (uint32_t)pitch + (lane & 3u) * 16u;
    const uint8_t* a_bits = X + (uint64_t)it.a_row0 * pitch;
    const uint8_t* b_bits = X + (uint64_t)it.b_row0 * pitch;

    v4i bits[2][4];   // [chunk parity][piece]
    auto load_chunk = [&](int par, uint32_t chunk) __attribute__((always_inline)) {
        const uint32_t c = min(chunk, n_chunks - 1u);   // (beyond the last chunk: re-read it, never consumed)
        const uint8_t* pa = a_bits + (uint64_t)c * 64u + voff;
        const uint8_t* pb = b_bits + (uint64_t)c * 64u + voff;
        bits[par][0] = *reinterpret_cast<const v4i*>(pa);
        bits[par][1] = *reinterpret_cast<const v4i*>(pa + 16u * pitch);
        bits[par][2] = *reinterpret_cast<const v4i*>(pb);
        bits[par][3] = *reinterpret_cast<const v4i*>(pb + 16u * pitch);
    };
    // piece p of class C into ring slot S (pair S >> 1, half S & 1)
    auto write_piece = [&](auto pc, auto cc, auto sc, const v4i& w) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value, C = decltype(cc)::value, S = decltype(sc)::value;
        const v4i e = inflate_class<C>(w);
        lds_write128<(S >> 1) * kPairBytes + (p & 1) * 16 * kRowBytes>(wr_addr[p >> 1][S & 1], e);
    };

    v4f acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = v4f{};
    v4i fa[8], fb[2][4];

    // ---- prologue: chunks 0 and 1 in registers, images 0, 1, 2 (classes 0, 1, 2 of chunk 0) written, fragments of image 0 read
    load_chunk(0, 0u);
    load_chunk(1, 1u);
    VMCNT(4);
    static_for<3>([&](auto cc) __attribute__((always_inline)) {
        static_for<4>([&](auto pc) __attribute__((always_inline)) { write_piece(pc, cc, cc, bits[0][decltype(pc)::value]); });
    });
    LGKM(0);
    __builtin_amdgcn_s_barrier();
    static_for<8>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        lds_read128<m * 2048>(fa[m], fa_addr[0]);
    });
    static_for<4>([&](auto nc) __attribute__((always_inline)) {
        constexpr int n = decltype(nc)::value;
        lds_read128<n * 2048>(fb[0][n], fb_addr[0]);
    });
    LGKM(0);
    if constexpr (kAblate & 8) {
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[1][n] = fb[0][n];
    }

    // ---- stages, eight per trip (two chunks): stage j multiplies image j (slot j & 3), reads image j + 1, writes image j + 3
    const uint32_t n_trips = n_chunks / 2u;   // (host: n_chunks even)
    for (uint32_t trip = 0; trip < n_trips; ++trip) {
        static_for<8>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            constexpr int S1 = (j + 1) & 3;                 // slot read
            constexpr int T = j + 3, S3 = T & 3, C3 = T & 3, PAR3 = (T >> 2) & 1;   // image written: class, chunk parity
            constexpr int cur = j & 1;                      // fb set multiplied in this stage
            if constexpr ((j & 3) == 1) VMCNT(4);           // first use of the chunk loaded a chunk ago (the newest 4 loads may fly)
            // where the wave's four pieces are inflated and stored (kSched): 0 = piece p in row 2 p; 1 = pieces 2 m, 2 m + 1 in
            // rows m = 0, 1 (a burst at the head of the stage); 2 = the same burst in rows 4, 5
            auto row_of = [](int p) constexpr { return (kSched == 0 || kSched >= 3) ? 2 * p : kSched == 1 ? p / 2 : 4 + p / 2; };
            constexpr int W3 = (row_of(0) == 3) + (row_of(1) == 3) + (row_of(2) == 3) + (row_of(3) == 3);
            constexpr int WL = (row_of(0) > 3) + (row_of(1) > 3) + (row_of(2) > 3) + (row_of(3) > 3);
            constexpr int kRow0Wait = kSched == 4 ? 8 : W3 + 4 + WL;   // LDS operations issued behind fb'[3] (row 3) up to the next stage's row 0
            v4i e[4];
            if constexpr (kSched == 3) {
                // LDS operations BETWEEN the MFMAs of a row instead of behind it (2-3 in a burst at the end of every row, from
                // all eight waves at once behind the barrier, stalled at the LDS's queue: SQ_WAIT_INST_LDS 14 % of wave cycles):
                // row m = MFMA 0, [fb'[m]], MFMA 1, MFMA 2, [store of piece m / 2, even m], MFMA 3, fa[m] reload
                static_for<8>([&](auto mc) __attribute__((always_inline)) {
                    constexpr int m = decltype(mc)::value;
                    constexpr int p = m >> 1;
                    if constexpr (m == 0) LGKM(7); else LGKM(15);
                    __builtin_amdgcn_sched_barrier(0);
                    auto mul = [&](auto nc) __attribute__((always_inline)) {
                        constexpr int n = decltype(nc)::value;
                        acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                            v8i{fa[m].x, fa[m].y, fa[m].z, fa[m].w, 0, 0, 0, 0},
                            v8i{fb[cur][n].x, fb[cur][n].y, fb[cur][n].z, fb[cur][n].w, 0, 0, 0, 0}, acc[m][n], 4, 4, 0, 0, 0, 0);
                    };
                    if constexpr ((m & 1) == 0) e[p] = inflate_class<C3>(bits[PAR3][p]);
                    mul(std::integral_constant<int, 0>{});
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (m < 4) lds_read128<(S1 >> 1) * kPairBytes + m * 2048>(fb[cur ^ 1][m], fb_addr[S1 & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mul(std::integral_constant<int, 1>{});
                    mul(std::integral_constant<int, 2>{});
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr ((m & 1) == 0)
                        lds_write128<(S3 >> 1) * kPairBytes + (p & 1) * 16 * kRowBytes>(wr_addr[p >> 1][S3 & 1], e[p]);
                    __builtin_amdgcn_sched_barrier(0);
                    mul(std::integral_constant<int, 3>{});
                    __builtin_amdgcn_sched_barrier(0);
                    lds_read128<(S1 >> 1) * kPairBytes + m * 2048>(fa[m], fa_addr[S1 & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else
            static_for<8>([&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                // kSched 5 / 6: every wave raises its priority in its even / odd rows: of two SIMD partners in different rows the
                // one in the favoured row wins the issue, whichever is older (same priority: the older wave always wins and
                // runs ahead to the barrier, the younger one finishes the stage alone)
                if constexpr (kSched == 5) { if constexpr ((m & 1) == 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
                if constexpr (kSched == 6) { if constexpr ((m & 1) == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
                if constexpr (m == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(kRow0Wait) : "memory"); else LGKM(15);
                __builtin_amdgcn_sched_barrier(0);
                static_for<4>([&](auto pc) __attribute__((always_inline)) {
                    constexpr int p = decltype(pc)::value;
                    if constexpr (row_of(p) == m) {
                        if constexpr (kAblate & 2) e[p] = bits[PAR3][p] & (int)(0x11111111u << (C3 % 3));
                        else e[p] = inflate_class<C3>(bits[PAR3][p]);
                    }
                });
                static_for<4>([&](auto nc) __attribute__((always_inline)) {
                    constexpr int n = decltype(nc)::value;
                    if constexpr (kSched == 8 && (m & 1) == 1 && n == 2) {   // the previous row's piece, between this row's MFMAs
                        __builtin_amdgcn_sched_barrier(0);
                        lds_write128<(S3 >> 1) * kPairBytes + ((m >> 1) & 1) * 16 * kRowBytes>(wr_addr[m >> 2][S3 & 1], e[m >> 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (kAblate & 64)
                        mfma_agpr(acc[m][n], fa[m], fb[cur][n]);
                    else
                        acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                            v8i{fa[m].x, fa[m].y, fa[m].z, fa[m].w, 0, 0, 0, 0},
                            v8i{fb[cur][n].x, fb[cur][n].y, fb[cur][n].z, fb[cur][n].w, 0, 0, 0, 0}, acc[m][n], 4, 4, 0, 0, 0, 0);
                });
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (kSched == 7 && (m & 1) == 0)   // the store FIRST at the end of its row
                    lds_write128<(S3 >> 1) * kPairBytes + ((m >> 1) & 1) * 16 * kRowBytes>(wr_addr[m >> 2][S3 & 1], e[m >> 1]);
                if constexpr (!(kAblate & 8)) {
                    lds_read128<(S1 >> 1) * kPairBytes + m * 2048>(fa[m], fa_addr[S1 & 1]);
                    if constexpr (m < 4) lds_read128<(S1 >> 1) * kPairBytes + m * 2048>(fb[cur ^ 1][m], fb_addr[S1 & 1]);
                }
                static_for<4>([&](auto pc) __attribute__((always_inline)) {
                    constexpr int p = decltype(pc)::value;
                    if constexpr (row_of(p) == m && !(kAblate & 4) && kSched != 4 && kSched != 7 && kSched != 8)
                        lds_write128<(S3 >> 1) * kPairBytes + (p & 1) * 16 * kRowBytes>(wr_addr[p >> 1][S3 & 1], e[p]);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (kSched == 4 && !(kAblate & 4)) {   // the stores in front of the barrier: they hold the SIMD while it waits anyway
                static_for<4>([&](auto pc) __attribute__((always_inline)) {
                    constexpr int p = decltype(pc)::value;
                    lds_write128<(S3 >> 1) * kPairBytes + (p & 1) * 16 * kRowBytes>(wr_addr[p >> 1][S3 & 1], e[p]);
                });
            }
            if constexpr ((j & 3) == 0 && !(kAblate & 16)) load_chunk(PAR3, 2u * trip + (uint32_t)(T >> 2) + 2u);   // the chunk two behind the one just finished
            LGKM(15);
            if constexpr (!(kAblate & 1) && (!(kAblate & 32) || (j & 1) == 1)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (stamp && tid == 0) {
        atomicAdd(&g_clock[0], (unsigned long long)(__builtin_amdgcn_s_memtime() - ck_t0));
        atomicAdd(&g_clock[1], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - ck_r0));
        atomicAdd(&g_clock[2], 1ull);
    }
    static_for<8>([&](auto mc) __attribute__((always_inline)) { keep(fa[decltype(mc)::value]); });
    static_for<4>([&](auto nc) __attribute__((always_inline)) {
        keep(fb[0][decltype(nc)::value]);
        keep(fb[1][decltype(nc)::value]);
    });

    // ---- epilogue (first version: direct 4-byte stores; C/D map: col = lane & 15, row = 4 (lane >> 4) + reg)
    const uint32_t l2 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t i = it.a_row0 + wa * 128u + (uint32_t)m * 16u + 4u * (l2 >> 4) + (uint32_t)r;
                const uint32_t jj = it.b_row0 + wb * 64u + (uint32_t)n * 16u + (l2 & 15u);
                out[(uint64_t)i * ld + jj] = (uint32_t)acc[m][n][r];
            }
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t next64() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

int main(int argc, char** argv) {
    const uint32_t N = argc > 1 ? (uint32_t)atoi(argv[1]) : 1024u;
    const uint32_t M = argc > 2 ? (uint32_t)atoi(argv[2]) : 4096u;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    const int ablate = argc > 4 ? atoi(argv[4]) : 0;
    const double seconds = argc > 5 ? atof(argv[5]) : 0.0;
    const int sched = argc > 6 ? atoi(argv[6]) : 0;
    const int sync = (ablate & 31) == 0;   // (bits 5..7 keep the results right)
    const uint32_t Np = (N + 255u) / 256u * 256u;
    const uint32_t W = M / 64u;                              // words per row
    const uint64_t pitch = ((uint64_t)W * 8u + 511u) / 512u * 512u + 512u;   // whole 512-byte chunks + one (off 1 KiB multiples)
    const uint32_t n_chunks = (uint32_t)((W * 8u + 63u) / 64u + 1u) / 2u * 2u;
    std::vector<uint8_t> h((size_t)Np * pitch, 0);
    for (uint32_t i = 0; i < N; ++i) {
        uint64_t* row = reinterpret_cast<uint64_t*>(&h[(size_t)i * pitch]);
        for (uint32_t w = 0; w < W; ++w) row[w] = next64() & next64() | (next64() & next64() & next64());   // density ~0.34
    }
    uint8_t* dX;
    uint32_t* dOut;
    TileItem* dItems;
    (void)hipMalloc(&dX, h.size());
    (void)hipMemcpy(dX, h.data(), h.size(), hipMemcpyHostToDevice);
    const uint64_t ld = Np;
    (void)hipMalloc(&dOut, (size_t)Np * ld * 4);
    (void)hipMemset(dOut, 0xff, (size_t)Np * ld * 4);
    std::vector<TileItem> items;
    for (uint32_t I = 0; I < Np / 256u; ++I)
        for (uint32_t J = I; J < Np / 256u; ++J) items.push_back({I * 256u, J * 256u});
    (void)hipMalloc(&dItems, items.size() * sizeof(TileItem));
    (void)hipMemcpy(dItems, items.data(), items.size() * sizeof(TileItem), hipMemcpyHostToDevice);

    auto launch = [&](int ab) {
        const dim3 g((uint32_t)items.size()), b(512);
#define TR_CASE(A) case A: if (sched == 0) tile_ring_kernel<A, 0><<<g, b>>>(dX, pitch, n_chunks, dItems, dOut, ld, 1); else if (sched == 1) tile_ring_kernel<A, 1><<<g, b>>>(dX, pitch, n_chunks, dItems, dOut, ld, 1); else if (sched == 2) tile_ring_kernel<A, 2><<<g, b>>>(dX, pitch, n_chunks, dItems, dOut, ld, 1); else if (sched == 3) tile_ring_kernel<A, 3><<<g, b>>>(dX, pitch, n_chunks, dItems, dOut, ld, 1); else if (sched == 4) tile_ring_kernel<A, 4><<<g, b>>>(dX, pitch, n_chunks, dItems, dOut, ld, 1); else if (sched == 7) tile_ring_kernel<A, 7><<<g, b>>>(dX, pitch, n_chunks, dItems, dOut, ld, 1); else tile_ring_kernel<A, 8><<<g, b>>>(dX, pitch, n_chunks, dItems, dOut, ld, 1); break;
        switch (ab) {
            TR_CASE(0) TR_CASE(1)
            default: printf("unknown ablation %d\n", ab); exit(2);
        }
    };
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < reps + 2; ++rep) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        if (rep == 2) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clock), z, sizeof(z));
        (void)hipEventRecord(e0);
        launch(ablate);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    double mono_begin = 0, mono_end = 0, ms_sustained = 0;
    if (seconds > 0) {   // sustained: back-to-back launches, for tools/clock_power.py's sampler
        auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
        const unsigned long long z[4] = {0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clock), z, sizeof(z));
        std::vector<double> t;
        mono_begin = now();
        while (now() - mono_begin < seconds) {
            (void)hipEventRecord(e0);
            for (int k = 0; k < 16; ++k) launch(ablate);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            t.push_back(ms / 16.0);
        }
        mono_end = now();
        const size_t q = t.size() / 4 ? t.size() / 4 : 1;
        for (size_t i = t.size() - q; i < t.size(); ++i) ms_sustained += t[i] / (double)q;
    }
    hipError_t err = hipDeviceSynchronize();
    unsigned long long ck[4] = {0, 0, 0, 0};
    (void)hipMemcpyFromSymbol(ck, HIP_SYMBOL(g_clock), sizeof(ck));
    // check sampled entries against a CPU popcount
    std::vector<uint32_t> ho((size_t)Np * ld);
    (void)hipMemcpy(ho.data(), dOut, ho.size() * 4, hipMemcpyDeviceToHost);
    uint64_t bad = 0, checked = 0;
    for (int t = 0; t < 20000; ++t) {
        const uint32_t i = (uint32_t)(next64() % N), j = (uint32_t)(next64() % N);
        if (i >= j) continue;
        const uint64_t* ri = reinterpret_cast<const uint64_t*>(&h[(size_t)i * pitch]);
        const uint64_t* rj = reinterpret_cast<const uint64_t*>(&h[(size_t)j * pitch]);
        uint32_t c = 0;
        for (uint32_t w = 0; w < W; ++w) c += (uint32_t)__builtin_popcountll(ri[w] & rj[w]);
        ++checked;
        if (ho[(size_t)i * ld + j] != c) {
            if (bad < 5) printf("mismatch (%u, %u): got %u want %u\n", i, j, ho[(size_t)i * ld + j], c);
            ++bad;
        }
    }
    const double flop = (double)items.size() * 256.0 * 256.0 * (double)n_chunks * 512.0 * 2.0;
    char bus[64] = {0};
    (void)hipDeviceGetPCIBusId(bus, sizeof(bus), 0);
    if (seconds > 0) best = (float)ms_sustained;
    printf("{\"kernel\": \"tile_ring\", \"sched\": %d, \"ablate\": %d, \"pci_bus\": \"%s\", \"mono_begin\": %.6f, \"mono_end\": %.6f, ", sched, ablate, bus, mono_begin, mono_end);
    printf("\"sync\": %d, \"rows\": %u, \"bits\": %u, \"tiles\": %zu, \"ms\": %.4f, \"frac_of_10_pflops_computed\": %.4f, "
           "\"in_kernel_clock_mhz\": %.1f, \"checked\": %llu, \"bad\": %llu, \"hip_error\": %d}\n",
           sync, N, M, items.size(), best, flop / (best * 1e-3) / 1e16, ck[1] ? 100.0 * (double)ck[0] / (double)ck[1] : 0.0,
           (unsigned long long)checked, (unsigned long long)bad, (int)err);
    return bad != 0 && sync == 1;
}
