// tools/probes/bitwave.hip — K2w: the stage stream with a private ring per wave.
// TOOLS BUILD ONLY (`make -C stormbitmaps_amd/csrc probes` -> libstorm_hip_probes.so): this file is a fragment of
// stormbitmaps_amd/csrc/storm_hip_mfma.hip, included there under -DSTORM_HIP_PROBES at the place the code used to
// stand; it is not part of the shipped library.

// ------------------------------------------------------------------------------------------
// K2w: the same stage stream with a PRIVATE ring per wave (option k2_strip_operands = 3; TOOLS BUILD ONLY:
// measured slower than bitstream_kernel everywhere but at N = 512 — 11.1 against 12.3 us there, 25.1 / 19.8 at
// N = 1024, 48.4 / 44.5 at 2048, 160 / 148 at 4096, 617 / 560 at 8192, same box, profiles/r03_g_wave_private_ring.txt:
// what a lone workgroup loses at its barrier is less than what four times the L2 -> LDS traffic and a ring of
// three stages cost).
//
// In bitstream_kernel the four waves of a workgroup share every B stage (one DMA piece each) and meet at a
// barrier every two stages; a workgroup alone on its CU has nothing to cover that wait and the DMA's (~290
// of 1670 clocks per stage), and the barrier couples four SIMDs. Here every wave DMAs the WHOLE stage (four
// pieces of 16 rows x 64 B) into its own ring of kRing stages: no barrier before the final fold, only vmcnt;
// the L2 -> LDS traffic is four times the shared ring's (16 B per clock and CU at the full rate: a quarter of
// the path), the HBM traffic is unchanged.
// What a stage is to a wave comes from ONE table word (bitwave tables, build_bitwave): bits 0..29 the start of
// the stage's 64 rows (64-byte units), bit 31 "these are my A rows: take them", bit 30 "and multiply them, at
// half weight" (a diagonal segment). A wave's list for a segment is its own block, the tile's later blocks
// (diagonal segments), the run of later blocks; lists of the four waves differ in length by up to three
// stages per diagonal segment and even out through the rotation of the blocks over the waves.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kBwOwn = 0x80000000u, kBwMul = 0x40000000u, kBwBase = 0x3fffffffu;

template <int kRing>
__global__ __launch_bounds__(kStripThreads, 3) void bitwave_kernel(
    const uint8_t* __restrict__ X, uint64_t pitch64, const uint32_t* __restrict__ first,
    const uint32_t* __restrict__ words, unsigned long long* __restrict__ slots,
    unsigned long long* __restrict__ out) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds_raw[kStripWaves * kRing * kSbStageBytes];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t pitch = (uint32_t)pitch64;
    uint8_t* ring = lds_raw + wave * (kRing * kSbStageBytes);

    const uint32_t goff = (lane >> 2) * pitch + (((lane & 3u) ^ ((lane >> 4) & 3u)) * 16u);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)ring;
    const uint32_t slot0 = (lane >> 5) ^ (((lane & 31u) >> 2) & 3u);
    const uint32_t baddr0 = lds_base + (lane & 31u) * kSbRowBytes + slot0 * 16u;
    const uint32_t baddr1 = lds_base + (lane & 31u) * kSbRowBytes + (slot0 ^ 2u) * 16u;

    v16f acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = v16f{};
    v4i a[2][4][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int m = 0; m < 2; ++m) a[g][c][m] = v4i{};
    uint32_t dbits = 0;

    const uint32_t w0i = first[blockIdx.x * 4u + wave];
    const uint32_t T = first[blockIdx.x * 4u + wave + 1u] - w0i;
    const uint32_t* tab = words + w0i;
    uint32_t issued = 0;
    uint32_t next_word = T ? tab[0] : 0u;  // of stage `issued`
    auto fire = [&]() {
        if (issued < T) {
            uint8_t* src = const_cast<uint8_t*>(X) + ((uint64_t)(next_word & kBwBase) << 6);
            uint8_t* dst = ring + (issued % kRing) * kSbStageBytes;
            // piece j: rows 16 j .. 16 j + 15 of the stage, 1 KiB further into the ring slot. ONE value of M0 per stage:
            // the instruction offset moves the LDS address (and the global one, which the base takes back).
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(src, 0, -1, 0x00020000);
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + j * 1024u), 16,
                                                         (int)(goff + j * 16u * pitch), 0, 0, 0);
            ++issued;
            next_word = tab[min(issued, T - 1u)];
        }
    };
#pragma unroll
    for (int k = 0; k < kRing - 1; ++k) fire();

#define STORM_BS_FETCH(dst, t, n, g) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(((g) ? baddr1 : baddr0) + ((t) % kRing) * kSbStageBytes), "n"((n) * 32 * kSbRowBytes))
#define STORM_BS_STEP(n, g, C, ecur, enxt, NEXT)                                                          \
    {                                                                                                     \
        acc[0][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                                      \
            v8i{a[g][C][0].x, a[g][C][0].y, a[g][C][0].z, a[g][C][0].w, 0, 0, 0, 0},                      \
            v8i{ecur.x, ecur.y, ecur.z, ecur.w, 0, 0, 0, 0}, acc[0][n], 4, 4, 0, tb_scale<C>(), 0, sb[C]); \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        const v4i en_ = NEXT;                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        acc[1][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                                      \
            v8i{a[g][C][1].x, a[g][C][1].y, a[g][C][1].z, a[g][C][1].w, 0, 0, 0, 0},                      \
            v8i{ecur.x, ecur.y, ecur.z, ecur.w, 0, 0, 0, 0}, acc[1][n], 4, 4, 0, tb_scale<C>(), 0, sb[C]); \
        enxt = en_;                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    }
#define STORM_BS_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0)
#define STORM_BS_KEEP() asm volatile("" ::"v"(w0), "v"(w1), "v"(e0))
    // one stage; the next stage's first word is NOT fetched here (its DMA is waited for at the top of the loop)
#define STORM_BW_STAGE(tc)                                          \
    STORM_BS_FETCH(w1, tc, 1, 0);                                   \
    __builtin_amdgcn_sched_barrier(0);                              \
    STORM_BS_STEP(0, 0, 0, e0, e0, tb_inflate<1>(w0));              \
    STORM_BS_STEP(0, 0, 1, e0, e0, tb_inflate<2>(w0));              \
    STORM_BS_STEP(0, 0, 2, e0, e0, tb_inflate<3>(w0));              \
    STORM_BS_WAIT();                                                \
    STORM_BS_STEP(0, 0, 3, e0, e0, tb_inflate<0>(w1));              \
    STORM_BS_FETCH(w0, tc, 0, 1);                                   \
    __builtin_amdgcn_sched_barrier(0);                              \
    STORM_BS_STEP(1, 0, 0, e0, e0, tb_inflate<1>(w1));              \
    STORM_BS_STEP(1, 0, 1, e0, e0, tb_inflate<2>(w1));              \
    STORM_BS_STEP(1, 0, 2, e0, e0, tb_inflate<3>(w1));              \
    STORM_BS_WAIT();                                                \
    STORM_BS_STEP(1, 0, 3, e0, e0, tb_inflate<0>(w0));              \
    STORM_BS_FETCH(w1, tc, 1, 1);                                   \
    __builtin_amdgcn_sched_barrier(0);                              \
    STORM_BS_STEP(0, 1, 0, e0, e0, tb_inflate<1>(w0));              \
    STORM_BS_STEP(0, 1, 1, e0, e0, tb_inflate<2>(w0));              \
    STORM_BS_STEP(0, 1, 2, e0, e0, tb_inflate<3>(w0));              \
    STORM_BS_WAIT();                                                \
    STORM_BS_STEP(0, 1, 3, e0, e0, tb_inflate<0>(w1));              \
    __builtin_amdgcn_sched_barrier(0);                              \
    STORM_BS_STEP(1, 1, 0, e0, e0, tb_inflate<1>(w1));              \
    STORM_BS_STEP(1, 1, 1, e0, e0, tb_inflate<2>(w1));              \
    STORM_BS_STEP(1, 1, 2, e0, e0, tb_inflate<3>(w1));              \
    STORM_BS_STEP(1, 1, 3, e0, e0, e0)

    v4i w0 = {}, w1 = {}, e0 = {};
    int sb[4] = {tb_scale<0>(), tb_scale<1>(), tb_scale<2>(), tb_scale<3>()};
    uint32_t cur_word = T ? tab[0] : 0u;
#pragma unroll 1
    for (uint32_t t = 0; t < T; ++t) {
        // stage t has landed when at most the pieces of the kRing - 2 younger stages are in flight
        if (issued >= t + (uint32_t)(kRing - 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kRing - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t word = cur_word;
        cur_word = tab[min(t + 1u, T - 1u)];
        fire();  // into the slot of stage t - 1: this wave's reads of it were waited for (lgkmcnt) in its stage
        __builtin_amdgcn_sched_barrier(0);
        const bool own = (word & kBwOwn) != 0u;
        const bool mul = !own || (word & kBwMul) != 0u;
        // The reads and the wait that covers them are ONE asm statement each: a read left in flight across
        // compiler-visible code is not safe — for a wait that ties the words ("+v") hipcc copied w0 into the
        // tied registers BEFORE the wait, i.e. before the data had landed (rows 32..63 of a block came out wrong).
        const uint32_t rd0 = baddr0 + (t % kRing) * kSbStageBytes, rd1 = baddr1 + (t % kRing) * kSbStageBytes;
        if (own) {
            v4i x1, x2, x3;
            asm volatile("ds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:2048\n\t"
                         "ds_read_b128 %2, %5 offset:0\n\tds_read_b128 %3, %5 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(w0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                         : "v"(rd0), "v"(rd1)
                         : "memory");
            __builtin_amdgcn_sched_barrier(0);
            a[0][0][0] = tb_inflate<0>(w0); a[0][1][0] = tb_inflate<1>(w0);
            a[0][2][0] = tb_inflate<2>(w0); a[0][3][0] = tb_inflate<3>(w0);
            a[0][0][1] = tb_inflate<0>(x1); a[0][1][1] = tb_inflate<1>(x1);
            a[0][2][1] = tb_inflate<2>(x1); a[0][3][1] = tb_inflate<3>(x1);
            a[1][0][0] = tb_inflate<0>(x2); a[1][1][0] = tb_inflate<1>(x2);
            a[1][2][0] = tb_inflate<2>(x2); a[1][3][0] = tb_inflate<3>(x2);
            a[1][0][1] = tb_inflate<0>(x3); a[1][1][1] = tb_inflate<1>(x3);
            a[1][2][1] = tb_inflate<2>(x3); a[1][3][1] = tb_inflate<3>(x3);
            if (mul) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    dbits += __builtin_popcount((uint32_t)w0[k]) + __builtin_popcount((uint32_t)x1[k]) +
                             __builtin_popcount((uint32_t)x2[k]) + __builtin_popcount((uint32_t)x3[k]);
            }
        } else {
            asm volatile("ds_read_b128 %0, %1 offset:0\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0) : "v"(rd0) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        if (mul) {
            const int half = own ? 1 : 0;
            sb[0] = tb_scale<0>() - half;
            sb[1] = tb_scale<1>() - half;
            sb[2] = tb_scale<2>() - half;
            sb[3] = tb_scale<3>() - half;
            e0 = tb_inflate<0>(w0);
            __builtin_amdgcn_sched_barrier(0);
            STORM_BW_STAGE(t);
            STORM_BS_WAIT();
            STORM_BS_KEEP();
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef STORM_BW_STAGE
#undef STORM_BS_KEEP
#undef STORM_BS_WAIT
#undef STORM_BS_STEP
#undef STORM_BS_FETCH

    long long mine2 = 0;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        uint32_t part = 0;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) part += (uint32_t)(acc[m][n][r] * 2.0f);
        mine2 += part;
    }
    mine2 -= (long long)dbits;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine2 += __shfl_down(mine2, o, 64);
    __builtin_amdgcn_s_barrier();
    long long* wsum = reinterpret_cast<long long*>(lds_raw);
    if (lane == 0) wsum[wave] = mine2;
    __syncthreads();
    if (tid == 0) {
        const long long tot2 = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (tot2 != 0) atomicAdd(&slots[blockIdx.x & (kBsFoldSlots - 1)], (unsigned long long)(tot2 / 2));
        __threadfence();
        const unsigned long long arrived = atomicAdd(&slots[kBsTicket], 1ull);
        wsum[4] = (arrived == (unsigned long long)gridDim.x - 1ull) ? 1 : 0;
    }
    __syncthreads();
    if (wsum[4] != 0 && wave == 0) {
        __threadfence();
        unsigned long long v = __hip_atomic_exchange(&slots[lane], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) {
            out[0] = v;
            __hip_atomic_store(&slots[kBsTicket], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

