// tools/probes/stripbits.hip — bit-operand strips, one item per workgroup, operands inflated in registers (superseded by K2q and K2b).
// TOOLS BUILD ONLY (`make -C stormbitmaps_amd/csrc probes` -> libstorm_hip_probes.so): this file is a fragment of
// stormbitmaps_amd/csrc/storm_hip_mfma.hip, included there under -DSTORM_HIP_PROBES at the place the code used to
// stand; it is not part of the shipped library.

constexpr int kSbRing = 4;
__global__ __launch_bounds__(kStripThreads, 3) void stripbits_kernel(
    const uint8_t* __restrict__ X, uint64_t pitch64, const StripItem* __restrict__ items,
    unsigned long long* __restrict__ slots) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds_raw[kSbRing * kSbStageBytes];
    auto lds = reinterpret_cast<uint8_t(*)[kSbStageBytes]>(lds_raw);

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wm = wave;  // waves stacked along A; every wave multiplies all 64 B rows
    const uint32_t item_idx = blockIdx.x;
    const StripItem it = items[item_idx];
    const uint32_t pitch = (uint32_t)pitch64;
    const uint8_t* Xk = X + (uint64_t)it.ks * kSbRowBytes;  // the item's k-slice of row 0
    constexpr uint32_t kATile = (uint32_t)kStripATile;
    const uint32_t D = it.diag ? kATile / (uint32_t)kStripBRows : 0u;
    const uint32_t T = D + (it.j1 - it.j0);

    // B stage = 4 LDS-DMA pieces of 16 rows x 64 B, one per wave. Lane L fills row L / 4 of the piece,
    // physical slot L % 4 = logical slot (L % 4) ^ ((L / 16) % 4)  (image: slot s of row r at s ^ ((r / 4) % 4))
    const uint32_t goff = (wave * 16u + (lane >> 2)) * pitch + (((lane & 3u) ^ ((lane >> 4) & 3u)) * 16u);
    auto issue = [&](uint32_t t) {
        const uint32_t blk = t < D ? it.a_row0 / (uint32_t)kStripBRows + t : it.j1 - 1u - (t - D);
        const uint8_t* base = Xk + (uint64_t)blk * ((uint64_t)kStripBRows * pitch64);
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, -1, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds[t % kSbRing] + wave * 1024u), 16, (int)goff, 0,
                                                 0, 0);
    };

    // A bits first (older in the VMEM queue than the DMAs)
    v4i abits[2][2];  // [k-group][row block]
    {
        const uint8_t* ap = Xk + (uint64_t)(it.a_row0 + wm * 64u + (lane & 31u)) * pitch64 + (lane >> 5) * 16u;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                abits[g][m] = *reinterpret_cast<const v4i*>(ap + (uint64_t)m * 32u * pitch64 + g * 32);
    }
#pragma unroll
    for (uint32_t t = 0; t < kSbRing - 1; ++t)
        if (t < T) issue(t);

    v16f acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = v16f{};

    // fragment (block n, k-group g): row 32 n + (lane & 31), logical slot 2 g + (lane >> 5)
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&lds[0][0];
    const uint32_t slot0 = (lane >> 5) ^ (((lane & 31u) >> 2) & 3u);
    const uint32_t baddr0 = lds_base + (lane & 31u) * kSbRowBytes + slot0 * 16u;
    const uint32_t baddr1 = lds_base + (lane & 31u) * kSbRowBytes + (slot0 ^ 2u) * 16u;

    // the A operands, all four classes (retires the A loads: older than the DMAs, the ring stays in flight)
    v4i a[2][4][2];  // [k-group][class][row block]
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            a[g][0][m] = tb_inflate<0>(abits[g][m]);
            a[g][1][m] = tb_inflate<1>(abits[g][m]);
            a[g][2][m] = tb_inflate<2>(abits[g][m]);
            a[g][3][m] = tb_inflate<3>(abits[g][m]);
        }

    // retire(t, ahead): this wave's piece of stage t has landed — `ahead` younger pieces (one per stage)
    // may stay in flight while that many stages exist beyond t, else everything is drained — and the
    // barrier makes every wave's piece visible and says that every wave is done with the stages before.
    auto retire = [&](uint32_t t, uint32_t ahead) {
        if (ahead == 2u && t + 2u < T) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (ahead == 1u && t + 1u < T) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // An inline-asm read completes long after hipcc thinks it has: a fetched word must stay LIVE (in hipcc's
    // eyes) until the wait that covers it. The first version's diagonal phase ended every stage with the same
    // look-ahead read as the main loop but never used its result; hipcc gave the dead output's registers to
    // the next inflated operand, the LDS data landed on top of it some 100 cycles later, and the totals came
    // out different from run to run — only with several workgroups per CU, where the LDS answers late enough
    // (tools/mfma_war_probe: the hardware itself never lets an LDS return overtake an MFMA's operand read).
    // STORM_SB_KEEP marks the words as used behind the wait.
#define STORM_SB_FETCH(dst, t, n, g) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(((g) ? baddr1 : baddr0) + ((t) % kSbRing) * kSbStageBytes), "n"((n) * 32 * kSbRowBytes))
#define STORM_SB_STEP(n, g, C, ecur, enxt, NEXT)                                                          \
    acc[0][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                                          \
        v8i{a[g][C][0].x, a[g][C][0].y, a[g][C][0].z, a[g][C][0].w, 0, 0, 0, 0},                          \
        v8i{ecur.x, ecur.y, ecur.z, ecur.w, 0, 0, 0, 0}, acc[0][n], 4, 4, 0, tb_scale<C>(), 0, tb_scale<C>()); \
    acc[1][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                                          \
        v8i{a[g][C][1].x, a[g][C][1].y, a[g][C][1].z, a[g][C][1].w, 0, 0, 0, 0},                          \
        v8i{ecur.x, ecur.y, ecur.z, ecur.w, 0, 0, 0, 0}, acc[1][n], 4, 4, 0, tb_scale<C>(), 0, tb_scale<C>()); \
    enxt = NEXT;                                                                                          \
    __builtin_amdgcn_sched_barrier(0)
#define STORM_SB_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0)
#define STORM_SB_KEEP() asm volatile("" ::"v"(w0), "v"(w1), "v"(e0))
    // One stage. On entry w0 holds the bits of (block 0, k-group 0) of stage `tc` and e0 their class 0; on
    // exit the same of stage `tn` (the look-ahead: the next stage, or a re-read that is never consumed).
    // Word order: (n, g) = (0,0) (1,0) (0,1) (1,1) in w0, w1, w0, w1; every word is fetched while the one
    // before it runs its first three classes.
#define STORM_SB_STAGE(tc, tn)                                      \
    STORM_SB_FETCH(w1, tc, 1, 0);                                   \
    __builtin_amdgcn_sched_barrier(0);                              \
    STORM_SB_STEP(0, 0, 0, e0, e0, tb_inflate<1>(w0));              \
    STORM_SB_STEP(0, 0, 1, e0, e0, tb_inflate<2>(w0));              \
    STORM_SB_STEP(0, 0, 2, e0, e0, tb_inflate<3>(w0));              \
    STORM_SB_WAIT();                                                \
    STORM_SB_STEP(0, 0, 3, e0, e0, tb_inflate<0>(w1));              \
    STORM_SB_FETCH(w0, tc, 0, 1);                                   \
    __builtin_amdgcn_sched_barrier(0);                              \
    STORM_SB_STEP(1, 0, 0, e0, e0, tb_inflate<1>(w1));              \
    STORM_SB_STEP(1, 0, 1, e0, e0, tb_inflate<2>(w1));              \
    STORM_SB_STEP(1, 0, 2, e0, e0, tb_inflate<3>(w1));              \
    STORM_SB_WAIT();                                                \
    STORM_SB_STEP(1, 0, 3, e0, e0, tb_inflate<0>(w0));              \
    STORM_SB_FETCH(w1, tc, 1, 1);                                   \
    __builtin_amdgcn_sched_barrier(0);                              \
    STORM_SB_STEP(0, 1, 0, e0, e0, tb_inflate<1>(w0));              \
    STORM_SB_STEP(0, 1, 1, e0, e0, tb_inflate<2>(w0));              \
    STORM_SB_STEP(0, 1, 2, e0, e0, tb_inflate<3>(w0));              \
    STORM_SB_WAIT();                                                \
    STORM_SB_STEP(0, 1, 3, e0, e0, tb_inflate<0>(w1));              \
    STORM_SB_FETCH(w0, tn, 0, 0);                                   \
    __builtin_amdgcn_sched_barrier(0);                              \
    STORM_SB_STEP(1, 1, 0, e0, e0, tb_inflate<1>(w1));              \
    STORM_SB_STEP(1, 1, 1, e0, e0, tb_inflate<2>(w1));              \
    STORM_SB_STEP(1, 1, 2, e0, e0, tb_inflate<3>(w1));              \
    STORM_SB_WAIT();                                                \
    STORM_SB_STEP(1, 1, 3, e0, e0, tb_inflate<0>(w0))

    v4i w0 = {}, w1 = {}, e0 = {};  // one inflated operand: the next one is computed behind the step's second MFMA
    uint32_t t = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // scalar loads of the item record
    // ---- the A tile's own 4 blocks (strict upper triangle), not pipelined across stages: wave wm skips
    //      the blocks before its own rows, masks its own 64 x 64 block, takes the later ones whole
#pragma unroll 1
    for (; t < D; ++t) {
        retire(t, 2u);
        if (t + kSbRing - 1 < T) issue(t + kSbRing - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (t >= wm) {
            STORM_SB_FETCH(w0, t, 0, 0);
            STORM_SB_WAIT();
            e0 = tb_inflate<0>(w0);
            STORM_SB_STAGE(t, t);
            STORM_SB_WAIT();  // the look-ahead read, not consumed in this phase ...
            STORM_SB_KEEP();  // ... but alive until it has landed
            if (t == wm) {
                // the accumulators have seen nothing but this stage: clear the pairs with i >= j in place
                // (C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
                acc[1][0] = v16f{};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const bool keep = row < (lane & 31u);
                    acc[0][0][r] = keep ? acc[0][0][r] : 0.0f;
                    acc[1][1][r] = keep ? acc[1][1][r] : 0.0f;
                }
            }
        }
    }
    // ---- later blocks: stage t + 1 is retired at the top of iteration t, so that its first word can be
    //      fetched while stage t still multiplies; the refill of the ring follows the barrier
    if (t < T) {
        retire(t, 2u);
        STORM_SB_FETCH(w0, t, 0, 0);
        STORM_SB_WAIT();
        e0 = tb_inflate<0>(w0);
        for (; t < T; ++t) {
            if (t + 1 < T) retire(t + 1, 1u);
            else __builtin_amdgcn_s_barrier();
            if (t + kSbRing - 1 < T) issue(t + kSbRing - 1);
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t tn = t + 1 < T ? t + 1 : t;
            STORM_SB_STAGE(t, tn);
        }
        STORM_SB_WAIT();
        STORM_SB_KEEP();
    }
#undef STORM_SB_STAGE
#undef STORM_SB_KEEP
#undef STORM_SB_WAIT
#undef STORM_SB_STEP
#undef STORM_SB_FETCH

    uint64_t mine = 0;
#pragma unroll
    for (int m = 0; m < 2; ++m) {  // one 32 x 64 strip at a time stays below 2^32
        uint32_t part = 0;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) part += (uint32_t)acc[m][n][r];
        mine += part;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if (lane == 0 && mine != 0)
        atomicAdd(&slots[(item_idx * (uint32_t)kStripWaves + wave) & (kSlots - 1)], (unsigned long long)mine);
}
