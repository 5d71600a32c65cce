// tools/probes/tile16_fp4.hip — FP4-shadow output kernel on 16x16x128 MFMAs (45 % slower than tilebits8_kernel).
// TOOLS BUILD ONLY (`make -C stormbitmaps_amd/csrc probes` -> libstorm_hip_probes.so): this file is a fragment of
// stormbitmaps_amd/csrc/storm_hip_mfma.hip, included there under -DSTORM_HIP_PROBES at the place the code used to
// stand; it is not part of the shipped library.

__global__ __launch_bounds__(kMfmaThreads, 2) void tile16_fp4_kernel(
    const uint8_t* __restrict__ X4, uint64_t row_bytes, const MfmaItem* __restrict__ items,
    uint32_t* __restrict__ out, uint64_t ld, uint32_t n_rows, const uint32_t* __restrict__ row_counts,
    uint32_t and_weight, uint32_t j_base, uint32_t j_count, uint32_t split_from, uint32_t i_lo,
    uint32_t n_cols) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[kT16Ring][kT16StageBytes];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t item_idx = blockIdx.x;
    const MfmaItem it = items[item_idx];
    const uint32_t a_row0 = (uint32_t)it.I * kTile, b_row0 = (uint32_t)it.J * kTile;
    // items count 64-byte stages (MfmaItem, shared with the 32x32 kernel): two per stage here; an odd
    // count cannot occur (rows are padded to 64 words = 32 such stages, k-parts are cut on even stages)
    const uint32_t S = it.n_stages / 2u;
    const uint64_t kbyte0 = (uint64_t)it.stage0 * kStageBytes;

    // B stage DMA: instruction n fills LDS bytes [1024 n, +1024) = rows 8 n .. 8 n + 7; piece
    // p = 64 n + lane is row p / 8, physical slot p % 8, which holds the row's logical 16-byte slot
    // (p % 8) ^ ((row / 2) % 8). Wave w issues instructions w, w + 8, w + 16, w + 24 (rows + 64 each:
    // same swizzle), so one per-lane offset serves all four.
    const uint32_t brow = wave * 8u + (lane >> 3);
    const uint32_t boff = brow * (uint32_t)row_bytes + (((lane & 7u) ^ ((brow >> 1) & 7u)) * 16u);
    auto issue_b = [&](uint32_t s) {
        const uint8_t* base = X4 + (uint64_t)b_row0 * row_bytes + kbyte0 + (uint64_t)s * kT16RowBytes;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, -1, 0x00020000);
        uint8_t* dst = lds[s % kT16Ring] + wave * 1024u;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + q * 8192u), 16, (int)boff,
                                                     (int)(q * 64u * (uint32_t)row_bytes), 0, 0);
    };
    // A fragments of stage s: rows a_row0 + 32 w + 16 m + (lane & 15), bytes 128 s + 64 kk + 16 (lane >> 4)
    // (inline asm: hipcc then leaves the waiting to the counted vmcnt below; for loads it can see it
    //  drains vmcnt(0) — and with it the DMA ring — in front of the first MFMA of a loop body)
    const uint8_t* ap = X4 + (uint64_t)(a_row0 + wave * 32u + (lane & 15u)) * row_bytes + kbyte0 +
                        (lane >> 4) * 16u;
    v4i aA[2][2], aB[2][2];  // [kk][m]; stage s lives in aA for even s, aB for odd s
    auto load_a = [&](uint32_t s, v4i (&dst)[2][2]) {
        const uint8_t* p0 = ap + (uint64_t)s * kT16RowBytes;
        const uint8_t* p1 = p0 + 16ull * row_bytes;
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                     "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %5, off offset:64"
                     : "=&v"(dst[0][0]), "=&v"(dst[0][1]), "=&v"(dst[1][0]), "=&v"(dst[1][1])
                     : "v"(p0), "v"(p1)
                     : "memory");
    };

    v4f acc[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 16; ++n) acc[m][n] = v4f{};

    // B fragment (k-step kk, block n) of a stage: row 16 n + (lane & 15), logical slot 4 kk + (lane >> 4)
    const uint32_t lds_base =
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&lds[0][0];
    const uint32_t swz = ((lane & 15u) >> 1) & 7u;
    uint32_t frag[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
        frag[kk] = lds_base + (lane & 15u) * kT16RowBytes + ((((uint32_t)kk * 4u + (lane >> 4)) ^ swz) * 16u);

    // prologue, in the loop's issue order: the stages that are older than everything, then the
    // pseudo-iteration -1
    issue_b(0);
    if (1 < S) issue_b(1);
    load_a(0, aA);
    if (2 < S) issue_b(2);

#define STORM_T16_FETCH(dst, stage_base, kk, n) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(frag[kk] + (stage_base)), "n"((n) * 16 * kT16RowBytes))
#define STORM_T16_MUL(kk, n, av, bv)                                                                \
    acc[0][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                                   \
        v8i{av[kk][0].x, av[kk][0].y, av[kk][0].z, av[kk][0].w, 0, 0, 0, 0},                        \
        v8i{bv.x, bv.y, bv.z, bv.w, 0, 0, 0, 0}, acc[0][n], 4, 4, 0, 0, 0, 0);                      \
    acc[1][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                                   \
        v8i{av[kk][1].x, av[kk][1].y, av[kk][1].z, av[kk][1].w, 0, 0, 0, 0},                        \
        v8i{bv.x, bv.y, bv.z, bv.w, 0, 0, 0, 0}, acc[1][n], 4, 4, 0, 0, 0, 0)
    // step t: waves 4-7 issue the stage's DMA in front of step 16 (see "stagger" above)
#define STORM_T16_STEP(kk, n, av, cur, nxt, nxt_base, nxt_kk, nxt_n, t)  \
    if ((t) == 16 && late_dma && dma_stage < S) issue_b(dma_stage);      \
    STORM_T16_FETCH(nxt, nxt_base, nxt_kk, nxt_n);                       \
    asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");                   \
    __builtin_amdgcn_sched_barrier(0);                                   \
    STORM_T16_MUL(kk, n, av, cur);                                       \
    __builtin_amdgcn_sched_barrier(0)
#define STORM_T16_STAGE(av) \
    STORM_T16_STEP(0, 0, av, b0, b3, sb, 0, 3, 0); \
    STORM_T16_STEP(0, 1, av, b1, b0, sb, 0, 4, 1); \
    STORM_T16_STEP(0, 2, av, b2, b1, sb, 0, 5, 2); \
    STORM_T16_STEP(0, 3, av, b3, b2, sb, 0, 6, 3); \
    STORM_T16_STEP(0, 4, av, b0, b3, sb, 0, 7, 4); \
    STORM_T16_STEP(0, 5, av, b1, b0, sb, 0, 8, 5); \
    STORM_T16_STEP(0, 6, av, b2, b1, sb, 0, 9, 6); \
    STORM_T16_STEP(0, 7, av, b3, b2, sb, 0, 10, 7); \
    STORM_T16_STEP(0, 8, av, b0, b3, sb, 0, 11, 8); \
    STORM_T16_STEP(0, 9, av, b1, b0, sb, 0, 12, 9); \
    STORM_T16_STEP(0, 10, av, b2, b1, sb, 0, 13, 10); \
    STORM_T16_STEP(0, 11, av, b3, b2, sb, 0, 14, 11); \
    STORM_T16_STEP(0, 12, av, b0, b3, sb, 0, 15, 12); \
    STORM_T16_STEP(0, 13, av, b1, b0, sb, 1, 0, 13); \
    STORM_T16_STEP(0, 14, av, b2, b1, sb, 1, 1, 14); \
    STORM_T16_STEP(0, 15, av, b3, b2, sb, 1, 2, 15); \
    STORM_T16_STEP(1, 0, av, b0, b3, sb, 1, 3, 16); \
    STORM_T16_STEP(1, 1, av, b1, b0, sb, 1, 4, 17); \
    STORM_T16_STEP(1, 2, av, b2, b1, sb, 1, 5, 18); \
    STORM_T16_STEP(1, 3, av, b3, b2, sb, 1, 6, 19); \
    STORM_T16_STEP(1, 4, av, b0, b3, sb, 1, 7, 20); \
    STORM_T16_STEP(1, 5, av, b1, b0, sb, 1, 8, 21); \
    STORM_T16_STEP(1, 6, av, b2, b1, sb, 1, 9, 22); \
    STORM_T16_STEP(1, 7, av, b3, b2, sb, 1, 10, 23); \
    STORM_T16_STEP(1, 8, av, b0, b3, sb, 1, 11, 24); \
    STORM_T16_STEP(1, 9, av, b1, b0, sb, 1, 12, 25); \
    STORM_T16_STEP(1, 10, av, b2, b1, sb, 1, 13, 26); \
    STORM_T16_STEP(1, 11, av, b3, b2, sb, 1, 14, 27); \
    STORM_T16_STEP(1, 12, av, b0, b3, sb, 1, 15, 28); \
    STORM_T16_STEP(1, 13, av, b1, b0, sn, 0, 0, 29); \
    STORM_T16_STEP(1, 14, av, b2, b1, sn, 0, 1, 30); \
    STORM_T16_STEP(1, 15, av, b3, b2, sn, 0, 2, 31);

    const bool late_dma = wave >= 4u;
    v4i b0 = {}, b1 = {}, b2 = {}, b3 = {};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // scalar loads of the item record
    // stage 0 must be in the LDS before its first three fragments are fetched (later stages: fetched
    // by the stage before); draining the prologue's prefetch once per item costs nothing measurable
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    STORM_T16_FETCH(b0, 0u, 0, 0);
    STORM_T16_FETCH(b1, 0u, 0, 1);
    STORM_T16_FETCH(b2, 0u, 0, 2);
    // One stage: A(s) in `use`, A(s+1) loaded into `into`. The loop body is two stages long so that
    // the alternation of the A registers is a matter of names, not of branches or moves (a
    // multi-armed body made hipcc keep several accumulator sets and spill).
#define STORM_T16_BODY(s, use, into)                                                           \
    {                                                                                          \
        /* A(s) and the B stages up to s + 1 are in once only stage s + 2's 4 DMAs remain */   \
        if ((s) + kT16Ring - 1 < S) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");           \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                  \
        __builtin_amdgcn_s_barrier();                                                          \
        const uint32_t sb = ((s) % kT16Ring) * kT16StageBytes;                                 \
        /* (after the last stage `sn` re-reads the same stage: never consumed; branch-free body) */ \
        const uint32_t sn = (((s) + 1 < S ? (s) + 1 : (s)) % kT16Ring) * kT16StageBytes;       \
        const uint32_t dma_stage = (s) + kT16Ring - 1;                                         \
        if ((s) + 1 < S) load_a((s) + 1, into);                                                \
        if (!late_dma && dma_stage < S) issue_b(dma_stage);                                    \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        STORM_T16_STAGE(use);                                                                  \
    }
    uint32_t s = 0;
    for (; s + 2 <= S; s += 2) {
        STORM_T16_BODY(s, aA, aB);
        STORM_T16_BODY(s + 1, aB, aA);
    }
    if (s < S) {
        STORM_T16_BODY(s, aA, aB);
        ++s;
    }
#undef STORM_T16_BODY
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef STORM_T16_STAGE
#undef STORM_T16_STEP
#undef STORM_T16_MUL
#undef STORM_T16_FETCH

    // ---- epilogue: C/D map of the 16x16 form: col = lane & 15, row = 4 * (lane >> 4) + reg
    const bool rect = j_count != 0;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const uint32_t j = b_row0 + (uint32_t)n * 16u + (lane & 15u);
        const bool j_ok = rect ? (j >= j_base && j - j_base < j_count) : j < n_cols;
        const uint32_t nj = (row_counts && j_ok) ? row_counts[j] : 0u;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t i = a_row0 + wave * 32u + (uint32_t)m * 16u + 4u * (lane >> 4) + (uint32_t)r;
                if (j_ok && i >= i_lo && i < n_rows && (rect || i < j)) {
                    const uint32_t c = (uint32_t)acc[m][n][r];
                    uint32_t* dst = &out[(uint64_t)(i - i_lo) * ld + (j - j_base)];
                    if (item_idx < split_from) {
                        *dst = row_counts ? row_counts[i] + nj - and_weight * c : c;
                    } else {  // partial over k: the n_i + n_j term once, mod 2^32 throughout
                        const uint32_t once = (row_counts && it.stage0 == 0) ? row_counts[i] + nj : 0u;
                        atomicAdd(dst, row_counts ? once - and_weight * c : c);
                    }
                }
            }
    }
}
