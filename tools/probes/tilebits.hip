// tools/probes/tilebits.hip — bit-operand output kernel, one wave per SIMD (3 % slower than tilebits8_kernel).
// TOOLS BUILD ONLY (`make -C stormbitmaps_amd/csrc probes` -> libstorm_hip_probes.so): this file is a fragment of
// stormbitmaps_amd/csrc/storm_hip_mfma.hip, included there under -DSTORM_HIP_PROBES at the place the code used to
// stand; it is not part of the shipped library.

__global__ __launch_bounds__(kTbThreads, 1) void tilebits_kernel(
    TileOperands ops, const MfmaItem* __restrict__ items, uint32_t* __restrict__ out, uint64_t ld,
    uint32_t n_rows, const uint32_t* __restrict__ row_counts, uint32_t and_weight, uint32_t j_base,
    uint32_t j_count, uint32_t split_from, uint32_t i_lo, uint32_t n_cols) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[kTbRing][kTbStageBytes];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wa = wave & 1u, wb = wave >> 1;
    const uint32_t item_idx = blockIdx.x;
    const MfmaItem it = items[item_idx];
    const uint32_t a_row0 = (uint32_t)it.I * kTile, b_row0 = (uint32_t)it.J * kTile;
    const uint32_t S = it.n_stages / 4u;                  // items count 128-bit stages; cuts fall on 4
    const uint32_t kbyte0 = it.stage0 * 16u;              // byte of the row where this item starts
    const uint32_t pitch = (uint32_t)ops.pitch;

    // operand windows: base of the tile's first row and the bytes of it that exist
    auto window = [&](uint32_t v0, const uint8_t*& base, uint32_t& bytes) {
        const bool second = v0 >= ops.split;
        const uint32_t r0 = second ? v0 - ops.split : v0;
        const uint32_t have = second ? ops.rows_b : ops.rows_a;
        const uint32_t rows = have > r0 ? min(have - r0, (uint32_t)kTile) : 0u;
        base = (second ? ops.xb : ops.xa) + (uint64_t)r0 * ops.pitch;
        bytes = rows * pitch;
    };
    const uint8_t *a_base, *b_base;
    uint32_t a_bytes, b_bytes;
    window(a_row0, a_base, a_bytes);
    window(b_row0, b_base, b_bytes);

    // DMA: an image is 16 wave-instructions of 1 KiB (16 rows each). Piece p of a stage (one per class
    // phase): instruction w + 4 (p % 4) of the A image (p < 4) or of the B image. Lane L fills row
    // L / 4 of the instruction, physical slot L % 4 = logical slot (L % 4) ^ ((L / 16) % 4).
    const uint32_t voff0 = (wave * 16u + (lane >> 2)) * pitch + (((lane & 3u) ^ ((lane >> 4) & 3u)) * 16u);
    auto issue_piece = [&](uint32_t s, uint32_t p) {
        const uint32_t koff = kbyte0 + s * kTbRowBytes;
        const bool second = p >= 4u;
        // (past the last stage the piece is still issued, with an empty range: a branch-free loop
        //  body keeps every piece where it is written, and the count below stays the same)
        const uint32_t bytes = s < S ? (second ? b_bytes : a_bytes) : 0u;
        // (a window with rows has koff < pitch <= bytes; written as a select on `bytes != 0`, not as
        //  a saturating subtraction, which has no scalar form and turns the descriptor divergent)
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>((second ? b_base : a_base) + koff), 0, bytes ? bytes - koff : 0u, 0x00020000);
        uint8_t* dst = lds[s % kTbRing] + (second ? kTbImageBytes : 0) + (wave + 4u * (p & 3u)) * 1024u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)dst, 16, (int)(voff0 + (p & 3u) * 64u * pitch), 0, 0, 0);
    };
    auto issue = [&](uint32_t s) {
#pragma unroll
        for (uint32_t p = 0; p < 8; ++p) issue_piece(s, p);
    };

    v16f acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = v16f{};

    // fragment of block t in k-group g: row 32 t + (lane & 31), logical slot 2 g + (lane >> 5)
    const uint32_t lds_base =
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&lds[0][0];
    const uint32_t slot = (lane >> 5) ^ (((lane & 31u) >> 2) & 3u);
    const uint32_t a_frag0 = lds_base + (wa * 128u + (lane & 31u)) * kTbRowBytes + slot * 16u;
    const uint32_t a_frag1 = lds_base + (wa * 128u + (lane & 31u)) * kTbRowBytes + (slot ^ 2u) * 16u;
    const uint32_t b_delta = kTbImageBytes + wb * 128u * kTbRowBytes - wa * 128u * kTbRowBytes;

    issue(0);
    issue(1);
    issue(2);

    // (every look-ahead read of these kernels feeds a loop-carried value, alive to the wait behind the loop:
    //  no read's output is dead in hipcc's eyes while it is still in flight — see STORM_SB_KEEP below)
#define STORM_TB_FETCH(dst, addr, n) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"((n) * 32 * kTbRowBytes))
#define STORM_TB_MUL(C, m, n, av, bv)                                                               \
    acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                                    \
        v8i{av[m].x, av[m].y, av[m].z, av[m].w, 0, 0, 0, 0}, v8i{bv.x, bv.y, bv.z, bv.w, 0, 0, 0, 0}, \
        acc[m][n], 4, 4, 0, tb_scale<C>(), 0, tb_scale<C>())

    v4i xa[4], xb[4], ya[4], yb[4];  // bits of the k-group in use / of the next one (x: even groups)
    v4i ao[4], an[4], bo, bn = {};   // inflated A blocks of the running / next class phase, B block
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // scalar loads of the item record
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const uint32_t b0 = a_frag0 + b_delta;
        STORM_TB_FETCH(xa[0], a_frag0, 0);
        STORM_TB_FETCH(xa[1], a_frag0, 1);
        STORM_TB_FETCH(xa[2], a_frag0, 2);
        STORM_TB_FETCH(xa[3], a_frag0, 3);
        STORM_TB_FETCH(xb[0], b0, 0);
        STORM_TB_FETCH(xb[1], b0, 1);
        STORM_TB_FETCH(xb[2], b0, 2);
        STORM_TB_FETCH(xb[3], b0, 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) ao[m] = tb_inflate<0>(xa[m]);
        bo = tb_inflate<0>(xb[0]);
    }
    // At the top of stage s the wave's DMA pieces of stage s + 1 must have landed (only stage
    // s + 2's eight may stay in flight): the second k-group reads one k-group ahead, into it. The
    // barrier makes that true of every wave's share and says that every wave is done with stage
    // s - 1, whose slot stage s + 3 takes.
    for (uint32_t s = 0; s < S; ++s) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // (after the last stage the look-ahead re-reads the same stage: never consumed)
        const uint32_t cur = (s % kTbRing) * kTbStageBytes;
        const uint32_t nxs = ((s + 1 < S ? s + 1 : s) % kTbRing) * kTbStageBytes;
        const uint32_t a1 = a_frag1 + cur, b1 = a1 + b_delta;
        const uint32_t a0n = a_frag0 + nxs, b0n = a0n + b_delta;
        const uint32_t dma_stage = s + kTbRing - 1;
        // k-group 0, class 0
        issue_piece(dma_stage, 0);
        STORM_TB_FETCH(ya[0], a1, 0);
        STORM_TB_FETCH(ya[1], a1, 1);
        STORM_TB_FETCH(ya[2], a1, 2);
        STORM_TB_FETCH(ya[3], a1, 3);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(0, 0, 0, ao, bo);
        bn = tb_inflate<0>(xb[1]);
        STORM_TB_MUL(0, 1, 0, ao, bo);
        an[0] = tb_inflate<1>(xa[0]);
        STORM_TB_MUL(0, 2, 0, ao, bo);
        STORM_TB_MUL(0, 3, 0, ao, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(0, 0, 1, ao, bn);
        bo = tb_inflate<0>(xb[2]);
        STORM_TB_MUL(0, 1, 1, ao, bn);
        an[1] = tb_inflate<1>(xa[1]);
        STORM_TB_MUL(0, 2, 1, ao, bn);
        STORM_TB_MUL(0, 3, 1, ao, bn);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(0, 0, 2, ao, bo);
        bn = tb_inflate<0>(xb[3]);
        STORM_TB_MUL(0, 1, 2, ao, bo);
        an[2] = tb_inflate<1>(xa[2]);
        STORM_TB_MUL(0, 2, 2, ao, bo);
        STORM_TB_MUL(0, 3, 2, ao, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(0, 0, 3, ao, bn);
        bo = tb_inflate<1>(xb[0]);
        STORM_TB_MUL(0, 1, 3, ao, bn);
        an[3] = tb_inflate<1>(xa[3]);
        STORM_TB_MUL(0, 2, 3, ao, bn);
        STORM_TB_MUL(0, 3, 3, ao, bn);
        __builtin_amdgcn_sched_barrier(0);
        // k-group 0, class 1
        issue_piece(dma_stage, 1);
        STORM_TB_FETCH(yb[0], b1, 0);
        STORM_TB_FETCH(yb[1], b1, 1);
        STORM_TB_FETCH(yb[2], b1, 2);
        STORM_TB_FETCH(yb[3], b1, 3);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(1, 0, 0, an, bo);
        bn = tb_inflate<1>(xb[1]);
        STORM_TB_MUL(1, 1, 0, an, bo);
        ao[0] = tb_inflate<2>(xa[0]);
        STORM_TB_MUL(1, 2, 0, an, bo);
        STORM_TB_MUL(1, 3, 0, an, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(1, 0, 1, an, bn);
        bo = tb_inflate<1>(xb[2]);
        STORM_TB_MUL(1, 1, 1, an, bn);
        ao[1] = tb_inflate<2>(xa[1]);
        STORM_TB_MUL(1, 2, 1, an, bn);
        STORM_TB_MUL(1, 3, 1, an, bn);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(1, 0, 2, an, bo);
        bn = tb_inflate<1>(xb[3]);
        STORM_TB_MUL(1, 1, 2, an, bo);
        ao[2] = tb_inflate<2>(xa[2]);
        STORM_TB_MUL(1, 2, 2, an, bo);
        STORM_TB_MUL(1, 3, 2, an, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(1, 0, 3, an, bn);
        bo = tb_inflate<2>(xb[0]);
        STORM_TB_MUL(1, 1, 3, an, bn);
        ao[3] = tb_inflate<2>(xa[3]);
        STORM_TB_MUL(1, 2, 3, an, bn);
        STORM_TB_MUL(1, 3, 3, an, bn);
        __builtin_amdgcn_sched_barrier(0);
        // k-group 0, class 2
        issue_piece(dma_stage, 2);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(2, 0, 0, ao, bo);
        bn = tb_inflate<2>(xb[1]);
        STORM_TB_MUL(2, 1, 0, ao, bo);
        an[0] = tb_inflate<3>(xa[0]);
        STORM_TB_MUL(2, 2, 0, ao, bo);
        STORM_TB_MUL(2, 3, 0, ao, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(2, 0, 1, ao, bn);
        bo = tb_inflate<2>(xb[2]);
        STORM_TB_MUL(2, 1, 1, ao, bn);
        an[1] = tb_inflate<3>(xa[1]);
        STORM_TB_MUL(2, 2, 1, ao, bn);
        STORM_TB_MUL(2, 3, 1, ao, bn);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(2, 0, 2, ao, bo);
        bn = tb_inflate<2>(xb[3]);
        STORM_TB_MUL(2, 1, 2, ao, bo);
        an[2] = tb_inflate<3>(xa[2]);
        STORM_TB_MUL(2, 2, 2, ao, bo);
        STORM_TB_MUL(2, 3, 2, ao, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(2, 0, 3, ao, bn);
        bo = tb_inflate<3>(xb[0]);
        STORM_TB_MUL(2, 1, 3, ao, bn);
        an[3] = tb_inflate<3>(xa[3]);
        STORM_TB_MUL(2, 2, 3, ao, bn);
        STORM_TB_MUL(2, 3, 3, ao, bn);
        __builtin_amdgcn_sched_barrier(0);
        // k-group 0, class 3
        issue_piece(dma_stage, 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(3, 0, 0, an, bo);
        bn = tb_inflate<3>(xb[1]);
        STORM_TB_MUL(3, 1, 0, an, bo);
        ao[0] = tb_inflate<0>(ya[0]);
        STORM_TB_MUL(3, 2, 0, an, bo);
        STORM_TB_MUL(3, 3, 0, an, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(3, 0, 1, an, bn);
        bo = tb_inflate<3>(xb[2]);
        STORM_TB_MUL(3, 1, 1, an, bn);
        ao[1] = tb_inflate<0>(ya[1]);
        STORM_TB_MUL(3, 2, 1, an, bn);
        STORM_TB_MUL(3, 3, 1, an, bn);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(3, 0, 2, an, bo);
        bn = tb_inflate<3>(xb[3]);
        STORM_TB_MUL(3, 1, 2, an, bo);
        ao[2] = tb_inflate<0>(ya[2]);
        STORM_TB_MUL(3, 2, 2, an, bo);
        STORM_TB_MUL(3, 3, 2, an, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(3, 0, 3, an, bn);
        bo = tb_inflate<0>(yb[0]);
        STORM_TB_MUL(3, 1, 3, an, bn);
        ao[3] = tb_inflate<0>(ya[3]);
        STORM_TB_MUL(3, 2, 3, an, bn);
        STORM_TB_MUL(3, 3, 3, an, bn);
        __builtin_amdgcn_sched_barrier(0);
        // k-group 1, class 0
        issue_piece(dma_stage, 4);
        STORM_TB_FETCH(xa[0], a0n, 0);
        STORM_TB_FETCH(xa[1], a0n, 1);
        STORM_TB_FETCH(xa[2], a0n, 2);
        STORM_TB_FETCH(xa[3], a0n, 3);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(0, 0, 0, ao, bo);
        bn = tb_inflate<0>(yb[1]);
        STORM_TB_MUL(0, 1, 0, ao, bo);
        an[0] = tb_inflate<1>(ya[0]);
        STORM_TB_MUL(0, 2, 0, ao, bo);
        STORM_TB_MUL(0, 3, 0, ao, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(0, 0, 1, ao, bn);
        bo = tb_inflate<0>(yb[2]);
        STORM_TB_MUL(0, 1, 1, ao, bn);
        an[1] = tb_inflate<1>(ya[1]);
        STORM_TB_MUL(0, 2, 1, ao, bn);
        STORM_TB_MUL(0, 3, 1, ao, bn);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(0, 0, 2, ao, bo);
        bn = tb_inflate<0>(yb[3]);
        STORM_TB_MUL(0, 1, 2, ao, bo);
        an[2] = tb_inflate<1>(ya[2]);
        STORM_TB_MUL(0, 2, 2, ao, bo);
        STORM_TB_MUL(0, 3, 2, ao, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(0, 0, 3, ao, bn);
        bo = tb_inflate<1>(yb[0]);
        STORM_TB_MUL(0, 1, 3, ao, bn);
        an[3] = tb_inflate<1>(ya[3]);
        STORM_TB_MUL(0, 2, 3, ao, bn);
        STORM_TB_MUL(0, 3, 3, ao, bn);
        __builtin_amdgcn_sched_barrier(0);
        // k-group 1, class 1
        issue_piece(dma_stage, 5);
        STORM_TB_FETCH(xb[0], b0n, 0);
        STORM_TB_FETCH(xb[1], b0n, 1);
        STORM_TB_FETCH(xb[2], b0n, 2);
        STORM_TB_FETCH(xb[3], b0n, 3);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(1, 0, 0, an, bo);
        bn = tb_inflate<1>(yb[1]);
        STORM_TB_MUL(1, 1, 0, an, bo);
        ao[0] = tb_inflate<2>(ya[0]);
        STORM_TB_MUL(1, 2, 0, an, bo);
        STORM_TB_MUL(1, 3, 0, an, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(1, 0, 1, an, bn);
        bo = tb_inflate<1>(yb[2]);
        STORM_TB_MUL(1, 1, 1, an, bn);
        ao[1] = tb_inflate<2>(ya[1]);
        STORM_TB_MUL(1, 2, 1, an, bn);
        STORM_TB_MUL(1, 3, 1, an, bn);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(1, 0, 2, an, bo);
        bn = tb_inflate<1>(yb[3]);
        STORM_TB_MUL(1, 1, 2, an, bo);
        ao[2] = tb_inflate<2>(ya[2]);
        STORM_TB_MUL(1, 2, 2, an, bo);
        STORM_TB_MUL(1, 3, 2, an, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(1, 0, 3, an, bn);
        bo = tb_inflate<2>(yb[0]);
        STORM_TB_MUL(1, 1, 3, an, bn);
        ao[3] = tb_inflate<2>(ya[3]);
        STORM_TB_MUL(1, 2, 3, an, bn);
        STORM_TB_MUL(1, 3, 3, an, bn);
        __builtin_amdgcn_sched_barrier(0);
        // k-group 1, class 2
        issue_piece(dma_stage, 6);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(2, 0, 0, ao, bo);
        bn = tb_inflate<2>(yb[1]);
        STORM_TB_MUL(2, 1, 0, ao, bo);
        an[0] = tb_inflate<3>(ya[0]);
        STORM_TB_MUL(2, 2, 0, ao, bo);
        STORM_TB_MUL(2, 3, 0, ao, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(2, 0, 1, ao, bn);
        bo = tb_inflate<2>(yb[2]);
        STORM_TB_MUL(2, 1, 1, ao, bn);
        an[1] = tb_inflate<3>(ya[1]);
        STORM_TB_MUL(2, 2, 1, ao, bn);
        STORM_TB_MUL(2, 3, 1, ao, bn);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(2, 0, 2, ao, bo);
        bn = tb_inflate<2>(yb[3]);
        STORM_TB_MUL(2, 1, 2, ao, bo);
        an[2] = tb_inflate<3>(ya[2]);
        STORM_TB_MUL(2, 2, 2, ao, bo);
        STORM_TB_MUL(2, 3, 2, ao, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(2, 0, 3, ao, bn);
        bo = tb_inflate<3>(yb[0]);
        STORM_TB_MUL(2, 1, 3, ao, bn);
        an[3] = tb_inflate<3>(ya[3]);
        STORM_TB_MUL(2, 2, 3, ao, bn);
        STORM_TB_MUL(2, 3, 3, ao, bn);
        __builtin_amdgcn_sched_barrier(0);
        // k-group 1, class 3
        issue_piece(dma_stage, 7);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(3, 0, 0, an, bo);
        bn = tb_inflate<3>(yb[1]);
        STORM_TB_MUL(3, 1, 0, an, bo);
        ao[0] = tb_inflate<0>(xa[0]);
        STORM_TB_MUL(3, 2, 0, an, bo);
        STORM_TB_MUL(3, 3, 0, an, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(3, 0, 1, an, bn);
        bo = tb_inflate<3>(yb[2]);
        STORM_TB_MUL(3, 1, 1, an, bn);
        ao[1] = tb_inflate<0>(xa[1]);
        STORM_TB_MUL(3, 2, 1, an, bn);
        STORM_TB_MUL(3, 3, 1, an, bn);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(3, 0, 2, an, bo);
        bn = tb_inflate<3>(yb[3]);
        STORM_TB_MUL(3, 1, 2, an, bo);
        ao[2] = tb_inflate<0>(xa[2]);
        STORM_TB_MUL(3, 2, 2, an, bo);
        STORM_TB_MUL(3, 3, 2, an, bo);
        __builtin_amdgcn_sched_barrier(0);
        STORM_TB_MUL(3, 0, 3, an, bn);
        bo = tb_inflate<0>(xb[0]);
        STORM_TB_MUL(3, 1, 3, an, bn);
        ao[3] = tb_inflate<0>(xa[3]);
        STORM_TB_MUL(3, 2, 3, an, bn);
        STORM_TB_MUL(3, 3, 3, an, bn);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the empty pieces of the tail, too
#undef STORM_TB_MUL
#undef STORM_TB_FETCH

    // ---- epilogue: C/D map of the 32x32 form: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const bool rect = j_count != 0;
    {
        const uint32_t col0 = b_row0 - j_base;  // rect: j_base <= b_row0 is implied by the range test
        const bool interior =
            item_idx < split_from && a_row0 >= i_lo && a_row0 + kTile <= n_rows &&
            (rect ? (b_row0 >= j_base && col0 + kTile <= j_count) : (b_row0 + kTile <= n_cols && a_row0 != b_row0)) &&
            (ld & 3u) == 0 && ((uintptr_t)out & 15u) == 0;
        if (interior) {
            __builtin_amdgcn_s_barrier();  // every wave has left the ring
            tb_store_interior<4>(acc, &lds[0][0] + wave * 16384u,
                                 &out[(uint64_t)(a_row0 + wa * 128u - i_lo) * ld + col0 + wb * 128u], ld, lane,
                                 row_counts, a_row0 + wa * 128u, b_row0 + wb * 128u, and_weight);
            return;
        }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const uint32_t j = b_row0 + wb * 128u + (uint32_t)n * 32u + (lane & 31u);
        const bool j_ok = rect ? (j >= j_base && j - j_base < j_count) : j < n_cols;
        const uint32_t nj = (row_counts && j_ok) ? row_counts[j] : 0u;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t i = a_row0 + wa * 128u + (uint32_t)m * 32u + (uint32_t)((r & 3) + 8 * (r >> 2)) +
                                   4u * (lane >> 5);
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
