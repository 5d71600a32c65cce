// tools/probes/strip_fp4_32.hip — 32x32x64 strips (plain, wide, persistent) and their timing probes.
// TOOLS BUILD ONLY (`make -C stormbitmaps_amd/csrc probes` -> libstorm_hip_probes.so): this file is a fragment of
// stormbitmaps_amd/csrc/storm_hip_mfma.hip, included there under -DSTORM_HIP_PROBES at the place the code used to
// stand; it is not part of the shipped library.

template <int kStripRing, int kProbe = 0, int kMB = 2, bool kPersist = false>
__global__ __launch_bounds__(kStripThreads, (kMB == 2 ? 4 : 2)) void strip_fp4_kernel(
    const uint8_t* __restrict__ X4, uint64_t row_bytes, const StripItem* __restrict__ items,
    unsigned long long* __restrict__ slots, unsigned long long* __restrict__ trace = nullptr,
    StripQueues queues = {}, unsigned int* __restrict__ heads = nullptr) {
    // the ring, plus one word through which thread 0 hands the next item to the other waves
    __shared__ __attribute__((aligned(1024))) uint8_t lds_raw[kStripRing * kStripStageBytes + (kPersist ? 64 : 0)];
    auto lds = reinterpret_cast<uint8_t(*)[kStripStageBytes]>(lds_raw);

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wm = wave;  // waves stacked along A; every wave multiplies all 64 B rows
    uint32_t item_idx = blockIdx.x;
    const uint32_t my_queue = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;  // XCC_ID
    for (;;) {
    if constexpr (kPersist) {
        const uint32_t word = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&lds[kStripRing][0];
        if (tid == 0) {
            uint32_t got = kNoItem;
            for (uint32_t r = 0; r < 8u && got == kNoItem; ++r) {  // own queue first, then steal
                const uint32_t q = (my_queue + r) & 7u;
                if (queues.count[q] == 0) continue;
                const uint32_t i = atomicAdd(&heads[q], 1u);
                if (i < queues.count[q]) got = queues.base[q] + i;
            }
            asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(word), "v"(got) : "memory");
        }
        // also: every wave is done with the previous item's ring before the next DMA lands
        __builtin_amdgcn_s_barrier();
        uint32_t got_v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(got_v) : "v"(word) : "memory");
        item_idx = __builtin_amdgcn_readfirstlane(got_v);
        if (item_idx == kNoItem) break;
    }
    // kProbe bit 3: schedule trace — per item {start, end (100 MHz counter), HW_ID, XCC_ID}
    unsigned long long t_start = 0, t_ready = 0, t_diag = 0, t_main = 0;
    if constexpr ((kProbe & 8) != 0) t_start = __builtin_amdgcn_s_memrealtime();
    constexpr uint32_t kWaveRows = 32u * kMB;                 // A rows of one wave
    constexpr uint32_t kATile = kWaveRows * kStripWaves;      // A rows of the workgroup
    constexpr uint32_t kBPW = kMB / 2;                        // 64-row B blocks per wave's rows
    static_assert(kMB == 2 || kMB == 4, "A rows per wave: 64 or 128");
    const StripItem it = items[item_idx];
    const uint64_t kbyte = (uint64_t)it.ks * kStripRowBytes;
    // Stage order: first (if it.diag) the 4 blocks of the A tile itself — wave wm contributes
    // nothing for blocks before its own rows, the strict upper triangle of its own 64x64 block,
    // and everything after — then the later blocks from the LAST one down.
    const uint32_t D = it.diag ? kATile / (uint32_t)kStripBRows : 0u;
    const uint32_t T = D + (it.j1 - it.j0);

    // B stage = 8 LDS-DMA instructions of 8 rows x 128 B; wave w issues instructions w and
    // w + 4. Piece p = n*64 + lane is row p/8, 16-byte slot (p%8) ^ ((row/2)%8) of the stage;
    // stepping n by 4 adds 32 rows and leaves the swizzle unchanged, so one per-lane offset
    // serves both and the rest is a scalar base (host guarantees 64 * row_bytes < 2^32).
    const uint32_t r0 = (wave * 64u + lane) >> 3;
    const uint32_t goff0 = r0 * (uint32_t)row_bytes + (((lane & 7u) ^ ((r0 >> 1) & 7u)) * 16u);
    auto issue = [&](uint32_t t) {
        if constexpr ((kProbe & 2) != 0) return;
        // B blocks are walked from the LAST one down: all items of one k-slice then start on
        // the same block at the same time and stay aligned (the shorter ones just stop
        // earlier), so one of them misses in L2 and the others hit. Walking up from j0, item I
        // trails item I+1 by four stages and the slice was re-fetched ~7x (profiles/r01_e_*).
        const uint32_t blk = t < D ? it.a_row0 / (uint32_t)kStripBRows + t
                                   : it.j1 - 1u - (t - D);
        const uint8_t* base = X4 + (uint64_t)(blk * (uint32_t)kStripBRows) * row_bytes + kbyte;
        uint8_t* dst = lds[t % kStripRing] + wave * 1024u;
        // LDS-DMA as buffer loads (scalar descriptor of the stage + 32-bit lane offsets), not
        // global_load_lds with 64-bit lane addresses: beside MFMA bursts the latter costs the
        // wave 18.3 ns per MFMA at 3-4 waves per SIMD, the former 14.2 — as much as no load at
        // all (tools/ubench_feed, profiles/r01_h_ubench_feed.txt).
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, -1, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)dst, 16, (int)goff0, 0, 0, 0);
        if constexpr (kStripPieces == 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + 4096u), 16, (int)goff0,
                                                     (int)(32u * (uint32_t)row_bytes), 0, 0);
    };

    // A fragments first (older in the VMEM queue than the DMAs, so waiting for them does not
    // drain the ring), then the first stages of B
    v4i a[4][kMB];
    {
        const uint8_t* ap = X4 + (uint64_t)(it.a_row0 + wm * kWaveRows + (lane & 31u)) *
                                     row_bytes + kbyte + (lane >> 5) * 16u;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int m = 0; m < kMB; ++m)
                a[kk][m] = *reinterpret_cast<const v4i*>(ap + (uint64_t)m * 32u * row_bytes +
                                                         kk * 32);
    }
#pragma unroll
    for (uint32_t t = 0; t < kStripRing - 1; ++t)
        if (t < T) issue(t);

    v16f acc[kMB][2];
#pragma unroll
    for (int m = 0; m < kMB; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = v16f{};

    // per-lane LDS byte offset of its 16-byte B piece inside a stage, per k-step
    const uint32_t swz = (lane >> 1) & 7u;
    const uint32_t lds_base =
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&lds[0][0];
    uint32_t boff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        boff[kk] = lds_base + (lane & 31u) * kStripRowBytes +
                   ((((uint32_t)kk * 2u + (lane >> 5)) ^ swz) * 16u);
    // The B-fragment reads are inline asm with hand-counted lgkmcnt: the two ds_read_b128 of
    // k-step k+1 are issued BEFORE the 4 MFMAs of k-step k and retired by lgkmcnt(2) ("all but
    // the 2 youngest") one step later; sched_barrier(0) keeps hipcc from moving MFMAs across
    // the asm (cdna guide §5.4 rule 18). hipcc's own waits for ds_reads it can see are
    // lgkmcnt(0) right behind the read.
    auto fetch = [&](uint32_t t, int kk, v4i (&b)[2]) {
        if constexpr ((kProbe & 4) != 0) {
            asm volatile("" : "+v"(b[0]), "+v"(b[1]));  // keep the fragments opaque, read nothing
            return;
        }
        const uint32_t addr = boff[kk] + (t % kStripRing) * kStripStageBytes;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096"
                     : "=&v"(b[0]), "=&v"(b[1])
                     : "v"(addr));
    };
    auto multiply = [&](int kk, const v4i (&b)[2]) {
        if constexpr ((kProbe & 16) != 0) __builtin_amdgcn_s_setprio(2);  // probe: MFMA bursts first
#pragma unroll
        for (int m = 0; m < kMB; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                    v8i{a[kk][m].x, a[kk][m].y, a[kk][m].z, a[kk][m].w, 0, 0, 0, 0},
                    v8i{b[n].x, b[n].y, b[n].z, b[n].w, 0, 0, 0, 0}, acc[m][n], 4, 4, 0, 0, 0, 0);
        if constexpr ((kProbe & 16) != 0) __builtin_amdgcn_s_setprio(0);
    };
#define STORM_LGKM(n)                                       \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0)
#define STORM_STEP(q, cur, nxt_fetch, wait) \
    nxt_fetch;                               \
    wait;                                    \
    multiply(q, cur);                        \
    __builtin_amdgcn_sched_barrier(0)

    // Make hipcc retire the A-fragment loads HERE (they are older than the DMAs, so its counted
    // wait leaves the ring in flight). Without this use it cannot prove inside the loop that the
    // loads are done and drains vmcnt(0) in front of the first MFMA of every stage.
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int m = 0; m < kMB; ++m) asm volatile("" ::"v"(a[kk][m]));

    // Ring protocol. Stages 0..2 are issued by the prologue; every wave issues 2 LDS-DMA
    // instructions per stage. retire(t, newest): wait until this wave's share of stage t has
    // landed — the younger stages issued so far (up to `newest`) may stay in flight, hence
    // vmcnt(2 x their number) — then the barrier makes every wave's share visible and proves that every wave is
    // done with the stages it read before arriving here.
    auto retire = [&](uint32_t t, uint32_t newest_issued) {
        const uint32_t younger = min(T - 1u, newest_issued) - t;  // issued stages newer than t
        if constexpr (kStripPieces == 2) {
            if (younger >= 3u) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (younger == 2u) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (younger == 1u) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (younger >= 3u) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (younger == 2u) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (younger == 1u) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr ((kProbe & 1) == 0) __builtin_amdgcn_s_barrier();
    };
    static_assert(kStripRing >= 3 && kStripRing <= 5, "vmcnt cases above cover rings of 3..5");

    v4i b0[2] = {}, b1[2] = {};
    uint32_t t = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // scalar loads of the item record
    if constexpr ((kProbe & 8) != 0) {  // trace only: when are the A fragments in?
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t_ready = __builtin_amdgcn_s_memrealtime();
    }
    // ---- the A tile's own 4 blocks (strict upper triangle), not software-pipelined: wave wm
    //      skips the blocks before its own rows (only pairs with i > j there), masks its own
    //      64x64 block, and takes the later ones whole. Kept apart from the main loop so that
    //      the main loop stays free of selects (hipcc turned an `if (x) frag = 0` inside it into
    //      v_cndmask on every k-step, which costs MFMA issue slots).
#pragma unroll 1
    for (; t < D; ++t) {
        // (stages up to t + kStripRing - 2 are in the ring; everyone finished stage t-1, so its
        //  buffer may take stage t + kStripRing - 1)
        retire(t, t + kStripRing - 2);
        if (t + kStripRing - 1 < T) issue(t + kStripRing - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (t >= kBPW * wm) {
            fetch(t, 0, b0);
            fetch(t, 1, b1);
            STORM_LGKM(2);
            multiply(0, b0);
            __builtin_amdgcn_sched_barrier(0);
            fetch(t, 2, b0);
            STORM_LGKM(2);
            multiply(1, b1);
            __builtin_amdgcn_sched_barrier(0);
            fetch(t, 3, b1);
            STORM_LGKM(2);
            multiply(2, b0);
            __builtin_amdgcn_sched_barrier(0);
            STORM_LGKM(0);
            multiply(3, b1);
            __builtin_amdgcn_sched_barrier(0);
            // Block t = kBPW * wm + q holds the q-th 64 rows of this wave (MFMA blocks 2q and
            // 2q+1). Up to here blocks 2q.. of the accumulators have seen nothing but this
            // stage (the earlier ones were cleared below), so the pairs with i >= j can be
            // cleared in place: the blocks of later rows entirely, (2q+1, 0) entirely, and the
            // two 32x32 blocks on the diagonal down to their strict upper triangle.
            // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#pragma unroll
            for (int q = 0; q < (int)kBPW; ++q) {
                if (t != kBPW * wm + q) continue;
#pragma unroll
                for (int m = 2 * q + 2; m < kMB; ++m) acc[m][0] = acc[m][1] = v16f{};
                acc[2 * q + 1][0] = v16f{};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const bool keep = row < (lane & 31u);
                    acc[2 * q][0][r] = keep ? acc[2 * q][0][r] : 0.0f;
                    acc[2 * q + 1][1][r] = keep ? acc[2 * q + 1][1][r] : 0.0f;
                }
            }
        }
    }
    // ---- later blocks, software-pipelined: stage t is retired one iteration early so that the
    //      fragments of its first k-step are fetched while stage t-1 is still being multiplied;
    //      iteration t therefore retires stage t+1, and refills the ring with stage t+3 (whose
    //      buffer held stage t-1: every wave finished it before this iteration's barrier).
    if constexpr ((kProbe & 8) != 0) t_diag = __builtin_amdgcn_s_memrealtime();
    if (t < T) {
        retire(t, t + kStripRing - 2);
        fetch(t, 0, b0);
        for (; t < T; ++t) {
            if (t + 1 < T) retire(t + 1, t + kStripRing - 2);  // the refill comes after the barrier
            else __builtin_amdgcn_s_barrier();
            if (t + kStripRing - 1 < T) issue(t + kStripRing - 1);
            __builtin_amdgcn_sched_barrier(0);
            STORM_STEP(0, b0, fetch(t, 1, b1), STORM_LGKM(2));
            STORM_STEP(1, b1, fetch(t, 2, b0), STORM_LGKM(2));
            STORM_STEP(2, b0, fetch(t, 3, b1), STORM_LGKM(2));
            // (after the last stage this re-reads k-step 0 of the same stage: never consumed;
            //  keeping the loop body branch-free lets hipcc accumulate in place — with a
            //  two-armed tail it ping-ponged between two accumulator sets and spilled)
            STORM_STEP(3, b1, fetch(t + 1 < T ? t + 1 : t, 0, b0), STORM_LGKM(2));
        }
        STORM_LGKM(0);
    }
#undef STORM_STEP
#undef STORM_LGKM
    if constexpr ((kProbe & 8) != 0) t_main = __builtin_amdgcn_s_memrealtime();

    uint64_t mine = 0;
#pragma unroll
    for (int m = 0; m < kMB; ++m) {  // one 32x64 strip at a time stays below 2^32
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
        atomicAdd(&slots[(item_idx * (uint32_t)kStripWaves + wave) & (kSlots - 1)],
                  (unsigned long long)mine);
    if constexpr ((kProbe & 8) != 0) {
        if (tid == 0 && trace) {
            trace[item_idx * 4ull + 0] = t_start;
            trace[item_idx * 4ull + 1] = __builtin_amdgcn_s_memrealtime();
            // phase marks relative to the start, 16 bits each in 10 ns units:
            // A fragments + first stages in | diagonal phase done | main loop done
            trace[item_idx * 4ull + 2] = ((t_ready - t_start) & 0xffffull) |
                                         (((t_diag - t_start) & 0xffffull) << 16) |
                                         (((t_main - t_start) & 0xffffull) << 32);
            trace[item_idx * 4ull + 3] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
        }
    }
    if constexpr (!kPersist) break;
    }  // items
}
