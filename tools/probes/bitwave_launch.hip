// tools/probes/bitwave_launch.hip — host side of K2w: ensure_bitwave, launch_pairw_bitwave.
// TOOLS BUILD ONLY (`make -C stormbitmaps_amd/csrc probes` -> libstorm_hip_probes.so): this file is a fragment of
// stormbitmaps_amd/csrc/storm_hip_mfma.hip, included there under -DSTORM_HIP_PROBES at the place the code used to
// stand; it is not part of the shipped library.

// K2w: per-wave stage words of the same plan (bitwave_kernel). One buffer: first[4 G + 1] | words.
static int ensure_bitwave(storm_hip_ctx_t* ctx, const std::vector<RowRange>& ranges, uint32_t n_kslices,
                          uint32_t shard_rank, uint32_t shard_count, uint64_t pitch) {
    const uint64_t key[4] = {ranges_hash(ranges) ^ (pitch * 0x9e3779b97f4a7c15ull) ^ 0x77aa77aa77aa77aaull ^
                                 ((uint64_t)ctx->k2_stream_w3_2 * 0xc2b2ae3d27d4eb4full),
                             n_kslices, ((uint64_t)shard_rank << 32) | shard_count,
                             ((uint64_t)(ctx->k2_stream_groups_per_cu & 0xff) << 32) |
                                 ((uint64_t)(ctx->k2_stream_min_piece & 0xffff) << 16) |
                                 (uint64_t)(ctx->k2_stream_min_run & 0xffff) |
                                 ((uint64_t)(ctx->k2_stream_w3_1 & 0x3ff) << 40)};
    if (ctx->d_bitfirst && !memcmp(key, ctx->bit_key, sizeof(key))) return STORM_HIP_OK;
    BitstreamShaping sh;
    sh.groups_per_cu = ctx->k2_stream_groups_per_cu;
    sh.min_piece = std::max(1, ctx->k2_stream_min_piece);
    sh.min_run = std::max(1, ctx->k2_stream_min_run);
    sh.w3_1 = ctx->k2_stream_w3_1;
    sh.w3_2 = ctx->k2_stream_w3_2;
    BitstreamPlan plan;
    build_bitstream(sh, ranges, n_kslices, shard_rank, shard_count, (uint32_t)std::max(1, ctx->n_cus), pitch, plan);
    if (!ranges.empty() && ranges.back().r1 * pitch / 64 + n_kslices + 4 * pitch >= (1ull << 30)) {
        set_error("K2w: the matrix is beyond the 30-bit stage addresses (64-byte units)");
        return STORM_HIP_EINVAL;
    }
    std::vector<uint32_t> first, words;
    first.reserve((size_t)plan.groups * 4 + 1);
    uint32_t longest = 0;
    for (uint32_t w = 0; w < plan.groups; ++w)
        for (uint32_t v = 0; v < 4; ++v) {
            first.push_back((uint32_t)words.size());
            for (uint32_t si = plan.first[w]; si < plan.first[w + 1]; ++si) {
                const BitSeg& sg = plan.segs[si];
                auto base = [&](uint32_t blk) { return (uint32_t)((uint64_t)sg.ks + (uint64_t)blk * pitch); };
                const uint32_t wm = (v + (sg.flags >> 8)) & 3u;
                const bool diag = (sg.flags & kBsDiag) != 0u;
                words.push_back(base(sg.a_blk + wm) | kBwOwn | (diag ? kBwMul : 0u));
                if (diag)
                    for (uint32_t b = wm + 1; b < 4u; ++b) words.push_back(base(sg.a_blk + b));
                for (uint32_t i = 0; i < sg.n_b; ++i) {
                    uint32_t rel = sg.b_first + i;
                    if (rel >= sg.range_nb) rel -= sg.range_nb;
                    words.push_back(base(sg.range_b0 + rel));
                }
            }
            longest = std::max(longest, (uint32_t)words.size() - first.back());
        }
    first.push_back((uint32_t)words.size());
    if (longest > kBsMaxStages) {
        set_error("K2w: a wave of %u stages exceeds the exact range of its accumulators", longest);
        return STORM_HIP_EINVAL;
    }
    std::vector<uint32_t> packed(first);
    packed.insert(packed.end(), words.begin(), words.end());
    const size_t bytes = std::max<size_t>(packed.size(), 1) * sizeof(uint32_t);
    if (bytes > ctx->bitfirst_capacity) {
        if (ctx->d_bitfirst) STORM_HIP_TRY(hipFree(ctx->d_bitfirst));
        ctx->d_bitfirst = nullptr;
        ctx->bitfirst_capacity = 0;
        STORM_HIP_TRY(hipMalloc(&ctx->d_bitfirst, bytes));
        ctx->bitfirst_capacity = bytes;
    }
    STORM_HIP_TRY(hipMemcpyAsync(ctx->d_bitfirst, packed.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
    STORM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->n_bit_groups = plan.groups;
    ctx->bit_stages = words.size();
    ctx->bit_max_stages = longest;
    ctx->n_bit_segs = (uint32_t)plan.segs.size();
    memcpy(ctx->bit_key, key, sizeof(key));
    return STORM_HIP_OK;
}

int launch_pairw_bitwave(storm_hip_ctx_t* ctx, const uint64_t* X, uint64_t pitch,
                         const std::vector<RowRange>& ranges, uint32_t n_kslices, uint32_t shard_rank,
                         uint32_t shard_count, uint64_t* d_total) {
    if (pitch * (uint64_t)kStripBRows >= (1ull << 32) || pitch % 64 != 0) {
        set_error("K2w: rows of %llu bytes are outside the bit-operand stream's 32-bit DMA offsets",
                  (unsigned long long)pitch);
        return STORM_HIP_EINVAL;
    }
    if (int rc = ensure_bitwave(ctx, ranges, n_kslices, shard_rank, shard_count, pitch)) return rc;
    ctx->n_items = 0;
    memset(ctx->items_key, 0xff, sizeof(ctx->items_key));
    ctx->last_info[0] = ctx->n_bit_groups;
    ctx->last_info[1] = ctx->bit_max_stages;
    ctx->last_info[2] = 1;
    ctx->last_info[3] = ctx->n_bit_segs;
    if (ctx->n_bit_groups == 0) {
        STORM_HIP_TRY(hipMemsetAsync(d_total, 0, sizeof(uint64_t), ctx->stream));
        return STORM_HIP_OK;
    }
    const uint32_t G = ctx->n_bit_groups, cus = (uint32_t)std::max(1, ctx->n_cus);
    int ring = ctx->k2_wave_ring;
    if (ring == 0) ring = G <= cus ? 8 : G <= 2 * cus ? 4 : 3;
    const uint32_t* first = static_cast<const uint32_t*>(ctx->d_bitfirst);
    const uint32_t* words = first + 4 * (size_t)G + 1;
    kernel_time_mark(ctx);
#define STORM_BW_LAUNCH(R)                                                                                      \
    hipLaunchKernelGGL(bitwave_kernel<R>, dim3(G), dim3(kStripThreads), 0, ctx->stream,                         \
                       reinterpret_cast<const uint8_t*>(X), pitch, first, words, ctx->d_slots,                  \
                       reinterpret_cast<unsigned long long*>(d_total))
    switch (ring) {
        case 8: STORM_BW_LAUNCH(8); break;
        case 6: STORM_BW_LAUNCH(6); break;
        case 4: STORM_BW_LAUNCH(4); break;
        default: STORM_BW_LAUNCH(3); break;
    }
#undef STORM_BW_LAUNCH
    kernel_time_mark(ctx);
    STORM_HIP_TRY(hipGetLastError());
    return STORM_HIP_OK;
}

