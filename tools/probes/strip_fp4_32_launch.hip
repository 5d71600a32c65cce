// tools/probes/strip_fp4_32_launch.hip — launch cases of the 32x32x64 strips, their timing probes and the schedule trace (inside launch_pairw_mfma_ranges' switch).
// TOOLS BUILD ONLY (`make -C stormbitmaps_amd/csrc probes` -> libstorm_hip_probes.so): this file is a fragment of
// stormbitmaps_amd/csrc/storm_hip_mfma.hip, included there under -DSTORM_HIP_PROBES at the place the code used to
// stand; it is not part of the shipped library.

                case 204:
                    hipLaunchKernelGGL((strip_fp4_kernel<4, 0, 2, true>), pgrid, sblock,
                                       (size_t)ctx->k2_lds_pad, ctx->stream,
                                       ctx->d_x4, pitch, sit, ctx->d_slots, nullptr, queues, heads);
                    break;
                case 218: {
                    const size_t need = (size_t)n_strip * 4 * sizeof(unsigned long long);
                    if (need > ctx->trace_capacity) {
                        if (ctx->d_trace) STORM_HIP_TRY(hipFree(ctx->d_trace));
                        ctx->d_trace = nullptr;
                        ctx->trace_capacity = 0;
                        STORM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_trace), need));
                        ctx->trace_capacity = need;
                    }
                    ctx->trace_items = n_strip;
                    ctx->trace_is_stream = false;
                    ctx->trace_is_stream = false;
                    hipLaunchKernelGGL((strip_fp4_kernel<4, 8, 2, true>), pgrid, sblock, 0, ctx->stream,
                                       ctx->d_x4, pitch, sit, ctx->d_slots, ctx->d_trace, queues, heads);
                    break;
                }
                case 103:
                    hipLaunchKernelGGL((strip_fp4_kernel<3, 0, 4>), sgrid, sblock, 0, ctx->stream,
                                       ctx->d_x4, pitch, sit, ctx->d_slots);
                    break;
                case 104:
                    hipLaunchKernelGGL((strip_fp4_kernel<4, 0, 4>), sgrid, sblock, 0, ctx->stream,
                                       ctx->d_x4, pitch, sit, ctx->d_slots);
                    break;
                case 105:
                    hipLaunchKernelGGL((strip_fp4_kernel<5, 0, 4>), sgrid, sblock, 0, ctx->stream,
                                       ctx->d_x4, pitch, sit, ctx->d_slots);
                    break;
#define STORM_WIDE_PROBE_CASE(n)                                                                  \
    case 110 + n:                                                                                 \
        hipLaunchKernelGGL((strip_fp4_kernel<4, n, 4>), sgrid, sblock, 0, ctx->stream, ctx->d_x4, \
                           pitch, sit, ctx->d_slots);                                         \
        break;
                STORM_WIDE_PROBE_CASE(1) STORM_WIDE_PROBE_CASE(2) STORM_WIDE_PROBE_CASE(4)
                STORM_WIDE_PROBE_CASE(6) STORM_WIDE_PROBE_CASE(7)
#undef STORM_WIDE_PROBE_CASE
                case 3:
                    hipLaunchKernelGGL(strip_fp4_kernel<3>, sgrid, sblock, 0, ctx->stream, ctx->d_x4,
                                       pitch, sit, ctx->d_slots);
                    break;
                case 5:
                    hipLaunchKernelGGL(strip_fp4_kernel<5>, sgrid, sblock, 0, ctx->stream, ctx->d_x4,
                                       pitch, sit, ctx->d_slots);
                    break;
#define STORM_PROBE_CASE(n)                                                                   \
    case 10 + n:                                                                              \
        hipLaunchKernelGGL((strip_fp4_kernel<4, n>), sgrid, sblock, 0, ctx->stream, ctx->d_x4, \
                           pitch, sit, ctx->d_slots);                                      \
        break;
                STORM_PROBE_CASE(1) STORM_PROBE_CASE(2) STORM_PROBE_CASE(4) STORM_PROBE_CASE(6)
                STORM_PROBE_CASE(7)
#undef STORM_PROBE_CASE
                case 26:  // probe: s_setprio around the MFMA bursts (results stay correct)
                    hipLaunchKernelGGL((strip_fp4_kernel<4, 16>), sgrid, sblock, 0, ctx->stream,
                                       ctx->d_x4, pitch, sit, ctx->d_slots);
                    break;
                case 18: {  // schedule trace (results stay correct)
                    const size_t need = (size_t)n_strip * 4 * sizeof(unsigned long long);
                    if (need > ctx->trace_capacity) {
                        if (ctx->d_trace) STORM_HIP_TRY(hipFree(ctx->d_trace));
                        ctx->d_trace = nullptr;
                        ctx->trace_capacity = 0;
                        STORM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_trace), need));
                        ctx->trace_capacity = need;
                    }
                    ctx->trace_items = n_strip;
                    ctx->trace_is_stream = false;
                    ctx->trace_is_stream = false;
                    hipLaunchKernelGGL((strip_fp4_kernel<4, 8>), sgrid, sblock, 0, ctx->stream,
                                       ctx->d_x4, pitch, sit, ctx->d_slots, ctx->d_trace);
                    break;
                }
