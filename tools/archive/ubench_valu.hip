// ubench_valu.hip — measures the VALU issue rate of the two instructions the dense kernel is
// made of (v_and_b32, v_bcnt_u32_b32 accumulating form) on gfx950, to price the kernel's
// compute ceiling (SURVEY.md §7 "v_bcnt full-rate is an assumption to microbenchmark").
// Usage: ubench_valu [waves_per_simd=4] [iters=20000]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                    \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

// 64 VALU instructions per loop trip, 8 independent accumulator chains
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed) {
    unsigned a[8], x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = 0;
        x[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u;
    }
    const unsigned m = seed | 0xF0F0F0F1u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) {  // bcnt only
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x[i]));
                } else if (MODE == 1) {  // and only (dependent chain per i, 8 chains)
                    asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[i]) : "v"(m));
                } else {  // the kernel's mix: and + accumulating bcnt (counts as 2 instrs; 4 pairs)
                    if (i < 4) {
                        unsigned t;
                        asm volatile("v_and_b32 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(x[i + 4]));
                        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(t));
                    }
                }
            }
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + x[i];
    if (s == 0xDEADBEEF) out[0] = s;
}

// ---- emulation of the dense kernel's inner loop (csrc/storm_hip.hip: popc_and / row_step /
// stage_compute), without any global traffic: MODE 3 = B words from registers, MODE 4 = B
// words from a 16 KiB LDS stage exactly like the product kernel reads them.
__device__ __forceinline__ unsigned popc_and(unsigned a_lo, unsigned a_hi, unsigned b_lo,
                                             unsigned b_hi, unsigned acc) {
    unsigned t;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(t) : "v"(a_lo & b_lo), "v"(acc));
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(acc) : "v"(a_hi & b_hi), "v"(t));
    return acc;
}

template <int MODE>
__global__ __launch_bounds__(256, 4) void kemu(unsigned* out, const unsigned long long* src,
                                               int iters) {
    __shared__ unsigned long long lds[32 * 64];
    const unsigned lane = threadIdx.x & 63u;
    unsigned a_lo[32], a_hi[32], acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const unsigned long long v = src[r * 64 + lane];
        a_lo[r] = (unsigned)v;
        a_hi[r] = (unsigned)(v >> 32);
    }
    for (int i = threadIdx.x; i < 32 * 64; i += 256) lds[i] = src[i] * 0x9E3779B97F4A7C15ull;
    __syncthreads();
    unsigned long long b = src[lane];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int jj = 0; jj < 32; ++jj) {
                const unsigned b_lo = (unsigned)b + jj, b_hi = (unsigned)(b >> 32);
#pragma unroll
                for (int r = 0; r < 32; ++r)
                    acc[r & 3] = popc_and(a_lo[r], a_hi[r], b_lo, b_hi, acc[r & 3]);
            }
            b += 0x100000001ull;
        } else {
            constexpr int G = 4;
            const unsigned long long* col = lds + lane;
            unsigned long long cur[G], nxt[G];
#pragma unroll
            for (int u = 0; u < G; ++u) cur[u] = col[u * 64];
#pragma unroll
            for (int g = 0; g < 32 / G; ++g) {
                if (g + 1 < 32 / G) {
#pragma unroll
                    for (int u = 0; u < G; ++u) nxt[u] = col[((g + 1) * G + u) * 64];
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const unsigned b_lo = (unsigned)cur[u], b_hi = (unsigned)(cur[u] >> 32);
#pragma unroll
                    for (int r = 0; r < 32; ++r)
                        acc[r & 3] = popc_and(a_lo[r], a_hi[r], b_lo, b_hi, acc[r & 3]);
                }
#pragma unroll
                for (int u = 0; u < G; ++u) cur[u] = nxt[u];
            }
            __syncthreads();
        }
    }
    const unsigned s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 0xDEADBEEF) out[0] = s;
}

template <int MODE>
static int run_emu(const char* name, int blocks, int iters, unsigned* d,
                   const unsigned long long* src) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kemu<MODE>, dim3(blocks), dim3(256), 0, 0, d, src, iters / 10 + 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kemu<MODE>, dim3(blocks), dim3(256), 0, 0, d, src, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)blocks * 256 * (double)iters * 32 * 32 * 4;  // lane-instructions
    printf("%-10s blocks=%d iters=%d  %.3f ms  %.3e lane-ops/s = %.3e word-pairs/s\n", name, blocks,
           iters, ms, instr / (ms * 1e-3), instr / 4 / (ms * 1e-3));
    return 0;
}

template <int MODE>
static int run(const char* name, int blocks, int iters, unsigned* d) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters / 10, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)blocks * 256 * (double)iters * 64.0;  // lane-instructions
    printf("%-10s blocks=%d iters=%d  %.3f ms  %.3e lane-ops/s  (%.1f%% of 256CU*4SIMD*32lanes*2.4GHz)\n",
           name, blocks, iters, ms, instr / (ms * 1e-3), 100.0 * instr / (ms * 1e-3) / 7.8643e13);
    return 0;
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 4;
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * wps;  // 256 threads = 4 waves = 1 per SIMD
    printf("%s  CUs=%d  clock=%d kHz  waves/SIMD=%d\n", p.gcnArchName, p.multiProcessorCount,
           p.clockRate, wps);
    unsigned* d;
    CHECK(hipMalloc(&d, 64));
    if (run<0>("bcnt", blocks, iters, d)) return 1;
    if (run<1>("and", blocks, iters, d)) return 1;
    if (run<2>("and+bcnt", blocks, iters, d)) return 1;
    unsigned long long* src;
    CHECK(hipMalloc(&src, 32 * 64 * 8));
    CHECK(hipMemset(src, 0x5A, 32 * 64 * 8));
    if (run_emu<3>("emu-regB", blocks, iters / 64 + 1, d, src)) return 1;
    if (run_emu<4>("emu-ldsB", blocks, iters / 64 + 1, d, src)) return 1;
    return 0;
}
