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
    return 0;
}
