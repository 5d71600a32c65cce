// ubench_lone.hip — what one wave ALONE on its SIMD reaches of the matrix pipe when the operand inflation of
// bitstream_kernel (K2q) sits between its MFMAs: clocks per "stage" of 32 v_mfma_scale_f32_32x32x64_f8f6f4
// (1024 clocks of matrix pipe) as a function of
//   NV  : VALU instructions between the two MFMAs of a pair (the kernel: 4, or 8 for class 3),
//   DEP : whether the second-next MFMA reads what those instructions wrote (the kernel: yes),
//   SC  : whether a v_mov of the block scale + s_nop 1 precedes each pair (what hipcc emits for a scale in an SGPR),
//   WPS : waves per SIMD (1..3).
// No memory traffic at all. Usage: ubench_lone [iters=2000]
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_lone tools/ubench_lone.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                       \
    do {                                                               \
        hipError_t e = (x);                                            \
        if (e != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));     \
            return 1;                                                  \
        }                                                              \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NV, bool DEP, bool SC, int WPS>
__global__ __launch_bounds__(256, WPS) void lone_kernel(float* out, unsigned long long* clk, int iters, int scale_in) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    v4i a[2][4][2];
    for (int g = 0; g < 2; ++g)
        for (int c = 0; c < 4; ++c)
            for (int m = 0; m < 2; ++m) {
                const int s = (int)(gid * 2654435761u) + g * 8 + c * 2 + m;
                a[g][c][m] = v4i{s & 0x11111111, (s >> 1) & 0x11111111, (s >> 2) & 0x11111111, (s >> 3) & 0x11111111};
            }
    v4i w = v4i{(int)gid, (int)(gid * 3u), (int)(gid * 5u), (int)(gid * 7u)};
    v4i e = v4i{w.x & 0x11111111, w.y & 0x11111111, w.z & 0x11111111, w.w & 0x11111111};
    v4i dummy = v4i{};
    v16f acc[2][2];
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n) acc[m][n] = v16f{};
    int sc = scale_in;  // a uniform value the compiler cannot fold: lives in an SGPR
    unsigned long long c0 = 0;
    if (threadIdx.x == 0) c0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {  // 16 pairs = 32 MFMAs = one stage
            const int n = (p >> 2) & 1, g = p >> 3, c = p & 3;
            const int sv = SC ? sc + c : 127;
            acc[0][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                v8i{a[g][c][0].x, a[g][c][0].y, a[g][c][0].z, a[g][c][0].w, 0, 0, 0, 0},
                v8i{e.x, e.y, e.z, e.w, 0, 0, 0, 0}, acc[0][n], 4, 4, 0, 127, 0, sv);
            __builtin_amdgcn_sched_barrier(0);
            v4i en = e;
            if constexpr (NV >= 4) {
                asm volatile("v_and_b32 %0, 0x22222222, %4\n\tv_and_b32 %1, 0x22222222, %5\n\t"
                             "v_and_b32 %2, 0x22222222, %6\n\tv_and_b32 %3, 0x22222222, %7"
                             : "=&v"(en.x), "=&v"(en.y), "=&v"(en.z), "=&v"(en.w)
                             : "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w));
            } else if constexpr (NV >= 1) {
                asm volatile("v_and_b32 %0, 0x22222222, %1" : "=&v"(en.x) : "v"(w.x));
                if constexpr (NV >= 2) asm volatile("v_and_b32 %0, 0x22222222, %1" : "=&v"(en.y) : "v"(w.y));
                if constexpr (NV >= 3) asm volatile("v_and_b32 %0, 0x22222222, %1" : "=&v"(en.z) : "v"(w.z));
            }
            if constexpr (NV >= 8) {
                asm volatile("v_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\t"
                             "v_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3"
                             : "+v"(en.x), "+v"(en.y), "+v"(en.z), "+v"(en.w));
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[1][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                v8i{a[g][c][1].x, a[g][c][1].y, a[g][c][1].z, a[g][c][1].w, 0, 0, 0, 0},
                v8i{e.x, e.y, e.z, e.w, 0, 0, 0, 0}, acc[1][n], 4, 4, 0, 127, 0, sv);
            if constexpr (DEP) e = en;
            else dummy = en;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (threadIdx.x == 0) clk[blockIdx.x] = __builtin_readcyclecounter() - c0;
    float s = 0;
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n)
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    if (s == 12345.678f || dummy.x == 0x7fffffff) out[gid] = s;
}

template <int NV, bool DEP, bool SC, int WPS>
static int run(int iters, float* d_out, unsigned long long* d_clk, int n_cus) {
    const int groups = n_cus * WPS;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int warm = 0; warm < 3; ++warm)
        hipLaunchKernelGGL((lone_kernel<NV, DEP, SC, WPS>), dim3(groups), dim3(256), 0, 0, d_out, d_clk, iters, 126);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((lone_kernel<NV, DEP, SC, WPS>), dim3(groups), dim3(256), 0, 0, d_out, d_clk, iters, 126);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> clk(groups);
    CHECK(hipMemcpy(clk.data(), d_clk, groups * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto c : clk) mean += (double)c;
    mean /= groups;
    const double per_stage = mean / iters;
    const double ghz = mean / (ms * 1e-3) / 1e9;  // counter ticks per second of kernel time (~ the shader clock)
    // pipe clocks per SIMD per stage-iteration: WPS waves x 1024
    printf("NV %d dep %d scale-mov %d  %d wave(s)/SIMD: %7.1f ticks per stage per wave, %6.1f per SIMD-stage  "
           "(pipe 1024; counter %.2f GHz over %.3f ms)  pipe use %.2f\n",
           NV, (int)DEP, (int)SC, WPS, per_stage, per_stage / WPS, ghz, ms, 1024.0 * WPS / per_stage);
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    float* d_out;
    unsigned long long* d_clk;
    CHECK(hipMalloc(&d_out, (size_t)n_cus * 3 * 256 * sizeof(float)));
    CHECK(hipMalloc(&d_clk, (size_t)n_cus * 3 * sizeof(unsigned long long)));
    if (run<0, false, false, 1>(iters, d_out, d_clk, n_cus)) return 1;
    if (run<1, true, false, 1>(iters, d_out, d_clk, n_cus)) return 1;
    if (run<2, true, false, 1>(iters, d_out, d_clk, n_cus)) return 1;
    if (run<3, true, false, 1>(iters, d_out, d_clk, n_cus)) return 1;
    if (run<4, false, false, 1>(iters, d_out, d_clk, n_cus)) return 1;
    if (run<4, true, false, 1>(iters, d_out, d_clk, n_cus)) return 1;
    if (run<8, true, false, 1>(iters, d_out, d_clk, n_cus)) return 1;
    if (run<4, true, true, 1>(iters, d_out, d_clk, n_cus)) return 1;
    if (run<0, false, true, 1>(iters, d_out, d_clk, n_cus)) return 1;
    if (run<4, true, false, 2>(iters, d_out, d_clk, n_cus)) return 1;
    if (run<4, true, true, 2>(iters, d_out, d_clk, n_cus)) return 1;
    if (run<4, true, false, 3>(iters, d_out, d_clk, n_cus)) return 1;
    if (run<4, true, true, 3>(iters, d_out, d_clk, n_cus)) return 1;
    if (run<8, true, true, 3>(iters, d_out, d_clk, n_cus)) return 1;
    return 0;
}
