// ubench_feed.hip — what does it cost a wave to feed the matrix pipe? A loop of 16 FP4 MFMAs
// (the strip kernel's stage) plus, per iteration, one of:
//   0 nothing | 1 two global_load_lds (16 B/lane, LDS-DMA) | 2 two global_load_dwordx4 into
//   registers | 3 mode 2 + two ds_write_b128 of the PREVIOUS iteration's registers |
//   4 eight ds_read_b128 | 5 mode 1 + mode 4 | 6 mode 3 + mode 4 |
//   7 two buffer_load_dwordx4 ... lds (SRD + 32-bit offsets) | 8 mode 7 + mode 4
// at W waves per SIMD (W workgroups of 256 threads per CU). Reports cycles-equivalents per
// iteration from the wall time (the clock under load is not known exactly; compare modes).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void feed(const uint8_t* __restrict__ src, uint64_t row_bytes,
                                                 float* out, int iters) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[4][8192];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    v4i a[2] = {v4i{0x22222222, 0x02020202, 0x20202020, 0x22002200}, v4i{0x22220000, 0x2222, 0x2, 0x20}};
    v4i b[2] = {v4i{0x22222222, 0x22222222, 0x2020202, 0x2200220}, v4i{0x2222000, 0x222, 0x22, 0x20}};
    v16f acc[4];
    for (int n = 0; n < 4; ++n) acc[n] = v16f{};
    // per-lane source address like the strip kernel: 8 rows x 128 B per instruction
    const uint32_t r0 = (wave * 64u + lane) >> 3;
    const uint8_t* base = src + (uint64_t)(blockIdx.x % 64u) * 64u * row_bytes;
    const uint32_t goff = r0 * (uint32_t)row_bytes + (lane & 7u) * 16u;
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&lds[0][0];
    // conflict-free fragment reads: 16-byte slot XOR-swizzled by the row, as in the strip kernel
    const uint32_t swz = (lane >> 1) & 7u;
    uint32_t laddr[4];
    for (int q = 0; q < 4; ++q)
        laddr[q] = lbase + (lane & 31u) * 128u + ((((uint32_t)q * 2u + (lane >> 5)) ^ swz) * 16u);
    v4i st0 = {}, st1 = {}, rd[8] = {};
    for (int it = 0; it < iters; ++it) {
        const uint8_t* g = base + (uint64_t)(it & 63) * 128u;  // walk along k, stays in L2
        if constexpr (MODE == 1 || MODE == 5) {
            uint8_t* dst = lds[it & 3] + wave * 1024u;
            __builtin_amdgcn_global_load_lds((gptr_t)(g + goff), (lptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(g + 32u * row_bytes + goff), (lptr_t)(dst + 4096u), 16, 0, 0);
        }
        if constexpr (MODE == 7 || MODE == 8) {
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, 0x7fffffff, 0x00020000);
            uint8_t* dst = lds[it & 3] + wave * 1024u;
            const uint32_t so = (uint32_t)(it & 63) * 128u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)dst, 16, (int)goff, (int)so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + 4096u), 16,
                                                 (int)(goff + 32u * (uint32_t)row_bytes), (int)so, 0, 0);
        }
        if constexpr (MODE == 3 || MODE == 6) {  // registers loaded one iteration ago -> LDS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t w = lbase + (it & 3) * 8192u + wave * 1024u + lane * 16u;
            asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:4096" ::"v"(w), "v"(st0), "v"(st1) : "memory");
        }
        if constexpr (MODE == 2 || MODE == 3 || MODE == 6) {
            asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off"
                         : "=&v"(st0), "=&v"(st1)
                         : "v"(g + goff), "v"(g + 32u * row_bytes + goff)
                         : "memory");
        }
        if constexpr (MODE >= 4 && MODE != 7) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096"
                             : "=&v"(rd[2 * q]), "=&v"(rd[2 * q + 1])
                             : "v"(laddr[q] + (it & 3) * 8192u));
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                    v8i{a[n & 1].x, a[n & 1].y, a[n & 1].z, a[n & 1].w, 0, 0, 0, 0},
                    v8i{b[n >> 1].x, b[n >> 1].y, b[n >> 1].z, b[n >> 1].w, 0, 0, 0, 0}, acc[n], 4, 4, 0, 0, 0, 0);
        if constexpr (MODE >= 4 && MODE != 7) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("" ::"v"(rd[q]));
        }
        if constexpr (MODE == 1 || MODE == 5 || MODE == 7 || MODE == 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float s = 0;
    for (int n = 0; n < 4; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    s += (float)(st0.x + st1.x);
    if (s == 12345.f) out[0] = s;
}

// RES: workgroups per CU actually launched (RES < WPS: the WPS-wave code at lower residency)
template <int MODE, int WPS, int RES = WPS>
static int run(const uint8_t* src, uint64_t row_bytes, float* out, int cus, const char* name) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((feed<MODE, WPS>), dim3(cus * RES), dim3(256), 0, 0, src, row_bytes, out, 200);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((feed<MODE, WPS>), dim3(cus * RES), dim3(256), 0, 0, src, row_bytes, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: WPS waves x iters iterations x 16 MFMAs
    const double us_per_wave_iter = ms * 1e3 / iters;
    const double mfma_ns = ms * 1e6 / ((double)iters * RES * 16);
    printf("%-44s code for %d, resident %d waves/SIMD: %.3f us per wave-iteration, %.2f ns per MFMA per SIMD\n",
           name, WPS, RES, us_per_wave_iter, mfma_ns);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const uint64_t row_bytes = 32768;  // the headline shape's nibble rows
    uint8_t* src; float* out;
    CHECK(hipMalloc(&src, 4096 * row_bytes)); CHECK(hipMemset(src, 0x22, 4096 * row_bytes));
    CHECK(hipMalloc(&out, 64));
#define ROW(M, NAME) \
    if (run<M, 4>(src, row_bytes, out, cus, NAME)) return 1; \
    if (run<M, 3>(src, row_bytes, out, cus, NAME)) return 1; \
    if (run<M, 2>(src, row_bytes, out, cus, NAME)) return 1;
    ROW(0, "16 MFMA")
    ROW(1, "16 MFMA + 2 global_load_lds")
    ROW(2, "16 MFMA + 2 global_load_dwordx4")
    ROW(3, "16 MFMA + 2 global_load_dwordx4 + 2 ds_write")
    ROW(4, "16 MFMA + 8 ds_read_b128")
    ROW(5, "16 MFMA + 2 global_load_lds + 8 ds_read")
    ROW(6, "16 MFMA + 2 gld + 2 ds_write + 8 ds_read")
    ROW(7, "16 MFMA + 2 buffer_load_lds")
    ROW(8, "16 MFMA + 2 buffer_load_lds + 8 ds_read")
    // the 4-wave code at lower residency: is the LDS-DMA cost a matter of code or of concurrency?
    if (run<1, 4, 2>(src, row_bytes, out, cus, "16 MFMA + 2 global_load_lds")) return 1;
    if (run<1, 4, 1>(src, row_bytes, out, cus, "16 MFMA + 2 global_load_lds")) return 1;
    if (run<5, 4, 2>(src, row_bytes, out, cus, "16 MFMA + 2 global_load_lds + 8 ds_read")) return 1;
    if (run<1, 2, 1>(src, row_bytes, out, cus, "16 MFMA + 2 global_load_lds")) return 1;
    return 0;
}
