// ubench_mall.hip — does the 256 MiB Infinity Cache absorb streaming WRITES (and serve the reads
// that follow)? A store kernel (16 B per lane, fully coalesced) rewrites a buffer of B bytes
// back to back, then a load kernel re-reads it; B from 32 MiB to 1 GiB. If the cache is
// write-back/allocating, buffers well under 256 MiB write faster than HBM streams (~5 TB/s).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void store_k(uint4* p, size_t n, uint32_t v) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        p[i] = uint4{v, v + 1, v + 2, (uint32_t)i};
}
__global__ __launch_bounds__(256) void load_k(const uint4* p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// the expansion's traffic shape: read B/4 bytes, write B bytes
__global__ __launch_bounds__(256) void expand_k(const uint32_t* src, uint4* dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t w = src[i];
        dst[i] = uint4{w & 0x22222222u, (w >> 1) & 0x22222222u, (w >> 2) & 0x22222222u, (w >> 3) & 0x22222222u};
    }
}

int main() {
    uint4* buf; uint32_t* src; uint32_t* out;
    const size_t maxb = 1ull << 30;
    CHECK(hipMalloc(&buf, maxb)); CHECK(hipMalloc(&src, maxb / 4)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(src, 0x5a, maxb / 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (size_t mb : {32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        float ms_w, ms_r, ms_e, ms_wr;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(store_k, dim3(8192), dim3(256), 0, 0, buf, n, 1u);
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(store_k, dim3(8192), dim3(256), 0, 0, buf, n, (uint32_t)r);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_w, e0, e1));
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(load_k, dim3(8192), dim3(256), 0, 0, buf, n, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_r, e0, e1));
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(expand_k, dim3(8192), dim3(256), 0, 0, src, buf, n);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_e, e0, e1));
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 20; ++r) {
            hipLaunchKernelGGL(store_k, dim3(8192), dim3(256), 0, 0, buf, n, (uint32_t)r);
            hipLaunchKernelGGL(load_k, dim3(8192), dim3(256), 0, 0, buf, n, out);
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_wr, e0, e1));
        printf("%5zu MiB: store %6.2f TB/s | load %6.2f TB/s | expand-shaped (1/4 read + write) %6.2f TB/s of written bytes, %6.1f us | store+load pair %6.2f TB/s\n",
               mb, bytes * 20 / (ms_w * 1e-3) / 1e12, bytes * 20 / (ms_r * 1e-3) / 1e12,
               bytes * 20 / (ms_e * 1e-3) / 1e12, ms_e * 1e3 / 20, bytes * 40 / (ms_wr * 1e-3) / 1e12);
    }
    return 0;
}
