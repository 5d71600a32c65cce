// mfma_fp4_probe.hip — checks that v_mfma_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands computes
// exact AND+popcount dot products of 0/1 data, finds out the operand/result lane maps the K2
// experiment relies on, and measures the instruction's issue rate on gfx950.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// one wave: lane l supplies 4 dwords of A (row l&31, k-half l>>5) and 4 dwords of B
__global__ void one_mfma(const uint32_t* a, const uint32_t* b, float* d) {
    const int l = threadIdx.x;
    v8i va = {}, vb = {};
    for (int i = 0; i < 4; ++i) { va[i] = (int)a[l * 4 + i]; vb[i] = (int)b[l * 4 + i]; }
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc, 4, 4, 0, 0, 0, 0);
    for (int r = 0; r < 16; ++r) d[l * 16 + r] = acc[r];
}

// the 16x16x128 form: lane l supplies 4 dwords of A (row l&15, k-quarter l>>4) and 4 dwords of B
__global__ void one_mfma16(const uint32_t* a, const uint32_t* b, float* d) {
    const int l = threadIdx.x;
    v8i va = {}, vb = {};
    for (int i = 0; i < 4; ++i) { va[i] = (int)a[l * 4 + i]; vb[i] = (int)b[l * 4 + i]; }
    v4f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, acc, 4, 4, 0, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

template <int NACC>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    v8i va = {}, vb = {};
    for (int i = 0; i < 4; ++i) { va[i] = 0x22222222 ^ (threadIdx.x * 0x01010101 & 0x22222222); vb[i] = 0x22220222; }
    v16f acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = v16f{};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n)
            acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc[n], 4, 4, 0, 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    if (s == 12345.f) out[0] = s;
}

int main() {
    // ---- semantics ----
    uint64_t rowsA[32], rowsB[32];
    srand(7);
    auto r64 = []() { uint64_t v = 0; for (int i = 0; i < 4; ++i) v = (v << 16) ^ (uint64_t)(rand() & 0xFFFF); return v; };
    for (int i = 0; i < 32; ++i) { rowsA[i] = r64(); rowsB[i] = r64() & r64(); }
    std::vector<uint32_t> ha(64 * 4), hb(64 * 4);
    for (int l = 0; l < 64; ++l)
        for (int dw = 0; dw < 4; ++dw) {
            uint32_t wa = 0, wb = 0;
            for (int n = 0; n < 8; ++n) {
                const int k = (l >> 5) * 32 + dw * 8 + n;
                if ((rowsA[l & 31] >> k) & 1) wa |= 0x2u << (4 * n);  // E2M1 0b0010 = 1.0
                if ((rowsB[l & 31] >> k) & 1) wb |= 0x2u << (4 * n);
            }
            ha[l * 4 + dw] = wa; hb[l * 4 + dw] = wb;
        }
    uint32_t *da, *db; float* dd;
    CHECK(hipMalloc(&da, ha.size() * 4)); CHECK(hipMalloc(&db, hb.size() * 4)); CHECK(hipMalloc(&dd, 64 * 16 * 4));
    CHECK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dd);
    std::vector<float> hd(64 * 16);
    CHECK(hipMemcpy(hd.data(), dd, hd.size() * 4, hipMemcpyDeviceToHost));
    // C/D map of the guide: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    int bad = 0; double sum = 0, want_sum = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            const int want = __builtin_popcountll(rowsA[row] & rowsB[col]);
            if (hd[l * 16 + r] != (float)want) ++bad;
            sum += hd[l * 16 + r];
        }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) want_sum += __builtin_popcountll(rowsA[i] & rowsB[j]);
    printf("semantics: %d of 1024 entries differ from popcount(A_row & B_col) under the guide's C/D map; tile sum %.0f (want %.0f)\n", bad, sum, want_sum);

    // ---- semantics of the 16x16x128 form (the strip kernel's default shape) ----
    {
        unsigned __int128 ra[16], rb[16];
        auto r128 = [&]() { return ((unsigned __int128)r64() << 64) | r64(); };
        for (int i = 0; i < 16; ++i) { ra[i] = r128(); rb[i] = r128() & r128(); }
        std::vector<uint32_t> ha16(64 * 4), hb16(64 * 4);
        for (int l = 0; l < 64; ++l)
            for (int dw = 0; dw < 4; ++dw) {
                uint32_t wa = 0, wb = 0;
                for (int n = 0; n < 8; ++n) {
                    const int k = (l >> 4) * 32 + dw * 8 + n;
                    if ((uint64_t)(ra[l & 15] >> k) & 1) wa |= 0x2u << (4 * n);
                    if ((uint64_t)(rb[l & 15] >> k) & 1) wb |= 0x2u << (4 * n);
                }
                ha16[l * 4 + dw] = wa; hb16[l * 4 + dw] = wb;
            }
        CHECK(hipMemcpy(da, ha16.data(), ha16.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(db, hb16.data(), hb16.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(one_mfma16, dim3(1), dim3(64), 0, 0, da, db, dd);
        std::vector<float> h16(64 * 4);
        CHECK(hipMemcpy(h16.data(), dd, h16.size() * 4, hipMemcpyDeviceToHost));
        // C/D map: col = lane & 15, row = 4 * (lane >> 4) + reg
        int bad16 = 0;
        auto pc = [](unsigned __int128 v) { return __builtin_popcountll((uint64_t)v) + __builtin_popcountll((uint64_t)(v >> 64)); };
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int col = l & 15, row = 4 * (l >> 4) + r;
                if (h16[l * 4 + r] != (float)pc(ra[row] & rb[col])) ++bad16;
            }
        printf("semantics 16x16x128: %d of 256 entries differ from popcount(A_row & B_col) under col = lane & 15, row = 4 * (lane >> 4) + reg\n", bad16);
    }

    // ---- rate ----
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    float* dout; CHECK(hipMalloc(&dout, 64));
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = p.multiProcessorCount * wps, iters = 20000;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(rate<4>, dim3(blocks), dim3(256), 0, 0, dout, 100);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate<4>, dim3(blocks), dim3(256), 0, 0, dout, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double mfmas = (double)blocks * 4 * iters * 4;
        printf("rate: %d wave/SIMD: %.3f ms, %.3e MFMA/s chip, %.2f ns per MFMA per SIMD, %.3e 64-bit word-pairs/s equivalent\n",
               wps, ms, mfmas / (ms * 1e-3), ms * 1e6 / (mfmas / 1024.0), mfmas * 1024.0 / (ms * 1e-3));
    }
    return 0;
}
