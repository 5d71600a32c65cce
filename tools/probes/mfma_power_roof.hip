// mfma_power_roof.hip — what an MFMA-ONLY loop of v_mfma_f32_16x16x128_f8f6f4 (FP4, the instruction of K2b) reaches on
// this chip as a function of the OPERAND DATA. Four waves per SIMD, 64 accumulators per wave, A fixed per wave (as in
// K2b), B cycling through eight register quads. Nothing else runs: whatever is lost against 10 PFLOP/s here is clock.
//   data 0: all operands zero            2: one-hot nibbles (0 or 0x2), density 0.39 (the headline matrix's)
//        1: one-hot nibbles, density 0.05     3: one-hot, density 1.0       4: random nibbles (raw bits as FP4 codes)
// With arguments `<data> <shape> <seconds>`: ONE data mode and shape, launched back to back for that many seconds (the
// chip's power management needs seconds, not milliseconds, to settle), with the in-kernel clock witness (shader-clock
// ticks / 100 MHz ticks per workgroup lifetime: MI355X_MICROARCH.md, DVFS give-back (6)) and CLOCK_MONOTONIC stamps for
// tools/clock_power.py's sysfs sampler.
// Build + run through gpurun:  hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_power_roof tools/probes/mfma_power_roof.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <ctime>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ unsigned long long g_clock[4];   // sum of shader-clock ticks, of 100 MHz ticks, workgroups (nothing else reads it)
#define CLOCK_BEGIN() const uint64_t ck_t0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime()
#define CLOCK_END()                                                                              \
    if (threadIdx.x == 0) {                                                                      \
        atomicAdd(&g_clock[0], (unsigned long long)(__builtin_amdgcn_s_memtime() - ck_t0));      \
        atomicAdd(&g_clock[1], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - ck_r0));  \
        atomicAdd(&g_clock[2], 1ull);                                                            \
    }

__global__ __launch_bounds__(256, 4) void mfma_loop(const uint32_t* __restrict__ data, uint32_t iters, float* out) {
    CLOCK_BEGIN();
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    v8i a[4], b[8];
    for (int m = 0; m < 4; ++m) {
        a[m] = v8i{};
        for (int i = 0; i < 4; ++i) a[m][i] = (int)data[(t * 48u + m * 4 + i) % (1u << 22)];
    }
    for (int q = 0; q < 8; ++q) {
        b[q] = v8i{};
        for (int i = 0; i < 4; ++i) b[q][i] = (int)data[(t * 48u + 16 + q * 4 + i) % (1u << 22)];
    }
    v4f acc[4][4];
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 4; ++n) acc[m][n] = v4f{};
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                acc[m][q & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m], b[q], acc[m][q & 3], 4, 4, 0, 0, 0, 0);
    }
    float s = 0;
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 4; ++n)
            for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    if (s == -1.0f) out[0] = s;
    CLOCK_END();
}

// the 32x32x64 form (the output kernels'): 2 x 4 blocks of 32 x 32 per wave = 128 accumulators, two waves per SIMD
__global__ __launch_bounds__(256, 2) void mfma_loop32(const uint32_t* __restrict__ data, uint32_t iters, float* out) {
    CLOCK_BEGIN();
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    v8i a[2], b[8];
    for (int m = 0; m < 2; ++m) {
        a[m] = v8i{};
        for (int i = 0; i < 4; ++i) a[m][i] = (int)data[(t * 48u + m * 4 + i) % (1u << 22)];
    }
    for (int q = 0; q < 8; ++q) {
        b[q] = v8i{};
        for (int i = 0; i < 4; ++i) b[q][i] = (int)data[(t * 48u + 16 + q * 4 + i) % (1u << 22)];
    }
    v16f acc[2][4];
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 4; ++n) acc[m][n] = v16f{};
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                acc[m][q & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], b[q], acc[m][q & 3], 4, 4, 0, 0, 0, 0);
    }
    float s = 0;
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 4; ++n)
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    if (s == -1.0f) out[0] = s;
    CLOCK_END();
}

static double mono_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char** argv) {
    const uint32_t blocks = 256 * 4 * 4, iters = 4000;   // 4 workgroups of 4 waves per CU, four rounds
    const int only_mode = argc > 3 ? atoi(argv[1]) : -1, only_shape = argc > 3 ? atoi(argv[2]) : 0;
    const double seconds = argc > 3 ? atof(argv[3]) : 0.0;
    std::vector<uint32_t> h(1u << 22);
    uint32_t* d;
    float* o;
    (void)hipMalloc(&d, h.size() * 4);
    (void)hipMalloc(&o, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    uint64_t rng = 0x9e3779b97f4a7c15ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 16); };
    // modes 5, 6: density 0.39 with the set code 0.5 / 2.0 (bit 0 / bit 2 of the nibble); 7: the code drawn per dword from
    // 0.5 / 1.0 / 2.0 (the class codes of the output kernels)
    const double dens[8] = {0.0, 0.05, 0.39, 1.0, 0.0, 0.39, 0.39, 0.39};
    for (int mode = 0; mode < 8; ++mode) {
        if (only_mode >= 0 && mode != only_mode) continue;
        for (auto& w : h) {
            uint32_t v = 0;
            const uint32_t code = mode == 5 ? 0x1u : mode == 6 ? 0x4u : mode == 7 ? (0x1u << (next() % 3u)) : 0x2u;
            if (mode == 4) v = next();
            else
                for (int n = 0; n < 8; ++n)
                    if ((next() & 0xffff) < (uint32_t)(dens[mode] * 65536.0)) v |= code << (4 * n);
            w = v;
        }
        (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int shape = 16; shape <= 32; shape += 16) {
            float best = 1e30f;
            const uint32_t nb = shape == 16 ? blocks : blocks / 2;   // two waves per SIMD in the 32 x 32 x 64 form
            if (only_mode >= 0) {
                if (shape != only_shape) continue;
                // sustained: back-to-back launches for `seconds`, a sync every 8 launches; ms per launch of the first and
                // of the last quarter, the in-kernel clock over the whole loop
                char bus[64] = {0};
                (void)hipDeviceGetPCIBusId(bus, sizeof(bus), 0);
                const unsigned long long z[4] = {0, 0, 0, 0};
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clock), z, sizeof(z));
                std::vector<double> t_ms;
                const double t_begin = mono_now();
                while (mono_now() - t_begin < seconds) {
                    (void)hipEventRecord(e0);
                    for (int k = 0; k < 8; ++k) {
                        if (shape == 16) mfma_loop<<<nb, 256>>>(d, iters, o);
                        else mfma_loop32<<<nb, 256>>>(d, iters, o);
                    }
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                    float ms;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    t_ms.push_back(ms / 8.0);
                }
                const double t_end = mono_now();
                unsigned long long ck[4];
                (void)hipMemcpyFromSymbol(ck, HIP_SYMBOL(g_clock), sizeof(ck));
                auto mean = [&](size_t a, size_t b) { double s = 0; for (size_t i = a; i < b; ++i) s += t_ms[i]; return s / (double)(b - a); };
                const size_t q = t_ms.size() / 4 ? t_ms.size() / 4 : 1;
                const double flop1 = (double)nb * 4 * iters * (shape == 16 ? 32 * 65536.0 : 16 * 131072.0);
                const double ms_last = mean(t_ms.size() - q, t_ms.size());
                printf("{\"kernel\": \"mfma_only_loop\", \"data\": %d, \"shape\": %d, \"pci_bus\": \"%s\", \"mono_begin\": %.6f, \"mono_end\": %.6f, "
                       "\"launches\": %zu, \"ms_first_quarter\": %.4f, \"ms_last_quarter\": %.4f, \"frac_of_10_pflops_last_quarter\": %.4f, "
                       "\"in_kernel_clock_mhz\": %.1f, \"workgroups_stamped\": %llu}\n",
                       mode, shape, bus, t_begin, t_end, t_ms.size() * 8, mean(0, q), ms_last, flop1 / (ms_last * 1e-3) / 1e16,
                       ck[1] ? 100.0 * (double)ck[0] / (double)ck[1] : 0.0, ck[2]);
                continue;
            }
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0);
                if (shape == 16) mfma_loop<<<nb, 256>>>(d, iters, o);
                else mfma_loop32<<<nb, 256>>>(d, iters, o);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;   // (the first run ramps the clock)
            }
            // per iteration and wave: 32 MFMAs of 2 * 16 * 16 * 128 flop, or 16 of 2 * 32 * 32 * 64
            const double flop = (double)nb * 4 * iters * (shape == 16 ? 32 * 65536.0 : 16 * 131072.0);
            printf("{\"data\": %d, \"shape\": %d, \"ms\": %.3f, \"pflops\": %.3f, \"frac_of_10_pflops\": %.4f}\n", mode, shape, best,
                   flop / (best * 1e-3) / 1e15, flop / (best * 1e-3) / 1e16);
        }
    }
    return 0;
}
