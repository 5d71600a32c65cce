// mfma_war_probe.hip — is an LDS return ordered behind the operand read of an MFMA issued before it?
//
// Every wave loops: read a fragment of ONES from the LDS into registers R, issue N MFMAs with B = R
// (distinct accumulators), then IMMEDIATELY read a fragment of ZEROS from the LDS into the same registers R
// ("+v": same physical registers), wait, repeat. If every MFMA has read R before the zeros land, every
// accumulator ends at iterations x K exactly; an MFMA that was still waiting for the pipe when the LDS
// data arrived multiplies zeros and the sum comes out short. Run for 1..4 waves per SIMD (all waves of the
// chip... see below: the SIMD partners of the reading waves run back-to-back MFMAs only, so the victims'
// MFMAs queue behind theirs), for the unscaled and the scaled
// (VGPR scale operands) FP4 forms of both shapes, and for 1, 2 and 4 MFMAs in front of the overwriting read.
// Prints the number of accumulator elements that came out short per configuration.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int SHAPE, bool SCALED, int NMFMA, int WPS>
__global__ __launch_bounds__(256 * WPS) void probe(unsigned long long* bad, int iters, int scale_bits) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[2][256];  // [0]: ones (0x22222222), [1]: zeros
    for (int i = threadIdx.x; i < 256; i += 256) { lds[0][i] = 0x22222222u; lds[1][i] = 0u; }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&lds[0][0];
    const uint32_t addr_ones = base + lane * 16u, addr_zeros = base + 1024u + lane * 16u;
    const v4i a = v4i{0x22222222, 0x22222222, 0x22222222, 0x22222222};  // 32 x 1.0 per lane
    const int sc = scale_bits;  // 127 = 2^0, in a register: the scaled form
    using acc_t = typename std::conditional<SHAPE == 32, v16f, v4f>::type;
    acc_t acc[NMFMA];
    for (int j = 0; j < NMFMA; ++j) acc[j] = acc_t{};
    v4i r = v4i{};
    // waves 4.. of the workgroup (the SIMD partners of waves 0..3) only keep the matrix pipe full: the
    // victims' MFMAs then queue behind theirs, as they do in a kernel that saturates the pipe
    if (threadIdx.x >= 256) {
        acc_t g[8];
        for (int j = 0; j < 8; ++j) g[j] = acc_t{};
        const int n = iters * (NMFMA + 6) / 8 + 1;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (SHAPE == 32)
                    g[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{a.x, a.y, a.z, a.w, 0, 0, 0, 0}, v8i{a.x, a.y, a.z, a.w, 0, 0, 0, 0}, g[j], 4, 4, 0, sc, 0, sc);
                else
                    g[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(v8i{a.x, a.y, a.z, a.w, 0, 0, 0, 0}, v8i{a.x, a.y, a.z, a.w, 0, 0, 0, 0}, g[j], 4, 4, 0, sc, 0, sc);
            }
        }
        float keep = 0;
        for (int j = 0; j < 8; ++j) keep += g[j][0];
        if (keep == -1.f) bad[1] = 1;
        return;
    }
    for (int it = 0; it < iters; ++it) {
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(addr_ones) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NMFMA; ++j) {
            if constexpr (SHAPE == 32) {
                if constexpr (SCALED)
                    acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{a.x, a.y, a.z, a.w, 0, 0, 0, 0}, v8i{r.x, r.y, r.z, r.w, 0, 0, 0, 0}, acc[j], 4, 4, 0, sc, 0, sc);
                else
                    acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{a.x, a.y, a.z, a.w, 0, 0, 0, 0}, v8i{r.x, r.y, r.z, r.w, 0, 0, 0, 0}, acc[j], 4, 4, 0, 0, 0, 0);
            } else {
                if constexpr (SCALED)
                    acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(v8i{a.x, a.y, a.z, a.w, 0, 0, 0, 0}, v8i{r.x, r.y, r.z, r.w, 0, 0, 0, 0}, acc[j], 4, 4, 0, sc, 0, sc);
                else
                    acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(v8i{a.x, a.y, a.z, a.w, 0, 0, 0, 0}, v8i{r.x, r.y, r.z, r.w, 0, 0, 0, 0}, acc[j], 4, 4, 0, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the overwriting read, right behind the MFMAs that read r
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(addr_zeros) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    // every element of every accumulator: iters x K (K = 64 for 32x32x64, 128 for 16x16x128), below 2^24
    const float want = (float)iters * (SHAPE == 32 ? 64.0f : 128.0f);
    unsigned long long short_elems = 0;
    for (int j = 0; j < NMFMA; ++j)
        for (int e = 0; e < (SHAPE == 32 ? 16 : 4); ++e) short_elems += acc[j][e] != want;
    if (short_elems) atomicAdd(bad, short_elems);
}

template <int SHAPE, bool SCALED, int NMFMA, int WPS>
static int run(unsigned long long* d_bad, int cus) {
    const int iters = 20000;
    CHECK(hipMemset(d_bad, 0, 8));
    hipLaunchKernelGGL((probe<SHAPE, SCALED, NMFMA, WPS>), dim3(cus), dim3(256 * WPS), 0, 0, d_bad, iters, 127);
    CHECK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    CHECK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    const double total = (double)cus * 256 * NMFMA * (SHAPE == 32 ? 16 : 4);  // victims: waves 0..3 of every workgroup
    printf("%2dx%2d %-8s %d MFMA(s) before the overwriting read, %d waves/SIMD: %llu of %.0f accumulator elements short (%.2e)\n",
           SHAPE, SHAPE, SCALED ? "scaled" : "unscaled", NMFMA, WPS, bad, total, bad / total);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    unsigned long long* d_bad; CHECK(hipMalloc(&d_bad, 16));
#define ALL_W(S, SC, N) \
    if (run<S, SC, N, 1>(d_bad, cus)) return 1; if (run<S, SC, N, 2>(d_bad, cus)) return 1; \
    if (run<S, SC, N, 3>(d_bad, cus)) return 1; if (run<S, SC, N, 4>(d_bad, cus)) return 1;
#define ALL_N(S, SC) ALL_W(S, SC, 1) ALL_W(S, SC, 2) ALL_W(S, SC, 4)
    ALL_N(32, false) ALL_N(32, true) ALL_N(16, false) ALL_N(16, true)
    return 0;
}
