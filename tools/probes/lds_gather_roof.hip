// lds_gather_roof.hip — what the LDS of one MI355X CU delivers for the access K4 (probe_lists_kernel) makes: one
// ds_read_u16 per lookup at a data-dependent address of a 16 KiB table, 64 lanes per instruction. Patterns:
//   0  every lane its own bank, same row of banks (the best case)           2  uniformly random positions
//   1  every lane its own bank, random row (conflict-free, scattered)       3  random, + the 2 VALU ops per lookup K4 has
// Build + run (from the repo root, through gpurun):  hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_gather_roof
//   tools/probes/lds_gather_roof.hip && /tmp/lds_gather_roof
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int kMode>
__global__ __launch_bounds__(1024, 8) void gather_kernel(const uint32_t* __restrict__ addr, uint32_t reps,
                                                         unsigned long long* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t table[4096];
    for (uint32_t w = threadIdx.x; w < 4096u; w += 1024u) table[w] = (w * 2654435761u) & 0x007f007fu;
    __syncthreads();
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(table);
    uint32_t a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = addr[(blockIdx.x * 16u + k) * 1024u + threadIdx.x];   // byte offsets, even
    uint32_t pk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pk[k] = a[2 * k] | (a[2 * k + 1] << 16);
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)tb;
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] += base;
    uint32_t count = 0;
    for (uint32_t r = 0; r < reps; ++r) {
        uint32_t c[16];
        if (kMode == 3) {   // K4's unpack: two positions per dword, one VALU operation each
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t lo, hi;
                asm volatile("v_and_b32 %0, 0xffff, %2\n\tv_lshrrev_b32 %1, 16, %2" : "=&v"(lo), "=&v"(hi) : "v"(pk[k]));
                asm volatile("ds_read_u16 %0, %1" : "=v"(c[2 * k]) : "v"(lo));   // (the table is the only LDS object: it starts at 0)
                asm volatile("ds_read_u16 %0, %1" : "=v"(c[2 * k + 1]) : "v"(hi));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("ds_read_u16 %0, %1" : "=v"(c[k]) : "v"(a[k]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 16; ++k) count += c[k];
    }
    if (count == 0xffffffffu) out[0] = count;
}

int main() {
    const uint32_t blocks = 512, reps = 2000;
    std::vector<uint32_t> h((size_t)blocks * 16 * 1024);
    uint32_t* d_addr;
    unsigned long long* d_out;
    hipMalloc(&d_addr, h.size() * 4);
    hipMalloc(&d_out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    uint64_t rng = 88172645463325252ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 11); };
    for (int mode = 0; mode < 4; ++mode) {
        for (size_t i = 0; i < h.size(); ++i) {
            const uint32_t lane = (uint32_t)(i & 63u);
            uint32_t byte;
            if (mode == 0) byte = lane * 4u;                                   // dword = lane: 64 banks in a row
            else if (mode == 1) byte = ((next() & 63u) * 64u + lane) * 4u + (next() & 2u);
            else byte = (next() & 0x1fffu) * 2u;
            h[i] = byte & 0x3ffeu;
        }
        hipMemcpy(d_addr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 3) gather_kernel<3><<<blocks, 1024>>>(d_addr, reps, d_out);
            else gather_kernel<0><<<blocks, 1024>>>(d_addr, reps, d_out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double lookups = (double)blocks * 1024 * 16 * reps;
        printf("{\"pattern\": %d, \"ms\": %.3f, \"lookups_per_s\": %.4g, \"per_clk_per_cu_at_2.4GHz\": %.2f}\n", mode, best,
               lookups / (best * 1e-3), lookups / (best * 1e-3) / 256.0 / 2.4e9);
    }
    return 0;
}
