// ubench_shape.hip — which FP4 MFMA shape sustains more FLOP/s on this chip under load?
//   v_mfma_scale_f32_32x32x64_f8f6f4  (32 cycles, 16 per wave-iteration)   vs
//   v_mfma_scale_f32_16x16x128_f8f6f4 (16 cycles, 32 per wave-iteration)
// Same wave tile either way (64 A rows x 64 B rows x 256 bits of k per iteration: the strip
// kernel's stage), same operand registers (32 A + 32 B), same accumulators (64). Operands are
// random 0/1 bits as E2M1 1.0 (0b0010) at the headline density (39 %), B rotating every
// iteration. MI355X_MICROARCH.md "DVFS give-back" (7) reports the 16x16 bf16 shape holding a
// higher clock than the 32x32 one at equal cycles; this asks the same of FP4.
//   FEED 0: registers only | 1: + the strip kernel's stage traffic (2 buffer_load ... lds per
//   wave + 8 ds_read_b128 per iteration, fragments consumed)
// Prints wall time per wave-iteration, the in-kernel clock (s_memtime / s_memrealtime) and
// the FP4 PFLOP/s of the whole chip.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
using lptr_t = __attribute__((address_space(3))) void*;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// 8 nibbles, each 0b0010 with probability ~0.39
__device__ __forceinline__ uint32_t rand_nibbles(uint32_t seed) {
    uint32_t out = 0;
    for (int i = 0; i < 8; ++i) {
        const uint32_t r = mix(seed * 8u + i) & 0xffffu;
        if (r < 25770u) out |= 2u << (4 * i);
    }
    return out;
}

// 32 bits, each set with probability ~0.39 (the bit-operand feed multiplies these in all four classes;
// a first version fed it the NIBBLE image, i.e. bits at 10 % density in one class only, and the nearly
// empty operands ran 2.29 GHz — 0.2 GHz above what real rows sustain)
__device__ __forceinline__ uint32_t rand_bits(uint32_t seed) {
    uint32_t out = 0;
    for (int i = 0; i < 32; ++i)
        if ((mix(seed * 32u + i) & 0xffffu) < 25770u) out |= 1u << i;
    return out;
}
__global__ void fill_random_bits(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = rand_bits((uint32_t)i * 2654435761u + 777u);
}

__global__ void fill_random(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = rand_nibbles((uint32_t)i * 2654435761u + 12345u);
}

template <int SHAPE, int FEED, int WPS>
__global__ __launch_bounds__(256, WPS) void shape_kernel(const uint8_t* __restrict__ src, uint64_t row_bytes,
                                                         float* out, unsigned long long* clk, int iters) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[4][8192];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    v4i a[8], b[4];
    for (int i = 0; i < 8; ++i)
        a[i] = v4i{(int)rand_nibbles(gid * 64u + i * 4u + 0), (int)rand_nibbles(gid * 64u + i * 4u + 1),
                   (int)rand_nibbles(gid * 64u + i * 4u + 2), (int)rand_nibbles(gid * 64u + i * 4u + 3)};
    for (int i = 0; i < 4; ++i)
        b[i] = v4i{(int)rand_nibbles(gid * 64u + 32u + i * 4u + 0), (int)rand_nibbles(gid * 64u + 32u + i * 4u + 1),
                   (int)rand_nibbles(gid * 64u + 32u + i * 4u + 2), (int)rand_nibbles(gid * 64u + 32u + i * 4u + 3)};
    constexpr int NACC = SHAPE == 32 ? 4 : 16;
    using acc_t = typename std::conditional<SHAPE == 32, v16f, v4f>::type;
    acc_t acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = acc_t{};
    const uint32_t r0 = (wave * 64u + lane) >> 3;
    const uint8_t* base = src + (uint64_t)(blockIdx.x % 64u) * 64u * row_bytes;
    const uint32_t goff = r0 * (uint32_t)row_bytes + (lane & 7u) * 16u;
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&lds[0][0];
    const uint32_t swz = (lane >> 1) & 7u;
    // fragment addresses. 32x32: k-step q (64 bits = 32 B), lane -> row lane&31 (+32 at offset 4096),
    // 16-byte half lane>>5. 16x16: k-step q (128 bits = 64 B), lane -> row lane&15 (+16 per B block
    // = +2048 B), 16-byte quarter lane>>4. Same XOR swizzle (conflict-free for both, see DESIGN).
    uint32_t laddr[4];
    if constexpr (SHAPE == 32) {
        for (int q = 0; q < 4; ++q)
            laddr[q] = lbase + (lane & 31u) * 128u + ((((uint32_t)q * 2u + (lane >> 5)) ^ swz) * 16u);
    } else {
        const uint32_t swz16 = ((lane & 15u) >> 1) & 7u;
        for (int q = 0; q < 2; ++q)
            laddr[q] = lbase + (lane & 15u) * 128u + ((((uint32_t)q * 4u + (lane >> 4)) ^ swz16) * 16u);
        laddr[2] = laddr[3] = 0;
    }
    auto mfma32 = [&](int m, int n, const v4i& av, const v4i& bv) {
        if constexpr (SHAPE == 32)
            acc[m * 2 + n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                v8i{av.x, av.y, av.z, av.w, 0, 0, 0, 0}, v8i{bv.x, bv.y, bv.z, bv.w, 0, 0, 0, 0},
                acc[m * 2 + n], 4, 4, 0, 0, 0, 0);
    };
    auto mfma16 = [&](int m, int n, const v4i& av, const v4i& bv) {
        if constexpr (SHAPE == 16)
            acc[m * 4 + n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                v8i{av.x, av.y, av.z, av.w, 0, 0, 0, 0}, v8i{bv.x, bv.y, bv.z, bv.w, 0, 0, 0, 0},
                acc[m * 4 + n], 4, 4, 0, 0, 0, 0);
    };
    if constexpr (FEED == 2) {
        for (uint32_t i = threadIdx.x; i < 4 * 8192 / 4; i += 256)
            reinterpret_cast<uint32_t*>(&lds[0][0])[i] = rand_nibbles(gid * 977u + i);
        __syncthreads();
    }
    unsigned long long c0 = 0, t0 = 0;
    if (iters > 1000 && threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); t0 = __builtin_amdgcn_s_memrealtime(); }
    if constexpr (FEED == 1 || FEED == 3) {  // two stages in flight before the loop
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, 0x7fffffff, 0x00020000);
        for (int it = -2; it < 0; ++it) {
            uint8_t* dst = lds[it & 3] + wave * 1024u;
            const uint32_t so = (uint32_t)(it & 63) * 128u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)dst, 16, (int)goff, (int)so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + 4096u), 16,
                                                     (int)(goff + 32u * (uint32_t)row_bytes), (int)so, 0, 0);
        }
    }
    for (int it = 0; it < iters; ++it) {
        if constexpr (FEED == 3) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, 0x7fffffff, 0x00020000);
            uint8_t* dst = lds[it & 3] + wave * 1024u;
            const uint32_t so = (uint32_t)(it & 63) * 128u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)dst, 16, (int)goff, (int)so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + 4096u), 16,
                                                     (int)(goff + 32u * (uint32_t)row_bytes), (int)so, 0, 0);
        }
        if constexpr (FEED == 1 || FEED == 2) {
            // stage it-2 has landed (2 younger stages = 4 instructions may stay in flight); barrier as
            // in the strip kernel, then refill the ring
            if constexpr (FEED == 1) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, 0x7fffffff, 0x00020000);
            uint8_t* dst = lds[it & 3] + wave * 1024u;
            const uint32_t so = (uint32_t)(it & 63) * 128u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)dst, 16, (int)goff, (int)so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + 4096u), 16,
                                                     (int)(goff + 32u * (uint32_t)row_bytes), (int)so, 0, 0);
            }
            const uint32_t sb = ((it + 2) & 3) * 8192u;  // the stage that landed
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SHAPE == 32) {
                // per k-step: 2 reads (next k-step) ahead of 4 MFMAs
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096" : "=&v"(b[0]), "=&v"(b[1]) : "v"(laddr[0] + sb));
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int cur = (kk & 1) * 2, nxt = 2 - cur;
                    if (kk < 3)
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096" : "=&v"(b[nxt]), "=&v"(b[nxt + 1]) : "v"(laddr[kk + 1] + sb));
                    if (kk < 3) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) mfma32(m, n, a[kk * 2 + m], b[cur + n]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // per (k-step, B block): 1 read (next block) ahead of 4 MFMAs
                asm volatile("ds_read_b128 %0, %1" : "=&v"(b[0]) : "v"(laddr[0] + sb));
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int kk = s >> 2, n = s & 3, cur = s & 1, nxt = 1 - cur;
                    if (s < 7) {
                        const int s1 = s + 1;
                        asm volatile("ds_read_b128 %0, %1" : "=&v"(b[nxt]) : "v"(laddr[s1 >> 2] + sb + (uint32_t)(s1 & 3) * 2048u));
                        asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 4; ++m) mfma16(m, n, a[kk * 4 + m], b[cur]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if constexpr (FEED == 0 || FEED == 3) {
            if constexpr (SHAPE == 32) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) mfma32(m, n, a[kk * 2 + m], b[(kk & 1) * 2 + n]);
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int m = 0; m < 4; ++m) mfma16(m, n, a[kk * 4 + m], b[n]);
            }
            // rotate B so that the operand lines toggle as they do when B streams through
            const v4i t = b[0];
            b[0] = b[1]; b[1] = b[2]; b[2] = b[3]; b[3] = t;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (iters > 1000 && threadIdx.x == 0) {
        clk[blockIdx.x * 2 + 0] = __builtin_amdgcn_s_memtime() - c0;
        clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - t0;
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < (SHAPE == 32 ? 16 : 4); ++r) s += acc[n][r];
    if (s == 12345.f) out[0] = s;
}

// FEED 4: the strip stage fed with BITS (the output kernels' operand trick): A stationary and inflated
// (64 rows x 512 bits = 16 operands = 64 VGPRs), B stage = 64 rows x 512 bits = 4 KiB of bits (one LDS-DMA
// piece per wave), every wave inflates its own copy of the B operands: per (B block, class) one
// v_and_b32 per operand dword (class 3: shift + and), then the MFMAs. Per wave-iteration
// 64 x 64 x 512 bit-MACs = 2^22 FLOP (twice the other feeds').
template <int C>
__device__ __forceinline__ v4i inflate(v4i w) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    if constexpr (C == 3) return (v4i)(((v4u)w >> 1u) & 0x44444444u);
    else return w & (int)(0x11111111u << C);
}
template <int C>
__device__ __forceinline__ int cscale() { return C == 0 ? 128 : C == 1 ? 127 : 126; }

template <int SHAPE, int WPS>
__global__ __launch_bounds__(256, WPS) void bits_kernel(const uint8_t* __restrict__ src, uint64_t row_bytes,
                                                        float* out, unsigned long long* clk, int iters) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[4][4096];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    v4i a[16];  // [row block or (row block, k-group)][class]
    for (int i = 0; i < 16; ++i) {
        const int c = i & 3;
        const v4i w = v4i{(int)rand_bits(gid * 64u + i * 4u + 0), (int)rand_bits(gid * 64u + i * 4u + 1),
                          (int)rand_bits(gid * 64u + i * 4u + 2), (int)rand_bits(gid * 64u + i * 4u + 3)};
        a[i] = c == 0 ? inflate<0>(w) : c == 1 ? inflate<1>(w) : c == 2 ? inflate<2>(w) : inflate<3>(w);
    }
    constexpr int NACC = SHAPE == 32 ? 4 : 16;
    using acc_t = typename std::conditional<SHAPE == 32, v16f, v4f>::type;
    acc_t acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = acc_t{};
    // DMA: piece = 16 rows x 64 B; lane L: row L / 4, 16-byte slot (L % 4) ^ ((L / 16) % 4)
    const uint8_t* base = src + (uint64_t)(blockIdx.x % 64u) * 64u * row_bytes;
    const uint32_t goff = (wave * 16u + (lane >> 2)) * (uint32_t)row_bytes + (((lane & 3u) ^ ((lane >> 4) & 3u)) * 16u);
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&lds[0][0];
    // fragments: 16x16: row lane & 15 (+16 per block), slot lane >> 4; 32x32: row lane & 31, slot 2 g + (lane >> 5)
    uint32_t laddr[2];
    if constexpr (SHAPE == 16) {
        laddr[0] = lbase + (lane & 15u) * 64u + (((lane >> 4) ^ (((lane & 15u) >> 2) & 3u)) * 16u);
        laddr[1] = 0;
    } else {
        const uint32_t sl = (lane >> 5) ^ (((lane & 31u) >> 2) & 3u);
        laddr[0] = lbase + (lane & 31u) * 64u + sl * 16u;
        laddr[1] = lbase + (lane & 31u) * 64u + (sl ^ 2u) * 16u;
    }
    unsigned long long c0 = 0, t0 = 0;
    if (iters > 1000 && threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); t0 = __builtin_amdgcn_s_memrealtime(); }
    {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, 0x7fffffff, 0x00020000);
        for (int it = -2; it < 0; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds[it & 3] + wave * 1024u), 16, (int)goff,
                                                     (int)((uint32_t)(it & 63) * 64u), 0, 0);
    }
    v4i w0, w1, e0, e1;
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds[it & 3] + wave * 1024u), 16, (int)goff,
                                                     (int)((uint32_t)(it & 63) * 64u), 0, 0);
        }
        const uint32_t sb = ((it + 2) & 3) * 4096u;  // the stage that landed
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SHAPE == 16) {
            // 4 B blocks x 4 classes; per (block, class): 4 MFMAs (the 4 A row blocks)
            asm volatile("ds_read_b128 %0, %1" : "=&v"(w0) : "v"(laddr[0] + sb));
            asm volatile("ds_read_b128 %0, %1 offset:1024" : "=&v"(w1) : "v"(laddr[0] + sb));
            asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            e0 = inflate<0>(w0);
#define UB_STEP(n, C, wcur, ecur, enxt, NEXT)                                                        \
            _Pragma("unroll") for (int m = 0; m < 4; ++m)                                            \
                acc[m * 4 + n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                   \
                    v8i{a[m * 4 + C].x, a[m * 4 + C].y, a[m * 4 + C].z, a[m * 4 + C].w, 0, 0, 0, 0}, \
                    v8i{ecur.x, ecur.y, ecur.z, ecur.w, 0, 0, 0, 0}, acc[m * 4 + n], 4, 4, 0,        \
                    cscale<C>(), 0, cscale<C>());                                                    \
            enxt = NEXT;                                                                             \
            __builtin_amdgcn_sched_barrier(0)
            UB_STEP(0, 0, w0, e0, e1, inflate<1>(w0));
            UB_STEP(0, 1, w0, e1, e0, inflate<2>(w0));
            UB_STEP(0, 2, w0, e0, e1, inflate<3>(w0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            UB_STEP(0, 3, w0, e1, e0, inflate<0>(w1));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=&v"(w0) : "v"(laddr[0] + sb));
            UB_STEP(1, 0, w1, e0, e1, inflate<1>(w1));
            UB_STEP(1, 1, w1, e1, e0, inflate<2>(w1));
            UB_STEP(1, 2, w1, e0, e1, inflate<3>(w1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            UB_STEP(1, 3, w1, e1, e0, inflate<0>(w0));
            asm volatile("ds_read_b128 %0, %1 offset:3072" : "=&v"(w1) : "v"(laddr[0] + sb));
            UB_STEP(2, 0, w0, e0, e1, inflate<1>(w0));
            UB_STEP(2, 1, w0, e1, e0, inflate<2>(w0));
            UB_STEP(2, 2, w0, e0, e1, inflate<3>(w0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            UB_STEP(2, 3, w0, e1, e0, inflate<0>(w1));
            UB_STEP(3, 0, w1, e0, e1, inflate<1>(w1));
            UB_STEP(3, 1, w1, e1, e0, inflate<2>(w1));
            UB_STEP(3, 2, w1, e0, e1, inflate<3>(w1));
            UB_STEP(3, 3, w1, e1, e0, e1);
#undef UB_STEP
        } else {
            // 2 B blocks x 2 k-groups x 4 classes; per (block, k-group, class): 2 MFMAs (the 2 A row blocks)
            asm volatile("ds_read_b128 %0, %1" : "=&v"(w0) : "v"(laddr[0] + sb));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=&v"(w1) : "v"(laddr[0] + sb));
            asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            e0 = inflate<0>(w0);
#define UB_STEP(n, g, C, ecur, enxt, NEXT)                                                           \
            _Pragma("unroll") for (int m = 0; m < 2; ++m)                                            \
                acc[m * 2 + n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                    \
                    v8i{a[(m * 2 + g) * 4 + C].x, a[(m * 2 + g) * 4 + C].y, a[(m * 2 + g) * 4 + C].z, \
                        a[(m * 2 + g) * 4 + C].w, 0, 0, 0, 0},                                       \
                    v8i{ecur.x, ecur.y, ecur.z, ecur.w, 0, 0, 0, 0}, acc[m * 2 + n], 4, 4, 0,        \
                    cscale<C>(), 0, cscale<C>());                                                    \
            enxt = NEXT;                                                                             \
            __builtin_amdgcn_sched_barrier(0)
            UB_STEP(0, 0, 0, e0, e1, inflate<1>(w0));
            UB_STEP(0, 0, 1, e1, e0, inflate<2>(w0));
            UB_STEP(0, 0, 2, e0, e1, inflate<3>(w0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            UB_STEP(0, 0, 3, e1, e0, inflate<0>(w1));
            asm volatile("ds_read_b128 %0, %1" : "=&v"(w0) : "v"(laddr[1] + sb));
            UB_STEP(1, 0, 0, e0, e1, inflate<1>(w1));
            UB_STEP(1, 0, 1, e1, e0, inflate<2>(w1));
            UB_STEP(1, 0, 2, e0, e1, inflate<3>(w1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            UB_STEP(1, 0, 3, e1, e0, inflate<0>(w0));
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=&v"(w1) : "v"(laddr[1] + sb));
            UB_STEP(0, 1, 0, e0, e1, inflate<1>(w0));
            UB_STEP(0, 1, 1, e1, e0, inflate<2>(w0));
            UB_STEP(0, 1, 2, e0, e1, inflate<3>(w0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            UB_STEP(0, 1, 3, e1, e0, inflate<0>(w1));
            UB_STEP(1, 1, 0, e0, e1, inflate<1>(w1));
            UB_STEP(1, 1, 1, e1, e0, inflate<2>(w1));
            UB_STEP(1, 1, 2, e0, e1, inflate<3>(w1));
            UB_STEP(1, 1, 3, e1, e0, e1);
#undef UB_STEP
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (iters > 1000 && threadIdx.x == 0) {
        clk[blockIdx.x * 2 + 0] = __builtin_amdgcn_s_memtime() - c0;
        clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - t0;
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < (SHAPE == 32 ? 16 : 4); ++r) s += acc[n][r];
    if (s == 12345.f) out[0] = s;
}

static int cmp_d(const void* x, const void* y) {
    const double a = *(const double*)x, b = *(const double*)y;
    return a < b ? -1 : a > b;
}

template <int SHAPE, int FEED, int WPS>
static int run(const uint8_t* src, uint64_t row_bytes, float* out, unsigned long long* d_clk, int cus) {
    const int iters = 40000;  // ~10-20 ms per launch: long enough for the clock to settle
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = cus * WPS;
    for (int w = 0; w < 3; ++w)
        hipLaunchKernelGGL((shape_kernel<SHAPE, FEED, WPS>), dim3(grid), dim3(256), 0, 0, src, row_bytes, out, d_clk, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((shape_kernel<SHAPE, FEED, WPS>), dim3(grid), dim3(256), 0, 0, src, row_bytes, out, d_clk, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long* h = (unsigned long long*)malloc(grid * 16);
    CHECK(hipMemcpy(h, d_clk, grid * 16, hipMemcpyDeviceToHost));
    double* ghz = (double*)malloc(grid * sizeof(double));
    for (int i = 0; i < grid; ++i) ghz[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0);  // 100 MHz ticks
    qsort(ghz, grid, sizeof(double), cmp_d);
    // per wave-iteration: 64 x 64 x 256 bit-MACs = 2^21 FLOP; 4 waves per workgroup
    const double flop = (double)grid * 4 * iters * 2097152.0;
    printf("shape %2dx%2d feed %d  %d waves/SIMD: %8.3f ms  %.3f us/wave-iter  clock %.3f GHz (median)  %.3f PFLOP/s\n",
           SHAPE, SHAPE, FEED, WPS, ms, ms * 1e3 / iters, ghz[grid / 2], flop / (ms * 1e-3) / 1e15);
    free(h); free(ghz);
    return 0;
}

template <int SHAPE, int WPS>
static int run_bits(const uint8_t* src, uint64_t row_bytes, float* out, unsigned long long* d_clk, int cus) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = cus * WPS;
    for (int w = 0; w < 3; ++w)
        hipLaunchKernelGGL((bits_kernel<SHAPE, WPS>), dim3(grid), dim3(256), 0, 0, src, row_bytes, out, d_clk, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((bits_kernel<SHAPE, WPS>), dim3(grid), dim3(256), 0, 0, src, row_bytes, out, d_clk, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long* h = (unsigned long long*)malloc(grid * 16);
    CHECK(hipMemcpy(h, d_clk, grid * 16, hipMemcpyDeviceToHost));
    double* ghz = (double*)malloc(grid * sizeof(double));
    for (int i = 0; i < grid; ++i) ghz[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0);
    qsort(ghz, grid, sizeof(double), cmp_d);
    const double flop = (double)grid * 4 * iters * 4194304.0;  // 64 x 64 x 512 bit-MACs per wave-iteration
    printf("shape %2dx%2d feed 4 (bit operands, inflated per wave)  %d waves/SIMD: %8.3f ms  %.3f us/wave-iter  clock %.3f GHz (median)  %.3f PFLOP/s\n",
           SHAPE, SHAPE, WPS, ms, ms * 1e3 / iters, ghz[grid / 2], flop / (ms * 1e-3) / 1e15);
    free(h); free(ghz);
    return 0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const uint64_t row_bytes = 32768;
    uint8_t* src; float* out; unsigned long long* clk;
    CHECK(hipMalloc(&src, 4096 * row_bytes));
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (uint32_t*)src, (size_t)(4096 * row_bytes / 4));
    CHECK(hipDeviceSynchronize());
    uint8_t* src_bits;
    CHECK(hipMalloc(&src_bits, 4096 * row_bytes));
    hipLaunchKernelGGL(fill_random_bits, dim3(4096), dim3(256), 0, 0, (uint32_t*)src_bits, (size_t)(4096 * row_bytes / 4));
    CHECK(hipDeviceSynchronize());
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMalloc(&clk, 4096 * 16));
    for (int rep = 0; rep < 2; ++rep) {
#define BOTH(F, W) \
    if (run<32, F, W>(src, row_bytes, out, clk, cus)) return 1; \
    if (run<16, F, W>(src, row_bytes, out, clk, cus)) return 1;
        BOTH(0, 1) BOTH(0, 2) BOTH(0, 4)
        BOTH(1, 2) BOTH(1, 3) BOTH(1, 4)
        BOTH(2, 4) BOTH(3, 4)
        if (run_bits<16, 2>(src_bits, row_bytes, out, clk, cus)) return 1;
        if (run_bits<16, 3>(src_bits, row_bytes, out, clk, cus)) return 1;
        if (run_bits<32, 2>(src_bits, row_bytes, out, clk, cus)) return 1;
        if (run_bits<32, 3>(src_bits, row_bytes, out, clk, cus)) return 1;
    }
    return 0;
}
