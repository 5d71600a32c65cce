// What a hipFree costs against a hipFreeAsync of hipMalloc'ed memory (gfx950, ROCm 7.2):  hipcc -O2 --offload-arch=gfx950 -o free_cost free_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void empty_kernel() {}
int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
    hipStreamSynchronize(s);
    for (int mode = 0; mode < 4; ++mode)
        for (size_t bytes : {(size_t)1 << 16, (size_t)1 << 24, (size_t)64 << 20}) {
            void* p[10];
            const double ta = now_ms();
            for (auto& q : p) {
                if (mode >= 2) { if (hipMallocAsync(&q, bytes, s) != hipSuccess) { printf("hipMallocAsync failed\n"); return 1; } }
                else if (hipMalloc(&q, bytes) != hipSuccess) return 1;
            }
            const double tb = now_ms();
            hipStreamSynchronize(s);
            printf("{\"alloc\": \"%s\", \"bytes\": %zu, \"ten_allocs_ms\": %.3f}\n", mode >= 2 ? "hipMallocAsync" : "hipMalloc", bytes, tb - ta);
            const double t0 = now_ms();
            int rc = 0;
            for (auto& q : p) rc |= (mode == 0 ? hipFree(q) : hipFreeAsync(q, s));
            const double t1 = now_ms();
            hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
            hipStreamSynchronize(s);
            const double t2 = now_ms();
            printf("{\"alloc\": \"%s\", \"free\": \"%s\", \"bytes\": %zu, \"ten_frees_ms\": %.3f, \"next_launch_and_wait_ms\": %.3f, \"rc\": %d}\n",
                   mode >= 2 ? "hipMallocAsync" : "hipMalloc", mode == 0 ? "hipFree" : "hipFreeAsync", bytes, t1 - t0, t2 - t1, rc);
        }
    return 0;
}
