// What does freeing host memory right after a pageable hipMemcpyAsync cost the NEXT kernel launch?  (gfx950, ROCm 7.2)
// For each size: mmap a buffer, copy it to the device (or from it) on a stream, wait, munmap, then time an empty kernel + wait.
//   hipcc --offload-arch=gfx950 -O2 -o pin_evict pin_evict.hip && ./pin_evict
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
__global__ void empty_kernel() {}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    hipStream_t s;
    hipStreamCreate(&s);
    void* d = nullptr;
    hipMalloc(&d, 64u << 20);
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
    hipStreamSynchronize(s);
    for (int dir = 0; dir < 2; ++dir)
        for (int keep = 0; keep < 2; ++keep)
            for (size_t bytes : {16u << 10, 64u << 10, 256u << 10, 1u << 20, 4u << 20, 16u << 20}) {
                double worst = 0, sum = 0;
                const int reps = 4;
                for (int r = 0; r < reps; ++r) {
                    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                    memset(p, 1, bytes);
                    if (dir == 0) hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, s);
                    else hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, s);
                    hipStreamSynchronize(s);
                    if (!keep) munmap(p, bytes);
                    const double t0 = now_ms();
                    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
                    hipStreamSynchronize(s);
                    const double dt = now_ms() - t0;
                    worst = dt > worst ? dt : worst;
                    sum += dt;
                    if (keep) munmap(p, bytes);
                }
                printf("{\"direction\": \"%s\", \"bytes\": %zu, \"freed_before_the_launch\": %s, \"next_launch_ms_mean\": %.3f, \"next_launch_ms_worst\": %.3f}\n",
                       dir ? "d2h" : "h2d", bytes, keep ? "false" : "true", sum / reps, worst);
            }
    return 0;
}
