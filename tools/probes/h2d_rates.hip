// Host-to-device copy rates on the GPU box: pageable hipMemcpy, pinned hipMemcpy, and std::memcpy pageable -> pinned with
// 1 / 2 / 4 / 8 threads — what a staged, multi-threaded upload of the raw-buffer wrappers could reach.
//   hipcc -O2 -o h2d_rates tools/probes/h2d_rates.hip -lpthread && ./h2d_rates [MB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 82) << 20;
    char* pageable = (char*)malloc(bytes);
    memset(pageable, 1, bytes);
    char *pinned = nullptr, *dev = nullptr;
    if (hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&dev, bytes) != hipSuccess) return 1;
    memset(pinned, 2, bytes);
    auto best = [&](auto fn) { double b = 1e9; for (int r = 0; r < 8; ++r) { double t0 = now(); fn(); b = std::min(b, now() - t0); } return b; };
    double t = best([&] { (void)hipMemcpy(dev, pageable, bytes, hipMemcpyHostToDevice); });
    printf("{\"MB\": %zu, \"pageable_hipMemcpy_GBs\": %.1f", bytes >> 20, bytes / t / 1e9);
    t = best([&] { (void)hipMemcpy(dev, pinned, bytes, hipMemcpyHostToDevice); });
    printf(", \"pinned_hipMemcpy_GBs\": %.1f", bytes / t / 1e9);
    for (int nt : {1, 2, 4, 8}) {
        t = best([&] {
            std::vector<std::thread> th;
            for (int k = 0; k < nt; ++k)
                th.emplace_back([&, k] { size_t a = bytes * k / nt, b = bytes * (k + 1) / nt; memcpy(pinned + a, pageable + a, b - a); });
            for (auto& x : th) x.join();
        });
        printf(", \"memcpy_%dthreads_GBs\": %.1f", nt, bytes / t / 1e9);
    }
    // staged pipeline: 8 panels, memcpy with 4 threads into one of two pinned halves while the other half's DMA runs
    {
        hipStream_t s; (void)hipStreamCreate(&s);
        const int panels = 8, nt = 4;
        const size_t pb = bytes / panels;
        hipEvent_t ev[2]; (void)hipEventCreate(&ev[0]); (void)hipEventCreate(&ev[1]);
        t = best([&] {
            for (int p = 0; p < panels; ++p) {
                char* stage = pinned + (size_t)(p & 1) * pb;
                if (p >= 2) (void)hipEventSynchronize(ev[p & 1]);
                std::vector<std::thread> th;
                for (int k = 0; k < nt; ++k)
                    th.emplace_back([&, k] { size_t a = pb * k / nt, b = pb * (k + 1) / nt; memcpy(stage + a, pageable + p * pb + a, b - a); });
                for (auto& x : th) x.join();
                (void)hipMemcpyAsync(dev + p * pb, stage, pb, hipMemcpyHostToDevice, s);
                (void)hipEventRecord(ev[p & 1], s);
            }
            (void)hipStreamSynchronize(s);
        });
        printf(", \"staged_8panels_4threads_GBs\": %.1f", bytes / t / 1e9);
    }
    {   // what the wrappers do: 8 panels, pageable source; 1-D async copies, 2-D async copies (8192-byte rows into a padded pitch)
        hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        const int panels = 8;
        const size_t pb = bytes / panels;
        t = best([&] {
            for (int p = 0; p < panels; ++p) (void)hipMemcpyAsync(dev + p * pb, pageable + p * pb, pb, hipMemcpyHostToDevice, s);
            (void)hipStreamSynchronize(s);
        });
        printf(", \"pageable_async_1d_8panels_GBs\": %.1f", bytes / t / 1e9);
        const size_t row = 8192, pitch = 8192 + 512, rows = bytes / pitch / panels;
        t = best([&] {
            for (int p = 0; p < panels; ++p)
                (void)hipMemcpy2DAsync(dev + p * rows * pitch, pitch, pageable + p * rows * row, row, row, rows, hipMemcpyHostToDevice, s);
            (void)hipStreamSynchronize(s);
        });
        printf(", \"pageable_async_2d_8panels_GBs\": %.1f", (double)rows * panels * row / t / 1e9);
        t = best([&] {
            for (int p = 0; p < panels; ++p)
                (void)hipMemcpy2DAsync(dev + p * rows * pitch, pitch, pinned + p * rows * row, row, row, rows, hipMemcpyHostToDevice, s);
            (void)hipStreamSynchronize(s);
        });
        printf(", \"pinned_async_2d_8panels_GBs\": %.1f", (double)rows * panels * row / t / 1e9);
        t = best([&] { (void)hipMemcpy2D(dev, pitch, pageable, row, row, rows * panels, hipMemcpyHostToDevice); });
        printf(", \"pageable_sync_2d_GBs\": %.1f", (double)rows * panels * row / t / 1e9);
    }
    {   // register the caller's buffer for the duration of one call: register, 8 asynchronous 2-D panel copies, unregister
        hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        const int panels = 8;
        const size_t row = 8192, pitch = 8192 + 512, rows = bytes / pitch / panels;
        double t_reg = 1e9, t_unreg = 1e9, t_issue = 1e9;
        t = best([&] {
            double a = now();
            hipError_t e = hipHostRegister(pageable, bytes, hipHostRegisterDefault);
            double b = now();
            if (e != hipSuccess) { printf(", \"register_failed\": %d", (int)e); return; }
            for (int p = 0; p < panels; ++p)
                (void)hipMemcpy2DAsync(dev + p * rows * pitch, pitch, pageable + p * rows * row, row, row, rows, hipMemcpyHostToDevice, s);
            t_issue = std::min(t_issue, now() - b);
            (void)hipStreamSynchronize(s);
            double c = now();
            (void)hipHostUnregister(pageable);
            double d = now();
            t_reg = std::min(t_reg, b - a);
            t_unreg = std::min(t_unreg, d - c);
        });
        printf(", \"registered_2d_8panels_total_GBs\": %.1f, \"register_us\": %.0f, \"unregister_us\": %.0f, \"registered_issue_8_copies_us\": %.0f", (double)rows * panels * row / t / 1e9, t_reg * 1e6, t_unreg * 1e6, t_issue * 1e6);
        double t_issue2 = 1e9;
        t = best([&] {
            double b = now();
            for (int p = 0; p < panels; ++p)
                (void)hipMemcpy2DAsync(dev + p * rows * pitch, pitch, pageable + p * rows * row, row, row, rows, hipMemcpyHostToDevice, s);
            t_issue2 = std::min(t_issue2, now() - b);
            (void)hipStreamSynchronize(s);
        });
        printf(", \"pageable_issue_8_copies_us\": %.0f", t_issue2 * 1e6);
    }
    {   // what sits between the wrapper's panel copies: an event on the copy stream + a wait and a kernel on a second stream
        hipStream_t sc, sk; (void)hipStreamCreateWithFlags(&sc, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sk, hipStreamNonBlocking);
        const int panels = 8;
        const size_t row = 8192, pitch = 8192 + 512, rows = bytes / pitch / panels;
        hipEvent_t ev[8];
        for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        for (int mode = 0; mode < 3; ++mode) {
            t = best([&] {
                for (int p = 0; p < panels; ++p) {
                    (void)hipMemcpy2DAsync(dev + p * rows * pitch, pitch, pageable + p * rows * row, row, row, rows, hipMemcpyHostToDevice, sc);
                    if (mode >= 1) (void)hipEventRecord(ev[p], sc);
                    if (mode >= 2) { (void)hipStreamWaitEvent(sk, ev[p], 0); (void)hipMemsetAsync(dev + bytes - 64, 0, 64, sk); }
                }
                (void)hipStreamSynchronize(sc);
                (void)hipStreamSynchronize(sk);
            });
            printf(", \"panels_mode%d_GBs\": %.1f", mode, (double)rows * panels * row / t / 1e9);
        }
        // the same dependency through stream memory operations: a flag written behind every copy, waited for on the other stream
        int can = 0;
        (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
        uint32_t* flag = nullptr;
        if (can && hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory) == hipSuccess) {
            uint32_t seq = 0;
            t = best([&] {
                for (int p = 0; p < panels; ++p) {
                    (void)hipMemcpy2DAsync(dev + p * rows * pitch, pitch, pageable + p * rows * row, row, row, rows, hipMemcpyHostToDevice, sc);
                    (void)hipStreamWriteValue32(sc, flag, ++seq, 0);
                    (void)hipStreamWaitValue32(sk, flag, seq, hipStreamWaitValueGte, 0xffffffffu);
                    (void)hipMemsetAsync(dev + bytes - 64, 0, 64, sk);
                }
                (void)hipStreamSynchronize(sc);
                (void)hipStreamSynchronize(sk);
            });
            printf(", \"panels_stream_values_GBs\": %.1f", (double)rows * panels * row / t / 1e9);
        } else {
            printf(", \"stream_wait_value\": %d", can);
        }
    }
    printf("}\n");
    return 0;
}
