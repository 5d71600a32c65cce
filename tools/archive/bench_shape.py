#!/usr/bin/env python3
"""Strip kernel MFMA shape A/B (option k2_shape: 32 = v_mfma 32x32x64, 16 = 16x16x128): totals
against the column identity at several shapes, then alternating timed batches at the headline
shape (kernel-only HIP event time from the library + whole pass)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import stormbitmaps_amd as sb
    stream = torch.cuda.current_stream()
    ctx = sb.HipContext(0, stream.cuda_stream)
    for N, M, d in ((256, 4096, 2048), (700, 9000, 3000), (2049, 65536, 20000), (5000, 131072, 300)):
        m = ctx.matrix(N, (M + 63) // 64)
        m.fill_synthetic(M, d, seed=N)
        want = m.column_identity()
        got = {}
        for shape in (32, 16):
            ctx.set_option("k2_shape", shape)
            got[shape] = m.pairw()
            parts = sum(m.pairw(r, 3) for r in range(3))
            assert got[shape] == want == parts, (N, M, d, shape, got[shape], want, parts)
        print(json.dumps({"check": [N, M, d], "total": want, "ok": True}), flush=True)
        m.close()
    N, M = 10000, 65536
    W = M // 64
    m = ctx.matrix(N, W)
    m.fill_synthetic(M, M // 2, seed=42)
    want = m.column_identity()
    total_t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    for _ in range(200):
        m.pairw_launch(total_t.data_ptr(), 0, 1)
    torch.cuda.synchronize()
    pairs = N * (N - 1) // 2
    for rep in range(4):
        for shape in (32, 16):
            ctx.set_option("k2_shape", shape)
            for _ in range(20):
                m.pairw_launch(total_t.data_ptr(), 0, 1)
            torch.cuda.synchronize()
            ctx.set_option("time_kernels", 1)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(100):
                m.pairw_launch(total_t.data_ptr(), 0, 1)
            b.record(stream)
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 100
            kms, kn = ctx.kernel_time()
            ctx.set_option("time_kernels", 0)
            assert int(total_t.item()) == want
            print(json.dumps({"shape": shape, "rep": rep, "ms_per_pass": round(ms, 4),
                              "strip_kernel_ms": round(kms / kn, 4),
                              "kernel_pflops": round(pairs * W * 128 / (kms / kn * 1e-3) / 1e15, 3)}), flush=True)
    m.close()


if __name__ == "__main__":
    main()
