#!/usr/bin/env python3
"""Mid-size N at M = 65536 (the size class of the reference's CI run, .travis.yml:193-201): whole-pass time of
storm_hip_pairw_dense_launch, data resident, on the default path (K2q: one launch on bit operands up to 8192
rows) and on the FP4 strips (k2_strip_operands = 4: expansion + strips + fold), same box, ~40 ms of warm-up per
case (the chip's clock needs it), and the fraction of the FP4 matrix peak (10 PFLOP/s) the whole pass is."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import stormbitmaps_amd as sb  # noqa: E402

stream = torch.cuda.current_stream()
ctx = sb.HipContext(0, stream.cuda_stream)
t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
M, W = 65536, 1024
for N in (256, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 10000):
    m = ctx.matrix(N, W)
    m.fill_synthetic(M, M // 2, seed=42)
    want = m.column_identity()
    flop = N * (N - 1) // 2 * W * 128
    row = {"rows": N}
    for name, operands in (("default", 0), ("fp4_strips", 4)):
        ctx.set_option("k2_strip_operands", operands)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.04:
            for _ in range(50):
                m.pairw_launch(t.data_ptr(), 0, 1)
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        a.record(stream)
        for _ in range(n):
            m.pairw_launch(t.data_ptr(), 0, 1)
        b.record(stream)
        torch.cuda.synchronize()
        assert int(t.item()) == want
        us = a.elapsed_time(b) * 1e3 / n
        row[f"{name}_us"] = round(us, 2)
        row[f"{name}_fp4_frac"] = round(flop / (us * 1e-6) / 1e16, 4)
        if operands == 0:
            row["default_kernel"] = {2: "bitstream_kernel (K2q)", 4: "strip16_fp4_kernel", 1: "stripbits_kernel"}[
                ctx.get_option("k2_operands_used")]
            row["default_workgroups"] = ctx.last_launch_info()["items"]
    ctx.set_option("k2_strip_operands", 0)
    print(json.dumps(row), flush=True)
    m.close()
