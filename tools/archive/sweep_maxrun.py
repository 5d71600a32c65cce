#!/usr/bin/env python3
"""k2_max_run 64 / 96 / 128 over N (M = 65536), whole pass of the default path."""
import json, sys, time
sys.path.insert(0, "/root/repo")
import torch
import stormbitmaps_amd as sb
ctx = sb.HipContext(0, torch.cuda.current_stream().cuda_stream)
stream = torch.cuda.current_stream()
t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
for rows in (3072, 4096, 5120, 6144, 7168, 8192, 9216, 10000, 12000, 16000):
    m = ctx.matrix(rows, 1024)
    m.fill_synthetic(65536, 32768, seed=42)
    r = {"rows": rows}
    for mr in (64, 96, 128, 0):
        ctx.set_option("k2_max_run", mr)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.03:
            for _ in range(10):
                m.pairw_launch(t.data_ptr(), 0, 1)
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(60):
            m.pairw_launch(t.data_ptr(), 0, 1)
        b.record(stream)
        torch.cuda.synchronize()
        r[f"run{mr}"] = round(a.elapsed_time(b) * 1e3 / 60, 1)
        r[f"items{mr}"] = ctx.last_launch_info()["items"]
    print(json.dumps(r), flush=True)
    m.close()
