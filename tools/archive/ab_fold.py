#!/usr/bin/env python3
"""The fold inside the K2b launch (k2_fold_inline = -1, round 5) against the fold launch behind it (0): us per pass, interleaved,
same box; N = 10000 / 4096 / 1024 / 256 at M = 65536."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stormbitmaps_amd as sb
ctx = sb.HipContext(0, torch.cuda.current_stream().cuda_stream)
t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
for rows, bits in ((10000, 65536), (4096, 65536), (2048, 65536), (1024, 65536), (512, 65536), (256, 65536)):
    m = ctx.matrix(rows, bits // 64)
    m.fill_synthetic(bits, bits // 2, seed=42)
    want = m.column_identity()
    stream = torch.cuda.current_stream()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(20):
            m.pairw_launch(t.data_ptr(), 0, 1)
        torch.cuda.synchronize()
    res = {0: [], -1: [], 1: []}
    ok = True
    for rep in range(4):
        for fold in (0, -1, 1):
            ctx.set_option("k2_fold_inline", fold)
            n = 300 if rows >= 4096 else 2000
            for _ in range(20):
                m.pairw_launch(t.data_ptr(), 0, 1)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(n):
                m.pairw_launch(t.data_ptr(), 0, 1)
            b.record(stream)
            torch.cuda.synchronize()
            res[fold].append(round(a.elapsed_time(b) * 1e3 / n, 2))
            ok = ok and int(t.item()) == want
    print(json.dumps({"rows": rows, "bits": bits, "us_fold_launch": res[0], "us_fold_auto": res[-1], "us_fold_in_kernel": res[1], "ok": ok}), flush=True)
    m.close()
