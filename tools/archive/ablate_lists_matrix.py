#!/usr/bin/env python3
"""Timing ablations of lists_matrix_kernel (option matrix_lists_debug: wrong results by design)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stormbitmaps_amd as sb

def opt(k, v):
    assert sb.load().STORM_hip_set_option(k.encode(), v) == 0

N, M = 10000, 524288
dev = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
opt("matrix_lists", 1)
for d in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "104,524,2621").split(",")]:
    s = sb.Storm()
    assert s.add_synthetic(M, N, d, seed=42) == N
    rec = {"draws": d}
    kern = int(os.environ.get("KERNEL", "1"))
    opt("matrix_lists_kernel", kern)
    rec["kernel"] = kern
    for dbg in ((0, 1, 4, 8, 16, 1 | 4, 1 | 4 | 8, 16 | 8, 0) if kern == 1 else (0, 1, 2, 0)):
        opt("matrix_lists_debug", dbg)
        s.pairw_matrix_device(dev.data_ptr(), N, N)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            s.pairw_matrix_device(dev.data_ptr(), N, N)
            ts.append(time.perf_counter() - t0)
        rec[f"dbg{dbg}"] = round(min(ts) * 1e3, 3)
    opt("matrix_lists_debug", 0)
    print(json.dumps(rec), flush=True)
    s.free()
