#!/usr/bin/env python3
"""STORM_wrapper_diag_blocked on a host buffer at the headline shape: ms per call (the PCIe-inclusive figure), with the
rows streamed in panels (one device, default) — the storm_benchmark row `bitmap-hip-blocked-31`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stormbitmaps_amd as sb
ctx = sb.HipContext(0)
for N, M in ((10000, 65536), (10000, 524288), (4096, 65536)):
    m = ctx.matrix(N, M // 64)
    m.fill_synthetic(M, M // 2, seed=42)
    want = m.pairw()
    host = m.download()
    m.close()
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        got = sb.wrapper_diag_blocked(host, 31)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"rows": N, "bits": M, "MB": round(host.nbytes / 1e6, 1), "ms_first": round(ts[0], 3), "ms_best": round(min(ts[1:]), 3),
                      "ms_median": round(sorted(ts[1:])[len(ts) // 2], 3), "ok": got == want,
                      "GB_per_s_if_copy_alone": round(host.nbytes / 1e9 / (min(ts[1:]) * 1e-3), 1)}), flush=True)
