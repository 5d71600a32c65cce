#!/usr/bin/env python3
"""All-pairs total (K2b, one launch) over row counts, M = 65536 dense: best and median of 400 calls through the device
library (launch + result word included), and the fraction of the FP4 peak.   bench_pass_sizes.py [rows,rows,...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stormbitmaps_amd as sb

ctx = sb.HipContext(0)
M = 65536
for N in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "256,512,1024,2048,4096,8192,10000").split(",")]:
    m = ctx.matrix(N, M // 64)
    m.fill_synthetic(M, M // 2, seed=42)
    want = m.column_identity()
    assert m.pairw() == want
    ts = []
    for _ in range(400):
        t0 = time.perf_counter()
        got = m.pairw()
        ts.append(time.perf_counter() - t0)
    assert got == want
    ts.sort()
    flop = N * (N - 1) / 2 * (M // 64) * 128
    print(json.dumps({"rows": N, "best_us": round(ts[0] * 1e6, 1), "median_us": round(ts[200] * 1e6, 1),
                      "frac_best": round(flop / ts[0] / 1e16, 3)}), flush=True)
    m.close()
