#!/usr/bin/env python3
"""Per-call wall time of STORM_pairw_intersect_cardinality_blocked at the sparse end of c4 (N = 10000, M = 524288):
best of many steady calls through storm.h, and the same with the fingerprint walk forced on every call."""
import json, os, sys, time
sys.path.insert(0, "/root/repo")
import stormbitmaps_amd as sb

for d in (1, 5, 104, 524, 2097, 5242, 20971):
    s = sb.Storm()
    assert s.add_synthetic(524288, 10000, d, seed=42) == 10000
    want = s.pairw_intersect_cardinality_blocked(0)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter()
        got = s.pairw_intersect_cardinality_blocked(0)
        ts.append(time.perf_counter() - t0)
    assert got == want
    ts.sort()
    print(json.dumps({"load": d, "best_us": round(ts[0] * 1e6, 1), "median_us": round(ts[len(ts) // 2] * 1e6, 1), "total": got}), flush=True)
    s.free()
