#!/usr/bin/env python3
"""Synchronous calls, one at a time, at the headline shape: the device entry point (storm_hip_pairw_dense on a resident
matrix) against the storm.h entry point on a STORM_contiguous_t holding the same rows."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stormbitmaps_amd as sb

N, M = 10000, 65536
ctx = sb.HipContext(0)
m = ctx.matrix(N, M // 64)
m.fill_synthetic(M, M // 2, seed=42)
want = m.pairw()
def best(f, n=30):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); v = f(); ts.append(time.perf_counter() - t0)
        assert v == want
    ts.sort()
    return round(ts[0] * 1e3, 4), round(ts[len(ts) // 2] * 1e3, 4)
print(json.dumps({"entry": "storm_hip_pairw_dense", "ms_best_median": best(m.pairw)}), flush=True)
c = sb.StormContig(M)
assert c.add_synthetic(N, M // 2, seed=42) == N
print(json.dumps({"entry": "STORM_contig_pairw_intersect_cardinality", "ms_best_median": best(c.pairw_intersect_cardinality)}), flush=True)
print(json.dumps({"entry": "STORM_contig_pairw_intersect_cardinality_blocked", "ms_best_median": best(lambda: c.pairw_intersect_cardinality_blocked(31))}), flush=True)
print(json.dumps({"entry": "storm_hip_pairw_dense (again)", "ms_best_median": best(m.pairw)}), flush=True)
