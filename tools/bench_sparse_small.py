#!/usr/bin/env python3
"""Per-call wall time of STORM_pairw_intersect_cardinality_blocked at the sparse end of c4 (N = 10000, M = 524288):
best of many steady calls through storm.h, and the same with the fingerprint walk forced on every call."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stormbitmaps_amd as sb

lib = sb.load()
for d in (1, 5, 104, 524, 2097, 5242, 20971):
    s = sb.Storm()
    assert s.add_synthetic(524288, 10000, d, seed=42) == 10000
    rec = {"load": d}
    for bundle in (1, 4):   # probe_lists_kernel (one group of 128 rows per workgroup) / probe_lists_fat_kernel (four) [r6]
        assert lib.STORM_hip_set_option(b"probe_bundle", bundle) == 0
        want = s.pairw_intersect_cardinality_blocked(0)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter()
            got = s.pairw_intersect_cardinality_blocked(0)
            ts.append(time.perf_counter() - t0)
        assert got == want
        ts.sort()
        rec[f"bundle{bundle}_best_us"] = round(ts[0] * 1e6, 1)
        rec[f"bundle{bundle}_median_us"] = round(ts[len(ts) // 2] * 1e6, 1)
        rec.setdefault("total", got)
        assert rec["total"] == got
    print(json.dumps(rec), flush=True)
    s.free()
lib.STORM_hip_set_option(b"probe_bundle", -1)
