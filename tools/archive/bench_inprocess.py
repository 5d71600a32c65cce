#!/usr/bin/env python3
"""In-process multi-device path of the storm.h containers (STORM_hip_set_devices / STORM_HIP_DEVICES), rehearsed on
ONE GPU: the same device ordinal configured G times gives G contexts, streams and host worker threads. Two
figures per G: wall time of STORM_contig_pairw_intersect_cardinality_blocked on the headline matrix (GPU-bound
on one card: G shards share it) and on a 256-row matrix, where the GPU work is a few microseconds and the call
time is the HOST's cost of driving G devices (launches, result reads, thread hand-off)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import stormbitmaps_amd as sb
    lib = sb.load()
    M = 65536
    for G in (1, 2, 4, 8):
        ids = (C.c_int * G)(*([0] * G))
        assert lib.STORM_hip_set_devices(G, ids) == 0
        row = {"devices_configured": G}
        for name, N, reps in (("headline_10000_rows_ms", 10000, 30), ("host_cost_256_rows_us", 256, 300)):
            c = sb.StormContig(M)
            assert c.add_synthetic(N, M // 2, seed=42) == N
            want = c.pairw_intersect_cardinality_blocked(31)
            for _ in range(5):
                assert c.pairw_intersect_cardinality_blocked(31) == want
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                got = c.pairw_intersect_cardinality_blocked(31)
                ts.append(time.perf_counter() - t0)
                assert got == want
            ts.sort()
            row[name] = round(ts[len(ts) // 2] * (1e3 if name.endswith("_ms") else 1e6), 3 if name.endswith("_ms") else 1)
            row[name.rsplit("_", 1)[0] + "_total"] = want
            c.free()
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
