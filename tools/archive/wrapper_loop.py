#!/usr/bin/env python3
"""STORM_wrapper_diag_blocked on one host buffer at the headline shape, N calls (for traces): wrapper_loop.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stormbitmaps_amd as sb
ctx = sb.HipContext(0)
m = ctx.matrix(10000, 1024)
m.fill_synthetic(65536, 32768, seed=42)
want = m.pairw()
host = m.download()
m.close()
ts = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    t0 = time.perf_counter()
    assert sb.wrapper_diag_blocked(host, 31) == want
    ts.append((time.perf_counter() - t0) * 1e3)
print("ms per call:", [round(t, 3) for t in ts])
