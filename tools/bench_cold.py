#!/usr/bin/env python3
"""Cold-start costs a real caller sees: library load, context, first and second call of the main entry points."""
import json, os, sys, time
t_start = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stormbitmaps_amd as sb
rec = {}
def lap(name, t0): rec[name] = round((time.perf_counter() - t0) * 1e3, 2)
t0 = time.perf_counter(); lib = sb.load(); lap("load_library_ms", t0)
t0 = time.perf_counter(); ctx = sb.HipContext(0); lap("context_ms", t0)
t0 = time.perf_counter(); m = ctx.matrix(1024, 1024); m.fill_synthetic(65536, 32768, seed=1); ctx.synchronize(); lap("matrix_create_fill_1024_ms", t0)
for k in range(3):
    t0 = time.perf_counter(); v = m.pairw(); lap(f"pairw_1024_call{k}_ms", t0)
m.close()
c = sb.StormContig(65536)
t0 = time.perf_counter(); c.add_synthetic(10000, 32768, seed=42); lap("contig_add_synthetic_c2_ms", t0)
for k in range(3):
    t0 = time.perf_counter(); v = c.pairw_intersect_cardinality_blocked(0); lap(f"contig_c2_call{k}_ms", t0)
c.free()
rows = [np.unique(np.random.default_rng(i).integers(0, 524288, size=524)).astype(np.uint32) for i in range(2000)]
s = sb.Storm()
t0 = time.perf_counter()
for r in rows: s.add(r)
lap("storm_add_2000x524_ms", t0)
for k in range(3):
    t0 = time.perf_counter(); v = s.pairw_intersect_cardinality(); lap(f"storm_call{k}_ms", t0)
for k in range(2):
    t0 = time.perf_counter(); mm = s.pairw_matrix(); lap(f"storm_matrix_call{k}_ms", t0)
s.free()
rec["process_total_s"] = round(time.perf_counter() - t_start, 2)
print(json.dumps(rec))
