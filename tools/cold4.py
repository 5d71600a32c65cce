import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stormbitmaps_amd as sb
rec = {}
def lap(name, t0): rec[name] = round((time.perf_counter() - t0) * 1e3, 3)
lib = sb.load()
what = sys.argv[1] if len(sys.argv) > 1 else "contig"
if what == "contig":
    c = sb.StormContig(65536)
    t0 = time.perf_counter(); c.add_synthetic(10000, 32768, seed=42); lap("contig_add_synthetic_c2_ms", t0)
    time.sleep(0.05)
    for k in range(3):
        t0 = time.perf_counter(); v = c.pairw_intersect_cardinality_blocked(0); lap(f"contig_c2_call{k}_ms", t0)
    c.free()
else:
    N, M, d = 10000, 524288, int(sys.argv[2]) if len(sys.argv) > 2 else 262144
    s = sb.Storm()
    t0 = time.perf_counter(); s.add_synthetic(M, N, d, seed=42); lap("storm_add_synthetic_ms", t0)
    time.sleep(0.05)
    for k in range(3):
        t0 = time.perf_counter(); v = s.pairw_intersect_cardinality_blocked(0); lap(f"storm_call{k}_ms", t0)
    s.free()
print(json.dumps(rec))
