"""First / second / third all-pairs call of a STORM_contiguous_t in a fresh process: contig_first_call.py <bits> <rows> [pause_ms]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stormbitmaps_amd as sb
sb.load()
M, N = int(sys.argv[1]), int(sys.argv[2])
c = sb.StormContig(M)
c.add_synthetic(N, M // 2, seed=42)
time.sleep((int(sys.argv[3]) if len(sys.argv) > 3 else 50) / 1e3)
rec = {"bits": M, "rows": N}
for k in range(3):
    t0 = time.perf_counter(); c.pairw_intersect_cardinality_blocked(0); rec[f"call{k}_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
print(json.dumps(rec))
