import sys, time
sys.path.insert(0, "/root/repo")
import stormbitmaps_amd as sb
for d in (20971, 5242):
    s = sb.Storm()
    assert s.add_synthetic(524288, 10000, d, seed=42) == 10000
    t0 = time.perf_counter(); tot = s.pairw_intersect_cardinality(); t1 = time.perf_counter()
    print("draws", d, "first call ms", round((t1 - t0) * 1e3, 1), tot, flush=True)
    s.free()
