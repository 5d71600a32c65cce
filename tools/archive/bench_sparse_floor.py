#!/usr/bin/env python3
"""Where a small STORM_t all-pairs call spends its time: through storm.h, through the device library's own entry point on
an arena of the same container, and the probe kernel alone (HIP events)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stormbitmaps_amd as sb

lib = sb.load()
ctx = sb.HipContext(0)
for d in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,104,524").split(",")]:
    s = sb.Storm()
    assert s.add_synthetic(524288, 10000, d, seed=42) == 10000
    want = s.pairw_intersect_cardinality_blocked(0)
    def best(fn, n=300):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        ts.sort()
        return round(ts[0] * 1e6, 1), round(ts[n // 2] * 1e6, 1)
    rec = {"load": d, "storm_h_us": best(lambda: s.pairw_intersect_cardinality_blocked(0))}
    data = s.serialize()
    h = C.c_void_p()
    assert lib.storm_hip_sparse_create_serialized(ctx._h, data.ctypes.data_as(C.c_void_p), data.size, C.byref(h)) == 0
    out = C.c_uint64()
    assert lib.storm_hip_pairw_sparse(ctx._h, h, 0, 1, C.byref(out)) == 0 and out.value == want
    rec["device_lib_us"] = best(lambda: lib.storm_hip_pairw_sparse(ctx._h, h, 0, 1, C.byref(out)))
    rec["begin_only_us"] = best(lambda: lib.storm_hip_pairw_sparse_begin(ctx._h, h, 0, 1), 100)
    lib.storm_hip_pairw_sparse_end(ctx._h, C.byref(out))
    ctx.set_option("time_kernels", 1)
    for _ in range(50):
        lib.storm_hip_pairw_sparse(ctx._h, h, 0, 1, C.byref(out))
    ms, n = ctx.kernel_time()
    ctx.set_option("time_kernels", 0)
    rec["kernel_us"] = round(ms / max(n, 1) * 1e3, 2)
    print(json.dumps(rec), flush=True)
    lib.storm_hip_sparse_destroy(ctx._h, h)
    s.free()
