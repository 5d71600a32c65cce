import sys, json, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, stormbitmaps_amd as sb
lib = sb.load()
M, N = 524288, 10000
# 9000 rows of short lists (524 draws) and 1000 bitmap rows (262144 draws), interleaved
s = sb.Storm()
assert s.add_synthetic(M, N, 524, seed=42) == N
data_lists = s.serialize(); s.free()
rng = np.random.default_rng(1)
s = sb.Storm()
for r in range(N):
    d = 262144 if r % 10 == 0 else 524
    s.add(np.unique(rng.integers(0, M, size=d, dtype=np.uint64)).astype(np.uint32))
data = s.serialize(); s.free()
ctx = sb.HipContext(0)
h = C.c_void_p()
assert lib.storm_hip_sparse_create_serialized(ctx._h, data.ctypes.data_as(C.c_void_p), data.size, C.byref(h)) == 0
out = C.c_uint64(); row = {}
for probe, name in ((0, "dense_everything_ms"), (-1, "per_block_split_ms")):
    ctx.set_option("sparse_probe", probe)
    for _ in range(3): assert lib.storm_hip_pairw_sparse(ctx._h, h, 0, 1, C.byref(out)) == 0
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); assert lib.storm_hip_pairw_sparse(ctx._h, h, 0, 1, C.byref(out)) == 0; ts.append(time.perf_counter() - t0)
    row[name] = round(min(ts) * 1e3, 3); row[name.replace("_ms", "_total")] = out.value
assert row["dense_everything_total"] == row["per_block_split_total"]
print(json.dumps({"container": "STORM_t N=10000 M=524288: every 10th row 262144 draws (bitmap blocks), the others 524 (lists)", **row}))
