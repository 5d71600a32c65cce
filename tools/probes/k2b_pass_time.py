"""Time of the all-pairs pass (K2b) WITHOUT checking the total — for ablated tools builds whose totals are wrong by design:
STORM_HIP_LIB=stormbitmaps_amd/libstorm_hip_abl1.so python3 tools/probes/k2b_pass_time.py [rows:bits,...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stormbitmaps_amd as sb
ctx = sb.HipContext(0)
for shape in (sys.argv[1] if len(sys.argv) > 1 else "10000:65536,10000:524288,2048:65536").split(","):
    N, M = (int(v) for v in shape.split(":"))
    m = ctx.matrix(N, M // 64)
    m.fill_synthetic(M, M // 2, seed=42)
    want = m.column_identity()
    got = m.pairw()
    n = 300 if N * M < 2e9 else 60
    for _ in range(n // 3): m.pairw()
    best = 1e9
    for rep in range(5):        # back-to-back launches, queue kept full
        ctx.set_option("time_kernels", 1)
        for _ in range(n): m.pairw()
        ms, k = ctx.kernel_time()
        best = min(best, ms / k)
    ctx.set_option("time_kernels", 0)
    print(json.dumps({"lib": os.environ.get("STORM_HIP_LIB", "libstorm_hip.so"), "rows": N, "bits": M, "kernel_ms_best_of_5_series": round(best, 5),
                      "total_correct": got == want}), flush=True)
    m.close()
