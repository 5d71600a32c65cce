import json, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import stormbitmaps_amd as sb
ctx = sb.HipContext(0)
M = 65536
for N in (256, 1024, 2048, 4096):
    m = ctx.matrix(N, M // 64)
    m.fill_synthetic(M, M // 2, seed=42)
    out = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
    rec = {"rows": N, "probe": os.environ.get("STORM_TW_PROBE", "0")}
    for wt in (22, 42):
        ctx.set_option("k2_tile_shape", 6)
        ctx.set_option("k2_wave_tile", wt)
        for _ in range(3):
            m.pairw_matrix_device(out.data_ptr(), N, "and")
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            m.pairw_matrix_device(out.data_ptr(), N, "and")
            ts.append(time.perf_counter() - t0)
        rec[f"wave{wt}_us"] = round(min(ts) * 1e6, 1)
    print(json.dumps(rec), flush=True)
    m.close()
