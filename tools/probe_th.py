import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stormbitmaps_amd as sb
ctx = sb.HipContext(0)
M = 65536
for N in (256, 1024, 2048):
    m = ctx.matrix(N, M // 64)
    m.fill_synthetic(M, M // 2, seed=42)
    out = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
    ctx.set_option("k2_tile_shape", 6)
    for slots, minc, narrow, poll in ((1, 8, 1, 300), (1, 8, 0, 300), (1, 8, 1, 0), (2, 8, 1, 300), (2, 16, 1, 300), (0, 8, 1, 300), (1, 12, 1, 300), (1, 6, 1, 300)):
        ctx.set_option("k2_part_slots", slots)
        ctx.set_option("k2_part_min_chunks", minc)
        ctx.set_option("k2_part_narrow", narrow)
        ctx.set_option("sync_poll_us", poll)
        for _ in range(3):
            m.pairw_matrix_device(out.data_ptr(), N, "and")
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            m.pairw_matrix_device(out.data_ptr(), N, "and")
            ts.append(time.perf_counter() - t0)
        print(json.dumps({"rows": N, "slots": slots, "min_chunks": minc, "narrow": narrow, "poll": poll, "call_us": round(min(ts) * 1e6, 1)}), flush=True)
    m.close()
