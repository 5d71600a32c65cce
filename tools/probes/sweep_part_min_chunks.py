import json, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import stormbitmaps_amd as sb
ctx = sb.HipContext(0)
M = 65536
for N in (256, 512, 1024, 2048, 3072):
    m = ctx.matrix(N, M // 64); m.fill_synthetic(M, M // 2, seed=42)
    out = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
    want = m.pairw()
    rec = {"rows": N}
    ctx.set_option("k2_tile_shape", 6)
    for mc in (2, 4, 8, 16, 32):
        for narrow in (1, 0):
            ctx.set_option("k2_part_min_chunks", mc); ctx.set_option("k2_part_narrow", narrow)
            for _ in range(3): m.pairw_matrix_device(out.data_ptr(), N, "and")
            ts = []
            for _ in range(40):
                t0 = time.perf_counter(); m.pairw_matrix_device(out.data_ptr(), N, "and"); ts.append(time.perf_counter() - t0)
            assert int(out.to(torch.int64).sum().item()) == want
            rec[f"mc{mc}_n{narrow}"] = round(min(ts) * 1e6, 1)
    ctx.set_option("k2_part_min_chunks", 8); ctx.set_option("k2_part_narrow", 1); ctx.set_option("k2_tile_shape", 0)
    print(json.dumps(rec), flush=True)
    m.close()
