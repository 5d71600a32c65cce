import json, sys, time, os
sys.path.insert(0, "/root/repo")
import torch
import stormbitmaps_amd as sb
ctx = sb.HipContext(0, torch.cuda.current_stream().cuda_stream)
t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
for rows, bits in ((10000, 65536), (4096, 65536), (1024, 65536)):
    m = ctx.matrix(rows, bits // 64)
    m.fill_synthetic(bits, bits // 2, seed=42)
    want = m.column_identity()
    stream = torch.cuda.current_stream()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        for _ in range(20):
            m.pairw_launch(t.data_ptr(), 0, 1)
        torch.cuda.synchronize()
    res = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(200):
            m.pairw_launch(t.data_ptr(), 0, 1)
        b.record(stream)
        torch.cuda.synchronize()
        res.append(round(a.elapsed_time(b) * 1e3 / 200, 2))
    print(json.dumps({"lib": os.environ.get("STORM_HIP_LIB", "default"), "rows": rows, "us": res, "ok": int(t.item()) == want}), flush=True)
    m.close()
