import sys, json
sys.path.insert(0, "/root/repo")
import torch, stormbitmaps_amd as sb
stream = torch.cuda.current_stream()
ctx = sb.HipContext(0, stream.cuda_stream)
t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
for N in (512, 1024, 2048, 3072, 4096, 6144):
    m = ctx.matrix(N, 1024); m.fill_synthetic(65536, 32768, seed=42); want = m.column_identity()
    row = {"rows": N}
    for lpt in (0, 6, 12, 63):
        ctx.set_option("k2_lpt_rounds", lpt)
        for _ in range(5): m.pairw_launch(t.data_ptr(), 0, 1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(100): m.pairw_launch(t.data_ptr(), 0, 1)
        b.record(stream); torch.cuda.synchronize()
        assert int(t.item()) == want
        row[f"lpt{lpt}_us"] = round(a.elapsed_time(b) * 10, 1)
    print(json.dumps(row), flush=True)
    m.close()
