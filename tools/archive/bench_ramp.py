import sys, json
sys.path.insert(0, "/root/repo")
import torch, stormbitmaps_amd as sb
stream = torch.cuda.current_stream()
ctx = sb.HipContext(0, stream.cuda_stream)
t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
m = ctx.matrix(10000, 1024); m.fill_synthetic(65536, 32768, seed=42)
torch.cuda.synchronize()
n = 400
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record(stream)
for i in range(n):
    m.pairw_launch(t.data_ptr(), 0, 1)
    ev[i + 1].record(stream)
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for a in (0, 1, 2, 3, 5, 10, 20, 30, 50, 100, 200, 300, 399):
    print(a, round(ts[a], 4))
print("mean first 20 after 3:", sum(ts[3:23]) / 20, "mean 200..400:", sum(ts[200:]) / 200)
