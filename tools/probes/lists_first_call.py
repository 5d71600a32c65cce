import json, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import stormbitmaps_amd as sb
sb.load()
N = 10000
mode = sys.argv[1] if len(sys.argv) > 1 else "base"
dev = torch.zeros((N, N), dtype=torch.int32, device="cuda:0"); torch.cuda.synchronize()
s = sb.Storm()
s.add_synthetic(524288, N, 524, seed=42)
if mode == "wake":
    x = torch.zeros(1 << 20, device="cuda:0"); x += 1; torch.cuda.synchronize()
if mode == "wake1":
    t0 = time.perf_counter(); x = torch.zeros(16, device="cuda:0"); x += 1; torch.cuda.synchronize(); print("wake op ms", round((time.perf_counter() - t0) * 1e3, 2))
if mode == "wake1ns":
    x = torch.zeros(16, device="cuda:0"); x += 1
if mode.startswith("ours"):      # a small pass of our own (256 rows x 1024 bits), then a host pause of <n> ms: ours30, ours5, ours0
    ctx = sb.HipContext(0); m = ctx.matrix(256, 16); m.fill_synthetic(1024, 100, seed=1); ctx.synchronize()
    time.sleep(0.5)
    t0 = time.perf_counter(); m.pairw(); print("small pass ms", round((time.perf_counter() - t0) * 1e3, 2), end=" ")
    time.sleep(int(mode[4:]) / 1e3)
if mode == "synconly":
    t0 = time.perf_counter(); torch.cuda.synchronize(); print("sync ms", round((time.perf_counter() - t0) * 1e3, 2))
if mode == "sleep":
    time.sleep(0.5)
if mode == "wakebig":
    t0 = time.perf_counter(); dev.add_(1); torch.cuda.synchronize(); print("wake op ms", round((time.perf_counter() - t0) * 1e3, 2)); dev.zero_(); torch.cuda.synchronize()
rec = {"mode": mode}
for k in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter(); s.pairw_matrix_device(dev.data_ptr(), N, N); w = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    rec[f"call{k}"] = (round(w * 1e3, 2), round(e0.elapsed_time(e1), 2))
print(json.dumps(rec))
