import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stormbitmaps_amd as sb
lib = sb.load()
N, M = 10000, 524288
dev = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
for d in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "104,524,2096,3670").split(",")]:
    s = sb.Storm()
    s.add_synthetic(M, N, d, seed=42)
    rec = {"positions_per_row": d}
    for k in range(3):
        t0 = time.perf_counter(); s.pairw_matrix_device(dev.data_ptr(), N, N); rec[f"matrix_call{k}_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    rec["sum"] = int(dev.to(torch.int64).sum().item()); rec["total"] = s.pairw_intersect_cardinality()
    assert rec["sum"] == rec["total"]
    print(json.dumps(rec), flush=True)
    s.free()
