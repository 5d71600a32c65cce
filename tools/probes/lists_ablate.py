"""K5 window kernel at the c4 shape with the kernel's ablation bits (matrix_lists_debug: 1 no element is loaded, 4 no barrier,
8 no store, 16 no step loop): ms per call. WRONG results by design for bits != 0.   lists_ablate.py [draws,draws]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import stormbitmaps_amd as sb
lib = sb.load()
N, M = 10000, 524288
dev = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
for d in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "524,2096").split(",")]:
    s = sb.Storm(); s.add_synthetic(M, N, d, seed=42)
    lib.STORM_hip_set_option(b"matrix_lists", 1); lib.STORM_hip_set_option(b"matrix_lists_kernel", 1)
    rec = {"draws": d}
    for dbg in (0, 1, 8, 9, 16):
        lib.STORM_hip_set_option(b"matrix_lists_debug", dbg)
        for _ in range(2): s.pairw_matrix_device(dev.data_ptr(), N, N)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); s.pairw_matrix_device(dev.data_ptr(), N, N); ts.append(time.perf_counter() - t0)
        rec[f"debug_{dbg}_ms"] = round(min(ts) * 1e3, 3)
    lib.STORM_hip_set_option(b"matrix_lists_debug", 0)
    print(json.dumps(rec), flush=True)
    s.free()
lib.STORM_hip_set_option(b"matrix_lists", -1); lib.STORM_hip_set_option(b"matrix_lists_kernel", 0)
