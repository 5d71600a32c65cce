#!/usr/bin/env python3
"""Materialised output over row counts (LD windows to the headline shape), both tile kernels, M = 65536 dense:
us per call into device memory (launch + completion included) per kernel and the fraction of the FP4 peak of the automatic choice.
bench_matrix_sizes.py [rows,rows,...] [--only shape:slots,...]     (shape 6 = tile128_kernel, 5 = tilering_kernel, 2 = tilebits8_kernel, 0 = auto)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stormbitmaps_amd as sb

ctx = sb.HipContext(0)
M = 65536
FORMS = ((6, 1), (6, 2), (6, 0), (5, 0), (2, 0), (0, 0))
if "--only" in sys.argv:
    FORMS = tuple(tuple(int(v) for v in f.split(":")) for f in sys.argv[sys.argv.index("--only") + 1].split(","))
sizes = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "256,512,1024,2048,3072,4096,6144,8192,10000"


def timed_total(m):
    t0 = time.perf_counter()
    m.pairw()
    return time.perf_counter() - t0


for N in [int(x) for x in sizes.split(",")]:
    m = ctx.matrix(N, M // 64)
    m.fill_synthetic(M, M // 2, seed=42)
    out = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
    want = m.pairw()
    rec = {"rows": N, "bits": M}
    for shape, parts in FORMS:   # 6: tile128_kernel with one / two / automatic segments per CU; 5 / 2: the 256 x 256 kernels, k-parts add into the cleared output; 0: the automatic rule
        ctx.set_option("k2_tile_shape", shape)
        ctx.set_option("k2_part_slots", parts if shape == 6 else 0)
        for _ in range(3):
            m.pairw_matrix_device(out.data_ptr(), N, "and")
        ts = []
        for _ in range(40):
            t0 = time.perf_counter()
            m.pairw_matrix_device(out.data_ptr(), N, "and")
            ts.append(time.perf_counter() - t0)
        t = min(ts)
        rec[{6: f"tile128_slots{parts}_us", 5: "ring_kparts_us", 2: "bits8_kparts_us", 0: "auto_us"}[shape]] = round(t * 1e6, 1)
        if shape == 0:
            rec["auto_kernel"] = ctx.get_option("k2_tile_shape_used")
            rec["auto_frac_fp4_peak"] = round(N * (N - 1) / 2 * (M // 64) * 128 / t / 1e16, 3)
        assert int(out.to(torch.int64).sum().item()) == want
    ctx.set_option("k2_tile_shape", 0)
    ctx.set_option("k2_part_slots", 0)
    rec["all_pairs_total_us"] = round(min(timed_total(m) for _ in range(20)) * 1e6, 1)
    print(json.dumps(rec), flush=True)
    del out
    m.close()
