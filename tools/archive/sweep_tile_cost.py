#!/usr/bin/env python3
"""The planner's cost assumptions for diagonal / ragged tiles (k2_tile_cost_*, k2_ring_cost_*) at the headline shape: ms per
materialised-output call for a grid of values, both kernels."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stormbitmaps_amd as sb
ctx = sb.HipContext(0)
N, M = 10000, 65536
m = ctx.matrix(N, M // 64)
m.fill_synthetic(M, M // 2, seed=42)
out = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
def run(reps=40):
    for _ in range(60):
        m.pairw_matrix_device(out.data_ptr(), N, "and")
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        m.pairw_matrix_device(out.data_ptr(), N, "and")
        ts.append(time.perf_counter() - t0)
    return round(min(ts) * 1e3, 4)
for shape, pre in ((2, "k2_tile_cost"), (5, "k2_ring_cost")):
    ctx.set_option("k2_tile_shape", shape)
    for split in (1, 0):
        ctx.set_option("k2_matrix_split", split)
        for diag in ((50, 63, 75, 90) if split else (63,)):
            for rag in ((15, 30, 45, 60) if split else (30,)):
                ctx.set_option(pre + "_diag", diag)
                ctx.set_option(pre + "_ragged", rag)
                print(json.dumps({"k2_tile_shape": shape, "split": split, "cost_diag": diag, "cost_ragged": rag, "ms": run()}), flush=True)
