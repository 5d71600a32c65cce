#!/usr/bin/env python3
"""One full round of interior tiles (a 4096 x 4096 rectangle = 256 tiles of 256 x 256 over all of k, one per CU) and 2 / 3 / 4
rounds, for both output kernels: what a tile costs when nothing is cut, skipped or added atomically."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import stormbitmaps_amd as sb
from stormbitmaps_amd import _lib
ctx = sb.HipContext(0)
lib = _lib.load()
M = 65536
for ra, rb in ((4096, 4096), (4096, 8192), (4096, 12288), (8192, 8192)):
    a, b = ctx.matrix(ra, M // 64), ctx.matrix(rb, M // 64)
    a.fill_synthetic(M, M // 2, seed=1)
    b.fill_synthetic(M, M // 2, seed=2)
    out = torch.zeros((ra, rb), dtype=torch.int32, device="cuda:0")
    res = {}
    for shape, sync in ((2, 0), (5, 0), (5, 1), (2, 0), (5, 0)):
        ctx.set_option("k2_tile_shape", shape)
        ctx.set_option("k2_ring_sync", sync)
        for _ in range(100):
            _lib.check(lib.storm_hip_square_matrix_device(ctx._h, a._h, b._h, 0, C.c_void_p(out.data_ptr()), rb), "sq")
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            _lib.check(lib.storm_hip_square_matrix_device(ctx._h, a._h, b._h, 0, C.c_void_p(out.data_ptr()), rb), "sq")
            ts.append(time.perf_counter() - t0)
        res.setdefault(f"shape{shape}_sync{sync}", []).append(round(min(ts) * 1e3, 4))
    tiles = (ra // 256) * (rb // 256)
    print(json.dumps({"tiles": tiles, "rounds": tiles / 256, "ms": res,
                      "fp4_frac_shape2": round(ra * rb * (M // 64) * 128 / (min(res["shape2_sync0"]) * 1e-3) / 1e16, 4),
                      "fp4_frac_shape5": round(ra * rb * (M // 64) * 128 / (min(res["shape5_sync0"]) * 1e-3) / 1e16, 4)}), flush=True)
    a.close(); b.close()
