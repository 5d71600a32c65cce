#!/usr/bin/env python3
"""tile16_bits_kernel (k2_tile_shape = 3) against tilebits8_kernel (2) and the oracle: triangle, ops, rectangle,
ragged edges, rows of zero; then time per call at the headline shape for both."""
import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
import stormbitmaps_amd as sb
from stormbitmaps_amd import synth
from tests._orc import Oracle

orc = Oracle()
ctx = sb.HipContext(0)
bad = 0
for M, N, d in ((4096, 256, 2048), (640, 65, 200), (1000, 257, 300), (9000, 700, 3000), (65536, 513, 9000), (300, 130, 100),
                (70000, 1029, 20000), (512, 300, 100), (520, 300, 100), (65536, 2000, 20000)):
    mat = synth.dense_matrix_c(M, N, d, seed=N + M)
    mat[N // 3] = 0
    m = ctx.matrix_from_host(mat)
    for op in ("and", "or", "xor"):
        ctx.set_option("k2_tile_shape", 2)
        ref = m.pairw_matrix(op)
        ctx.set_option("k2_tile_shape", 3)
        got = m.pairw_matrix(op)
        ctx.set_option("k2_tile_shape", 4)
        got2 = m.pairw_matrix(op)
        ok = np.array_equal(ref, got) and np.array_equal(ref, got2)
        if N <= 300 and op == "and":
            want = np.triu(orc.tile_counts(mat, 0, N, 0, N), k=1).astype(np.uint32)
            ok = ok and np.array_equal(want, got)
        bad += not ok
        print(M, N, d, op, "OK" if ok else f"FAIL mismatches {int((ref != got).sum())} first {np.argwhere(ref != got)[:3].tolist()}", flush=True)
    if N >= 257:
        na = N // 2
        ma, mb = ctx.matrix_from_host(mat[:na]), ctx.matrix_from_host(mat[na:])
        ctx.set_option("k2_tile_shape", 2)
        ref = ma.square_matrix(mb, "and")
        ctx.set_option("k2_tile_shape", 3)
        got = ma.square_matrix(mb, "and")
        ctx.set_option("k2_tile_shape", 4)
        ok = np.array_equal(ref, got) and np.array_equal(ref, ma.square_matrix(mb, "and"))
        bad += not ok
        print(M, N, d, "square", "OK" if ok else f"FAIL {int((ref != got).sum())}", flush=True)
        ma.close(); mb.close()
    m.close()
print("BAD", bad, flush=True)
N, M = 10000, 65536
m = ctx.matrix(N, M // 64)
m.fill_synthetic(M, M // 2, seed=42)
out = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
want = m.pairw()
for shape in (2, 3, 4, 2, 3, 4):
    ctx.set_option("k2_tile_shape", shape)
    for _ in range(5):
        m.pairw_matrix_device(out.data_ptr(), N, "and")
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        m.pairw_matrix_device(out.data_ptr(), N, "and")
        ts.append(time.perf_counter() - t0)
    got = int(out.to(torch.int64).sum().item())
    t = min(ts)
    print(json.dumps({"k2_tile_shape": shape, "ms_per_call": round(t * 1e3, 4), "fp4_frac": round(N * (N - 1) // 2 * (M // 64) * 128 / t / 1e16, 4), "match": got == want}), flush=True)
sys.exit(1 if bad else 0)
