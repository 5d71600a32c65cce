#!/usr/bin/env python3
"""Materialised XX^T upper triangle at the headline shape: time
storm_hip_pairw_matrix_device into a resident N x N uint32 buffer, check that the matrix sums to
the all-pairs total of the summing path, and print one JSON object per op."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--bits", type=int, default=65536)
    ap.add_argument("--draws", type=int, default=32768)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--ops", default="and,or,xor")
    args = ap.parse_args()

    import torch
    import stormbitmaps_amd as sb
    ctx = sb.HipContext(0)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    N, M = args.rows, args.bits
    W = (M + 63) // 64
    m = ctx.matrix(N, W)
    m.fill_synthetic(M, args.draws, seed=42)
    out = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    for op in args.ops.split(","):
        want = m.pairw_op(op)
        m.pairw_matrix_device(out.data_ptr(), N, op)
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            m.pairw_matrix_device(out.data_ptr(), N, op)
            ts.append(time.perf_counter() - t0)
        got = int(out.to(torch.int64).sum().item())
        t = min(ts)
        pairs = N * (N - 1) // 2
        print(json.dumps({"op": op, "rows": N, "bits": M, "draws": args.draws,
                          "ms_per_call": round(t * 1e3, 4), "words_per_s": pairs * 2 * W / t,
                          "bytes_written": pairs * 4, "matrix_sum": got, "pairw_total": want,
                          "match": got == want}))
        assert got == want
    m.close()


if __name__ == "__main__":
    main()
