#!/usr/bin/env python3
"""A/B of one context option on the default dense path: alternating timed batches at one shape
(default: the headline shape), whole pass (torch events on the launch stream) and the dominant
kernel alone (the library's own HIP events). Every batch's total is checked against the column
identity.   python tools/bench_ab.py --ab k2_panels=1,4 [--rows N --bits M] [--opt key=value ...]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ab", required=True, help="key=v1,v2,...")
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--bits", type=int, default=65536)
    ap.add_argument("--draws", type=int, default=0)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    import torch
    import stormbitmaps_amd as sb
    key, vals = args.ab.split("=")
    vals = [int(v) for v in vals.split(",")]
    stream = torch.cuda.current_stream()
    ctx = sb.HipContext(0, stream.cuda_stream)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    N, M = args.rows, args.bits
    W = (M + 63) // 64
    m = ctx.matrix(N, W)
    m.fill_synthetic(M, args.draws or M // 2, seed=42)
    want = m.column_identity()
    total_t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    for _ in range(200):
        m.pairw_launch(total_t.data_ptr(), 0, 1)
    torch.cuda.synchronize()
    pairs = N * (N - 1) // 2
    for rep in range(args.reps):
        for v in vals:
            ctx.set_option(key, v)
            for _ in range(20):
                m.pairw_launch(total_t.data_ptr(), 0, 1)
            torch.cuda.synchronize()
            ctx.set_option("time_kernels", 1)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(args.steps):
                m.pairw_launch(total_t.data_ptr(), 0, 1)
            b.record(stream)
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / args.steps
            kms, kn = ctx.kernel_time()
            ctx.set_option("time_kernels", 0)
            got = int(total_t.item())
            print(json.dumps({key: v, "rep": rep, "ms_per_pass": round(ms, 4),
                              "dominant_kernel_ms": round(kms / max(kn, 1), 4),
                              "pass_pflops": round(pairs * W * 128 / (ms * 1e-3) / 1e15, 3),
                              "ok": got == want}), flush=True)
            assert got == want, (key, v, got, want)
    m.close()


if __name__ == "__main__":
    main()
