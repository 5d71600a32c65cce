#!/usr/bin/env python3
"""Per-rank cost of the two-level work split (storm_hip_strip_plan) on ONE GPU (rehearsal for the multi-GPU bench): time
storm_hip_pairw_dense_launch for rank r of `world` at the headline shape, for several worlds.
Ideal is t(1)/world; prints the launch time of the slowest rank and the implied scaling."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--bits", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--worlds", default="1,2,4,8")
    args = ap.parse_args()
    import torch
    import stormbitmaps_amd as sb
    stream = torch.cuda.current_stream()
    ctx = sb.HipContext(0, stream.cuda_stream)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    m = ctx.matrix(args.rows, (args.bits + 63) // 64)
    m.fill_synthetic(args.bits, args.bits // 2, seed=42)
    want = m.column_identity()
    total_t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    base = None
    for world in [int(w) for w in args.worlds.split(",")]:
        worst, parts = 0.0, 0
        for rank in range(world):
            for _ in range(5):
                m.pairw_launch(total_t.data_ptr(), rank, world)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(args.steps):
                m.pairw_launch(total_t.data_ptr(), rank, world)
            b.record(stream)
            torch.cuda.synchronize()
            worst = max(worst, a.elapsed_time(b) / args.steps)
            parts += int(total_t.item())
        assert parts == want, (world, parts, want)
        base = base or worst
        print(json.dumps({"world": world, "ms_per_launch_slowest_rank": round(worst, 4),
                          "scaling_vs_1": round(base / worst, 3)}))
    m.close()


if __name__ == "__main__":
    main()
