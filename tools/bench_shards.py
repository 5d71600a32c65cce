#!/usr/bin/env python3
"""ONE-GPU REHEARSAL of the multi-GPU work split (no multi-GPU hardware is involved): rank r of `world` is timed on
the same card, rank after rank, for both ownership modes — whole k-slices first (k2_shard_pairs = 0, the default) and
every slice cut along the pair space (k2_shard_pairs = 1, north_star's literal split) — at a given shape. Per world:
the pass of the slowest rank and the scaling it would allow before the 8-byte all-reduce (t(1) / slowest rank); the
sum of the ranks' partial totals is checked against the column identity. Large shapes sample ranks (--ranks-sampled)."""
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
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--modes", default="0,1", help="k2_shard_pairs values to rehearse")
    ap.add_argument("--ranks-sampled", type=int, default=0, help="time only this many ranks per world (first, last, middle ...); 0 = all")
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
    t0 = time.perf_counter()          # clock ramp
    while time.perf_counter() - t0 < 0.05:
        m.pairw_launch(total_t.data_ptr(), 0, 1)
        torch.cuda.synchronize()
    for mode in [int(x) for x in args.modes.split(",")]:
        ctx.set_option("k2_shard_pairs", mode)
        base = None
        for world in [int(w) for w in args.worlds.split(",")]:
            ranks = list(range(world))
            if args.ranks_sampled and world > args.ranks_sampled:
                ranks = sorted({0, world - 1, world // 2, world // 3}.__iter__())[:args.ranks_sampled]
            worst, parts, times = 0.0, 0, []
            for rank in ranks:
                for _ in range(3):
                    m.pairw_launch(total_t.data_ptr(), rank, world)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                for _ in range(args.steps):
                    m.pairw_launch(total_t.data_ptr(), rank, world)
                b.record(stream)
                torch.cuda.synchronize()
                times.append(a.elapsed_time(b) / args.steps)
                parts += int(total_t.item())
            worst = max(times)
            checked = len(ranks) == world
            if checked:
                assert parts == want, (world, parts, want)
            base = base or worst
            print(json.dumps({"one_gpu_rehearsal": True, "rows": args.rows, "bits": args.bits,
                              "ownership": "pair space (k2_shard_pairs=1)" if mode else "k-slices first (default)",
                              "world": world, "ranks_timed": len(ranks), "ms_per_pass_slowest_rank": round(worst, 4),
                              "ms_fastest_rank": round(min(times), 4), "projected_scaling_vs_1": round(base / worst, 3),
                              "partials_sum_to_identity": bool(checked and parts == want) if checked else None,
                              "items_last_rank": ctx.last_launch_info()["items"]}), flush=True)
    m.close()


if __name__ == "__main__":
    main()
