#!/usr/bin/env python3
"""K2q share weights (build_bitstream: shares of one round sized for the SIMD arbiter's oldest-first issue):
pass time at a shape for a list of weight settings, all in one process on one box, interleaved twice.
  --set "w3_1,w3_2": shares of the second and third workgroup of a CU in percent of the first one's."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--bits", type=int, default=65536)
    ap.add_argument("--passes", type=int, default=300)
    ap.add_argument("--set", action="append", default=[])
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    import torch
    import stormbitmaps_amd as sb
    stream = torch.cuda.current_stream()
    ctx = sb.HipContext(0, stream.cuda_stream)
    ctx.set_option("k2_strip_operands", 2)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    W = (args.bits + 63) // 64
    t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    m = ctx.matrix(args.rows, W)
    m.fill_synthetic(args.bits, args.bits // 2, seed=42)
    want = m.column_identity()
    flop = args.rows * (args.rows - 1) // 2 * W * 128
    best = {}
    for rnd in range(2):
        for s in args.set:
            v = [int(x) for x in s.split(",")]
            ctx.set_option("k2_stream_w3_1", v[0])
            ctx.set_option("k2_stream_w3_2", v[1])
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.04:
                for _ in range(50):
                    m.pairw_launch(t.data_ptr(), 0, 1)
                torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(args.passes):
                m.pairw_launch(t.data_ptr(), 0, 1)
            b.record(stream)
            torch.cuda.synchronize()
            assert int(t.item()) == want, (s, int(t.item()), want)
            us = a.elapsed_time(b) * 1e3 / args.passes
            best[s] = min(best.get(s, 1e30), us)
    info = ctx.last_launch_info()
    for s in args.set:
        print(json.dumps({"rows": args.rows, "weights": s, "us_per_pass": round(best[s], 2),
                          "fp4_frac_whole_pass": round(flop / (best[s] * 1e-6) / 1e16, 4), "groups": info["items"]}),
              flush=True)
    m.close()
    ctx.close()


if __name__ == "__main__":
    main()
