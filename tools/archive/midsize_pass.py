#!/usr/bin/env python3
"""One mid-size shape, many passes: wall time per pass and the fraction of the FP4 peak it is
(tuning aid; run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--bits", type=int, default=65536)
    ap.add_argument("--passes", type=int, default=200)
    ap.add_argument("--warm-ms", type=float, default=40.0)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    import torch
    import stormbitmaps_amd as sb
    stream = torch.cuda.current_stream()
    ctx = sb.HipContext(0, stream.cuda_stream)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    W = (args.bits + 63) // 64
    t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    m = ctx.matrix(args.rows, W)
    m.fill_synthetic(args.bits, args.bits // 2, seed=42)
    want = m.column_identity()
    import time
    t0 = time.perf_counter()   # clock ramp: ~30 ms of back-to-back passes (profiles/r02_k_bench_nccl_ramp.txt)
    while time.perf_counter() - t0 < args.warm_ms * 1e-3:
        for _ in range(50):
            m.pairw_launch(t.data_ptr(), 0, 1)
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(args.passes):
        m.pairw_launch(t.data_ptr(), 0, 1)
    b.record(stream)
    torch.cuda.synchronize()
    assert int(t.item()) == want, (int(t.item()), want)
    us = a.elapsed_time(b) * 1e3 / args.passes
    flop = args.rows * (args.rows - 1) // 2 * W * 128
    print(json.dumps({"rows": args.rows, "bits": args.bits, "opts": args.opt, "us_per_pass": round(us, 2),
                      "fp4_frac_whole_pass": round(flop / (us * 1e-6) / 1e16, 4),
                      "items": ctx.last_launch_info()["items"], "variant": ctx.get_option("variant_used")}),
          flush=True)
    m.close()
    ctx.close()


if __name__ == "__main__":
    main()
