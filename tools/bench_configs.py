#!/usr/bin/env python3
"""All dense BASELINE configs on one GPU with the default path (c4, the sparse container, is
tools/archive/bench_sparse.py): per config one JSON line with the pass time (HIP events on the launch
stream, data resident), words/s, and the total checked against the column identity."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c1": (256, 4096, 50),          # benchmark 4096 256
    "c2": (10000, 65536, 50),       # README headline
    "c3": (10000, 524288, 10),
    "c5": (100000, 1048576, 2),     # genomics scale, one GPU's worth of the 8-GPU config
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c1,c2,c3,c5")
    args = ap.parse_args()
    import torch
    import stormbitmaps_amd as sb
    stream = torch.cuda.current_stream()
    ctx = sb.HipContext(0, stream.cuda_stream)
    total_t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    for name in args.configs.split(","):
        N, M, steps = CONFIGS[name]
        W = (M + 63) // 64
        m = ctx.matrix(N, W)
        m.fill_synthetic(M, M // 2, seed=42)
        want = m.column_identity()
        m.pairw_launch(total_t.data_ptr(), 0, 1)
        torch.cuda.synchronize()
        ctx.set_option("time_kernels", 1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(steps):
            m.pairw_launch(total_t.data_ptr(), 0, 1)
        b.record(stream)
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / steps
        kms, kn = ctx.kernel_time()
        ctx.set_option("time_kernels", 0)
        pairs = N * (N - 1) // 2
        got = int(total_t.item())
        print(json.dumps({"config": name, "rows": N, "bits": M, "variant": ctx.get_option("variant_used"),
                          "ms_per_pass": round(ms, 4), "dominant_kernel_ms": round(kms / max(kn, 1), 4),
                          "words_per_s": pairs * 2 * W / (ms * 1e-3),
                          "fp4_pflops": pairs * W * 128 / (ms * 1e-3) / 1e15,
                          "total": got, "matches_column_identity": got == want}), flush=True)
        assert got == want
        m.close()


if __name__ == "__main__":
    main()
