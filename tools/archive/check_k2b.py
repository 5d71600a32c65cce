#!/usr/bin/env python3
"""K2b (strip16_bits_kernel, k2_strip_operands = 5) against the column identity, the FP4 strips and its own
shards on shapes around every edge of the strip decomposition; then wall time per pass at a few sizes against
the other operand forms (same box, same process)."""
import json
import sys
import time

sys.path.insert(0, "/root/repo")
import torch
import stormbitmaps_amd as sb


def check(ctx):
    bad = 0
    shapes = [(2, 64), (3, 512), (63, 4096), (64, 4096), (65, 640), (200, 4096), (256, 4096), (257, 1000),
              (300, 65536), (511, 8192), (512, 65536), (513, 4160), (700, 65536), (1000, 30000), (1024, 65536),
              (1100, 4096), (1500, 12345), (2048, 65536), (2300, 20000), (3000, 65536), (4096, 16384), (10000, 4096)]
    for N, M in shapes:
        W = (M + 63) // 64
        for draws in (M // 2, max(1, M // 50)):
            m = ctx.matrix(N, W)
            m.fill_synthetic(M, draws, seed=N + M)
            want = m.column_identity()
            for ring, fold in ((3, 1), (3, 0)):
                pass
                ctx.set_option("k2_fold_inline", fold)
                got = m.pairw()
                got2 = m.pairw()
                parts = [sum(m.pairw(r, G) for r in range(G)) for G in (2, 3, 5)]
                ok = got == want and got2 == want and all(p == want for p in parts)
                bad += not ok
                print(N, M, draws, ring, fold, "OK" if ok else f"FAIL got {got} {got2} parts {parts} want {want}",
                      ctx.get_option("k2_operands_used"), flush=True)
            m.close()
    print("BAD", bad, flush=True)
    return bad


def bench(ctx, rows, bits, opts, passes=200, warm_ms=60.0):
    stream = torch.cuda.current_stream()
    for k, v in opts.items():
        ctx.set_option(k, v)
    W = (bits + 63) // 64
    t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    m = ctx.matrix(rows, W)
    m.fill_synthetic(bits, bits // 2, seed=42)
    want = m.column_identity()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_ms * 1e-3:
        for _ in range(20):
            m.pairw_launch(t.data_ptr(), 0, 1)
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(passes):
        m.pairw_launch(t.data_ptr(), 0, 1)
    b.record(stream)
    torch.cuda.synchronize()
    ok = int(t.item()) == want
    us = a.elapsed_time(b) * 1e3 / passes
    flop = rows * (rows - 1) // 2 * W * 128
    print(json.dumps({"rows": rows, "bits": bits, "opts": opts, "us_per_pass": round(us, 2), "ok": ok,
                      "fp4_frac_whole_pass": round(flop / (us * 1e-6) / 1e16, 4),
                      "operands_used": ctx.get_option("k2_operands_used")}), flush=True)
    m.close()


def main():
    stream = torch.cuda.current_stream()
    ctx = sb.HipContext(0, stream.cuda_stream)
    ctx.set_option("variant", 4)
    ctx.set_option("k2_strip_operands", 5)
    bad = 0
    if "--no-check" not in sys.argv:
        bad = check(ctx)
    for rows, bits in ((10000, 65536), (2048, 65536), (10000, 524288)):
        passes = 200 if bits <= 65536 else 30
        for opts in ({"k2_strip_operands": 4, "k2_matrix_pad": 0}, {"k2_strip_operands": 2, "k2_matrix_pad": 0},
                     {"k2_strip_operands": 5, "k2_fold_inline": 0, "k2_matrix_pad": 0},
                     {"k2_strip_operands": 5, "k2_fold_inline": 0, "k2_matrix_pad": 1},
                     {"k2_strip_operands": 2, "k2_fold_inline": 0, "k2_matrix_pad": 1},
                     {"k2_strip_operands": 4, "k2_matrix_pad": 0}, {"k2_strip_operands": 5, "k2_fold_inline": 0, "k2_matrix_pad": 0},
                     {"k2_strip_operands": 5, "k2_fold_inline": 0, "k2_matrix_pad": 1}):
            bench(ctx, rows, bits, opts, passes)
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
