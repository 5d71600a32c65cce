#!/usr/bin/env python3
"""K2b against K2q and the FP4 strips over N (M = 65536), whole pass, same process; and 1/G shards of the headline shape."""
import json, sys, time
sys.path.insert(0, "/root/repo")
import torch
import stormbitmaps_amd as sb

def bench(ctx, rows, bits, opts, passes, rank=0, count=1, warm_ms=40.0):
    stream = torch.cuda.current_stream()
    for k, v in opts.items():
        ctx.set_option(k, v)
    W = (bits + 63) // 64
    t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    m = ctx.matrix(rows, W)
    m.fill_synthetic(bits, bits // 2, seed=42)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_ms * 1e-3:
        for _ in range(20):
            m.pairw_launch(t.data_ptr(), rank, count)
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(passes):
        m.pairw_launch(t.data_ptr(), rank, count)
    b.record(stream)
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / passes
    m.close()
    return us

def main():
    ctx = sb.HipContext(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_option("variant", 4)
    forms = {"fp4": {"k2_strip_operands": 4}, "k2q": {"k2_strip_operands": 2},
             "k2b": {"k2_strip_operands": 5, "k2_fold_inline": 0}, "k2b_inl": {"k2_strip_operands": 5, "k2_fold_inline": 1}}
    for pad in (1,):
        ctx.set_option("k2_matrix_pad", pad)
        for rows in (256, 512, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 10000, 12000):
            r = {"rows": rows, "pad": pad}
            for name, o in forms.items():
                if name == "k2b_inl" and rows > 4096:
                    continue
                r[name] = round(bench(ctx, rows, 65536, o, 300 if rows <= 4096 else 100), 2)
            print(json.dumps(r), flush=True)
        for G in (2, 4, 8):
            r = {"shard_of": 10000, "G": G}
            for name in ("fp4", "k2b"):
                r[name] = round(max(bench(ctx, 10000, 65536, forms[name], 200, rank, G) for rank in (0, G - 1)), 2)
            print(json.dumps(r), flush=True)
    ctx.close()

main()
