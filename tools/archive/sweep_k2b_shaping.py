#!/usr/bin/env python3
"""Work-list shaping of the strips re-swept for K2b at the headline shape (whole pass, same process)."""
import json, sys, time
sys.path.insert(0, "/root/repo")
import torch
import stormbitmaps_amd as sb

def bench(ctx, m, t, passes=150, warm_ms=30.0):
    stream = torch.cuda.current_stream()
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
    return a.elapsed_time(b) * 1e3 / passes

def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    ctx = sb.HipContext(0, torch.cuda.current_stream().cuda_stream)
    t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    m = ctx.matrix(rows, 1024)
    m.fill_synthetic(65536, 32768, seed=42)
    want = m.column_identity()
    base = {"k2_max_run": 0, "k2_tail_run": 32, "k2_tail_slices": 3, "k2_lpt_rounds": 6}
    trials = [dict(base)]
    for k, vals in (("k2_max_run", (64, 96, 160, 192, 256, 512)), ("k2_tail_run", (8, 16, 24, 48, 64)),
                    ("k2_tail_slices", (1, 2, 4, 6, 8)), ("k2_lpt_rounds", (0, 2, 12, 20))):
        for v in vals:
            d = dict(base); d[k] = v; trials.append(d)
    trials.append(dict(base))
    for d in trials:
        for k, v in d.items():
            ctx.set_option(k, v)
        us = bench(ctx, m, t)
        ok = int(t.item()) == want
        print(json.dumps({"rows": rows, **d, "us": round(us, 2), "ok": ok, "items": ctx.last_launch_info()["items"]}), flush=True)
    m.close(); ctx.close()
main()
