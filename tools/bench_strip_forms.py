#!/usr/bin/env python3
"""K2b with 256-row tiles (strip16_bits_kernel, k2_strip_operands 5) against 512-row tiles (strip16_bits2_kernel, 6): kernel
time by HIP events over series of back-to-back passes (best of 5 series) and one synchronous call by the host's clock.
bench_strip_forms.py [rows:bits,...] [--opt key=value ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stormbitmaps_amd as sb
ctx = sb.HipContext(0)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = [a for a in sys.argv[1:] if "=" in a and a.startswith("--opt=")]
for o in opts:
    k, v = o[6:].split("="); ctx.set_option(k, int(v))
for shape in (args[0] if args else "10000:65536,10000:524288,2048:65536,4096:65536,8192:65536").split(","):
    N, M = (int(v) for v in shape.split(":"))
    m = ctx.matrix(N, M // 64)
    m.fill_synthetic(M, M // 2, seed=42)
    want = m.column_identity()
    rec = {"rows": N, "bits": M}
    n = 300 if N * M < 2e9 else 60
    for form in (5, 6, 5, 6):
        ctx.set_option("k2_strip_operands", form)
        assert m.pairw() == want
        for _ in range(n // 3): m.pairw()
        best, wall = 1e9, 1e9
        for rep in range(5):
            ctx.set_option("time_kernels", 1)
            t0 = time.perf_counter()
            for _ in range(n): m.pairw()
            ctx.synchronize()
            wall = min(wall, (time.perf_counter() - t0) / n * 1e3)
            ms, k = ctx.kernel_time()
            best = min(best, ms / k)
        ctx.set_option("time_kernels", 0)
        key = f"form{form}_used{ctx.get_option('k2_operands_used')}"
        rec.setdefault(key + "_kernel_ms", []).append(round(best, 5))
        rec.setdefault(key + "_pass_ms", []).append(round(wall, 5))
        rec[key + "_items"] = ctx.last_launch_info()["items"]
    ctx.set_option("k2_strip_operands", 0)
    print(json.dumps(rec), flush=True)
    m.close()
