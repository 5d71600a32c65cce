#!/usr/bin/env python3
"""Cold-start costs a real caller sees.  Every scenario runs in its own fresh process (the first pageable copy, the lazy code-object
load and the pinned ring are per-process costs), prints one JSON line, and the driver collects them.

  python3 tools/bench_cold.py                 # all scenarios
  python3 tools/bench_cold.py --one contig    # one scenario in this process (what the driver starts)
"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SCENARIOS = ["library", "contig", "storm:262144", "storm:52428", "storm:20971", "storm:524", "storm:104",
             "lists:524", "lists:2096", "lists:3145", "lists:3670"]


def one(what):
    t_start = time.perf_counter()
    import numpy as np
    import stormbitmaps_amd as sb
    rec = {"scenario": what}
    def lap(name, t0): rec[name] = round((time.perf_counter() - t0) * 1e3, 3)
    t0 = time.perf_counter(); sb.load(); lap("load_library_ms", t0)
    kind, _, arg = what.partition(":")
    if kind == "library":
        t0 = time.perf_counter(); ctx = sb.HipContext(0); lap("context_ms", t0)
        t0 = time.perf_counter(); m = ctx.matrix(1024, 1024); m.fill_synthetic(65536, 32768, seed=1); ctx.synchronize()
        lap("matrix_create_fill_1024_ms", t0)
        for k in range(3):
            t0 = time.perf_counter(); m.pairw(); lap(f"pairw_1024_call{k}_ms", t0)
        m.close()
    elif kind == "contig":
        c = sb.StormContig(65536)
        t0 = time.perf_counter(); c.add_synthetic(10000, 32768, seed=42); lap("add_synthetic_c2_ms", t0)
        time.sleep(0.05)
        for k in range(3):
            t0 = time.perf_counter(); c.pairw_intersect_cardinality_blocked(0); lap(f"call{k}_ms", t0)
        c.free()
    elif kind == "storm":
        s = sb.Storm()
        t0 = time.perf_counter(); s.add_synthetic(524288, 10000, int(arg), seed=42); lap("add_synthetic_c4_ms", t0)
        time.sleep(0.05)
        for k in range(3):
            t0 = time.perf_counter(); s.pairw_intersect_cardinality_blocked(0); lap(f"call{k}_ms", t0)
        s.free()
    elif kind == "lists":
        import torch                                  # only for an output buffer in device memory
        N = 10000
        dev = torch.zeros((N, N), dtype=torch.int32, device="cuda:0"); torch.cuda.synchronize()
        s = sb.Storm()
        s.add_synthetic(524288, N, int(arg), seed=42)
        for k in range(3):
            t0 = time.perf_counter(); s.pairw_matrix_device(dev.data_ptr(), N, N); lap(f"matrix_call{k}_ms", t0)
        assert int(dev.to(torch.int64).sum().item()) == s.pairw_intersect_cardinality()
        s.free()
    else:
        raise SystemExit(f"unknown scenario {what}")
    rec["process_total_s"] = round(time.perf_counter() - t_start, 2)
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2])
    else:
        for sc in (sys.argv[1:] or SCENARIOS):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", sc], capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(line[-1] if line else json.dumps({"scenario": sc, "error": r.stderr[-300:]}), flush=True)
