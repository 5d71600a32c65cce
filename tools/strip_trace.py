#!/usr/bin/env python3
"""Schedule trace of the strip kernel at the headline shape (tuning aid): per-item start/end on
the device's 100 MHz counter + XCD, written as .npy under gpurun_out/ and summarised."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--bits", type=int, default=65536)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "strip_trace.npy"))
    args = ap.parse_args()
    import stormbitmaps_amd as sb
    ctx = sb.HipContext(0)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    m = ctx.matrix(args.rows, (args.bits + 63) // 64)
    m.fill_synthetic(args.bits, args.bits // 2, seed=42)
    want = m.column_identity()
    for _ in range(3):
        m.pairw()
    ctx.set_option("k2_ring", 18)
    assert m.pairw() == want
    n = C.c_uint64(0)
    lib = sb.load()
    assert lib.storm_hip_debug_strip_trace(ctx._h, None, 0, C.byref(n)) == 0
    out = np.zeros((n.value, 8), dtype=np.uint64)
    assert lib.storm_hip_debug_strip_trace(ctx._h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)) == 0
    np.save(args.out, out)
    t0 = out[:, 0].min()
    start = (out[:, 0] - t0).astype(np.float64) / 100.0   # us
    end = (out[:, 1] - t0).astype(np.float64) / 100.0
    xcc = out[:, 3] & 0xf
    stages = out[:, 6] + 4 * out[:, 5]
    print(f"items {n.value}  span {end.max():.1f} us")
    for x in range(8):
        sel = xcc == x
        if not sel.any():
            continue
        busy = (end[sel] - start[sel]).sum()
        print(f"xcc {x}: items {sel.sum():5d} first start {start[sel].min():7.1f} last end {end[sel].max():7.1f} "
              f"slot-us {busy:9.0f} stages {stages[sel].sum():7d} us/stage {busy / stages[sel].sum():.3f}")
    # concurrency over time (all XCDs)
    grid = np.linspace(0, end.max(), 41)
    for a, b in zip(grid[:-1], grid[1:]):
        mid = 0.5 * (a + b)
        print(f"t={mid:7.1f} us running {(np.sum((start <= mid) & (end > mid))):5d}")
    ph = out[:, 2]
    ready = (ph & 0xffff).astype(np.float64) / 100.0
    diag_done = ((ph >> 16) & 0xffff).astype(np.float64) / 100.0
    main_done = ((ph >> 32) & 0xffff).astype(np.float64) / 100.0
    dur = end - start
    has_diag = out[:, 5] == 1
    print(f"phases (us, mean): operands in {ready.mean():.2f} | diagonal phase {(diag_done - ready)[has_diag].mean():.2f} "
          f"(items with one) | main loop {(main_done - diag_done).mean():.2f} | epilogue {(dur - main_done).mean():.2f}")
    ms = (out[:, 6]).astype(np.float64)
    sel = ms >= 8
    print(f"main loop us/stage (items >= 8 stages): {((main_done - diag_done)[sel] / ms[sel]).mean():.3f}")
    per_stage = dur / np.maximum(stages, 1)
    for lo, hi in ((1, 8), (8, 32), (32, 64), (64, 128), (128, 400)):
        sel = (stages >= lo) & (stages < hi)
        if sel.any():
            print(f"stages [{lo},{hi}): n {sel.sum():5d} us/stage mean {per_stage[sel].mean():.3f} "
                  f"dur mean {dur[sel].mean():.1f}")
    m.close()


if __name__ == "__main__":
    main()
