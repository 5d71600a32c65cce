#!/usr/bin/env python3
"""Schedule trace of bitstream_kernel (K2q; probes build): per-workgroup start / first operands / end on
the 100 MHz counter, stages, XCC and CU, summarised. STORM_HIP_LIB must point at libstorm_hip_probes.so."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--bits", type=int, default=65536)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    import stormbitmaps_amd as sb
    ctx = sb.HipContext(0)
    ctx.set_option("k2_strip_operands", 2)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    m = ctx.matrix(args.rows, (args.bits + 63) // 64)
    m.fill_synthetic(args.bits, args.bits // 2, seed=42)
    want = m.column_identity()
    import time
    t0 = time.perf_counter()   # clock ramp: the chip needs tens of milliseconds of back-to-back passes
    while time.perf_counter() - t0 < 0.06:
        m.pairw()
    ctx.set_option("k2_ring", 18)
    assert m.pairw() == want
    n = C.c_uint64(0)
    lib = sb.load()
    assert lib.storm_hip_debug_strip_trace(ctx._h, None, 0, C.byref(n)) == 0
    out = np.zeros((n.value, 8), dtype=np.uint64)
    assert lib.storm_hip_debug_strip_trace(ctx._h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)) == 0
    t0 = out[:, 0].min()
    start = (out[:, 0] - t0).astype(np.float64) / 100.0
    end = (out[:, 1] - t0).astype(np.float64) / 100.0
    ready = (out[:, 2] & 0xffffffff).astype(np.float64) / 100.0
    stages = (out[:, 2] >> 32).astype(np.int64)
    xcc = (out[:, 3] & 0xf).astype(np.int64)
    hw = (out[:, 3] >> 32).astype(np.int64)
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"rows {args.rows} groups {n.value} span {end.max():.2f} us  stages/group min {stages.min()} mean {stages.mean():.1f} max {stages.max()}")
    print(f"start: mean {start.mean():.2f} max {start.max():.2f} | first operands after start: mean {ready.mean():.2f} max {ready.max():.2f} | "
          f"end: min {end.min():.2f} mean {end.mean():.2f} max {end.max():.2f}")
    body = end - start - ready
    print(f"stage loop: mean {body.mean():.2f} us = {(body / np.maximum(stages, 1)).mean():.3f} us/stage")
    nm = np.maximum(out[:, 7].astype(np.float64), 1)
    st = np.maximum(stages.astype(np.float64), 1)
    print(f"wave 0 clocks per stage: barrier+vmcnt wait {(out[:, 4] / st).mean():.0f} | issue+cursor {(out[:, 5] / st).mean():.0f} | "
          f"body {(out[:, 6] / st).mean():.0f} (per multiplied stage {(out[:, 6] / nm).mean():.0f}; {nm.mean():.1f} of {st.mean():.1f} stages multiplied)")
    for x in range(8):
        sel = xcc == x
        print(f"  xcc {x}: end mean {end[sel].mean():.2f} max {end[sel].max():.2f}")
    # dispatch round = workgroup index / CUs; rank = order of arrival on its CU
    n_cus = len(np.unique(cuid))
    rnd = np.arange(n.value) // max(1, n_cus)
    rank = np.zeros(n.value, dtype=np.int64)
    for c in np.unique(cuid):
        idx = np.nonzero(cuid == c)[0]
        rank[idx[np.argsort(start[idx], kind="stable")]] = np.arange(len(idx))
    for r in range(int(rnd.max()) + 1):
        sel = rnd == r
        print(f"  dispatch round {r}: start mean {start[sel].mean():.2f}  end mean {end[sel].mean():.2f} max {end[sel].max():.2f}  "
              f"stages mean {stages[sel].mean():.1f}  arrival rank on the CU mean {rank[sel].mean():.2f}")
    per_cu = np.bincount(cuid)
    per_cu = per_cu[per_cu > 0]
    print(f"CUs used {len(per_cu)}  groups per CU: min {per_cu.min()} max {per_cu.max()}  histogram {np.bincount(per_cu).tolist()}")
    grid = np.linspace(0, end.max(), 21)
    print("running:", " ".join(f"{int(np.sum((start <= 0.5 * (a + b)) & (end > 0.5 * (a + b))))}" for a, b in zip(grid[:-1], grid[1:])))
    m.close()


if __name__ == "__main__":
    main()
