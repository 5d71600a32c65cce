"""Per-pair output of a STORM_t (STORM_pairw_matrix): first call (dense replica built from the containers) and steady
calls at BASELINE c4's shape, with the all-pairs total of the same handle beside it as the check.
    python tools/bench_storm_matrix.py [--rows 10000] [--bits 524288] [--draws 524,20971,262144] > out.jsonl"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stormbitmaps_amd as sb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--bits", type=int, default=524288)
    ap.add_argument("--draws", default="524,20971,262144")
    a = ap.parse_args()
    for d in [int(x) for x in a.draws.split(",")]:
        s = sb.Storm()
        assert s.add_synthetic(a.bits, a.rows, d, seed=42) == a.rows
        out = np.zeros((a.rows, a.rows), dtype=np.uint32)
        lib, ptr = s._lib, out.ctypes.data
        t0 = time.perf_counter()
        assert lib.STORM_pairw_matrix(s._h, 0, ptr, a.rows, a.rows) == 0
        first = time.perf_counter() - t0
        steady = []
        for _ in range(3):
            t0 = time.perf_counter()
            assert lib.STORM_pairw_matrix(s._h, 0, ptr, a.rows, a.rows) == 0
            steady.append(time.perf_counter() - t0)
        total = s.pairw_intersect_cardinality()
        # the same into DEVICE memory (STORM_pairw_matrix_device, round 5), for both tile kernels
        import torch
        dev = torch.zeros((a.rows, a.rows), dtype=torch.int32, device="cuda:0")
        dev_ms = {}
        same = True
        lib = sb.load()
        # (matrix_lists 0: the dense replica under either tile kernel; -1: the automatic rule — the lists where they pay)
        for name, lists, shape in (("dense_tilebits8", 0, 2), ("dense_tilering", 0, 5), ("auto", -1, 0)):
            lib.STORM_hip_set_option(b"matrix_lists", lists)
            lib.STORM_hip_set_option(b"k2_tile_shape", shape)
            s.pairw_matrix_device(dev.data_ptr(), a.rows, a.rows)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                s.pairw_matrix_device(dev.data_ptr(), a.rows, a.rows)
                ts.append(time.perf_counter() - t0)
            dev_ms[name] = round(min(ts) * 1e3, 3)
            same = same and int(dev.to(torch.int64).sum().item()) == total
        lib.STORM_hip_set_option(b"k2_tile_shape", 0)
        lib.STORM_hip_set_option(b"matrix_lists", -1)
        import ctypes as C
        rep = (C.c_uint64 * 4)()
        lib.STORM_hip_last_pass(rep)
        print(json.dumps({"rows": a.rows, "bits": a.bits, "draws": d, "first_call_ms": round(first * 1e3, 2),
                          "steady_ms": round(min(steady) * 1e3, 2), "output_mb": out.nbytes / 1e6,
                          "device_output_ms": dev_ms, "auto_ran_lists": bool(rep[0] & 64), "auto_group_rows": int(rep[3]),
                          "device_output_sum_equals_total": same,
                          "sum_equals_all_pairs_total": int(out.sum(dtype=np.uint64)) == total}), flush=True)
        del dev
        s.free()


def contig():
    """STORM_contig_pairw_matrix into host memory at the headline shape."""
    N, M = 10000, 65536
    c = sb.StormContig(M)
    assert c.add_synthetic(N, M // 2, seed=42) == N
    out = np.zeros((N, N), dtype=np.uint32)
    lib, ptr = c._lib, out.ctypes.data
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        assert lib.STORM_contig_pairw_matrix(c._h, 0, ptr, N, N) == 0
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"entry": "STORM_contig_pairw_matrix", "rows": N, "bits": M, "first_call_ms": round(ts[0] * 1e3, 2),
                      "steady_ms": round(min(ts[1:]) * 1e3, 2), "output_mb": out.nbytes / 1e6,
                      "sum_equals_all_pairs_total": int(out.sum(dtype=np.uint64)) == c.pairw_intersect_cardinality()}), flush=True)
    c.free()


if __name__ == "__main__":
    contig()
    main()
