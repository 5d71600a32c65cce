#!/usr/bin/env python3
"""K5 (lists_matrix_kernel, storm_hip_lists.hip): the per-pair matrix of a list-only STORM_t from its lists, against the
oracle's per-pair function and against the dense replica's tile kernels; then time per call at BASELINE c4's shape over
the sparse loads.   check_lists_matrix.py [--quick] [--draws 104,524,2621,5242,10485]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import stormbitmaps_amd as sb  # noqa: E402
from stormbitmaps_amd import synth  # noqa: E402
from tests._orc import Oracle  # noqa: E402


def opt(key, value):
    assert sb.load().STORM_hip_set_option(key.encode(), value) == 0


def device_matrix(s, n, op="and"):
    dev = torch.full((n, n), -7, dtype=torch.int32, device="cuda:0")
    s.pairw_matrix_device(dev.data_ptr(), n, n, op)
    return dev.cpu().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--draws", default="5,104,524,1048,2621,5242")
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--bits", type=int, default=524288)
    a = ap.parse_args()
    orc = Oracle()
    bad = 0
    for M, N, d in ((65536, 2, 5), (65536, 65, 40), (200000, 300, 60), (524288, 257, 524), (524288, 700, 100), (1 << 22, 130, 300),
                    (8192, 64, 100), (8193, 321, 7), (70000, 1000, 1)):
        rows = synth.positions(M, N, d, seed=N + d)
        rows[N // 3] = rows[N // 3][:0]          # an empty row
        s = sb.Storm()
        for r in rows:
            s.add(r)
        want = orc.storm(rows).pair_counts().astype(np.int64)
        lens = np.array([len(r) for r in rows], dtype=np.int64)
        upper = np.triu(np.ones((N, N), dtype=bool), k=1)
        for op in ("and", "or", "xor"):
            ref = {"and": want, "or": lens[:, None] + lens[None, :] - want, "xor": lens[:, None] + lens[None, :] - 2 * want}[op]
            opt("matrix_lists", 1)
            ok = True
            for kern in (1, 2):   # window kernel, hash kernel
                opt("matrix_lists_kernel", kern)
                got = device_matrix(s, N, op)
                ok = ok and np.array_equal(got[upper], ref[upper]) and bool((got[~upper] == -7).all()) and _ran_lists()
            opt("matrix_lists_kernel", 0)
            opt("matrix_lists", 0)
            dense = device_matrix(s, N, op)
            ok = ok and np.array_equal(dense[upper], ref[upper])
            bad += not ok
            print(M, N, d, op, "OK" if ok else f"FAIL {int((got[upper] != ref[upper]).sum())} entries, untouched {bool((got[~upper] == -7).all())}", flush=True)
        s.free()
    opt("matrix_lists", -1)
    print("BAD", bad, flush=True)
    if a.quick:
        sys.exit(1 if bad else 0)
    N, M = a.rows, a.bits
    dev = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
    for d in [int(x) for x in a.draws.split(",")]:
        s = sb.Storm()
        assert s.add_synthetic(M, N, d, seed=42) == N
        total = s.pairw_intersect_cardinality()
        rec = {"rows": N, "bits": M, "draws": d}
        for mode, kern, name in ((1, 1, "windows_ms"), (1, 2, "hash_ms"), (0, 0, "dense_ms")):
            opt("matrix_lists", mode)
            opt("matrix_lists_kernel", kern)
            dev.zero_()
            t0 = time.perf_counter()
            s.pairw_matrix_device(dev.data_ptr(), N, N)
            rec[name.replace("_ms", "_first_ms")] = round((time.perf_counter() - t0) * 1e3, 2)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                s.pairw_matrix_device(dev.data_ptr(), N, N)
                ts.append(time.perf_counter() - t0)
            rec[name] = round(min(ts) * 1e3, 3)
            rec[name.replace("_ms", "_sum_ok")] = int(dev.to(torch.int64).sum().item()) == total
        opt("matrix_lists", -1)
        opt("matrix_lists_kernel", 0)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            s.pairw_matrix_device(dev.data_ptr(), N, N)
            ts.append(time.perf_counter() - t0)
        rec["auto_ms"] = round(min(ts) * 1e3, 3)
        rec["auto_took_lists"] = _ran_lists()
        rec["auto_group_rows"] = _report()[3]
        print(json.dumps(rec), flush=True)
        s.free()
    sys.exit(1 if bad else 0)


def _report():
    import ctypes as C
    out = (C.c_uint64 * 4)()
    sb.load().STORM_hip_last_pass(out)
    return [int(x) for x in out]


def _ran_lists():
    return bool(_report()[0] & 64)


if __name__ == "__main__":
    main()
