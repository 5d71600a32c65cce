#!/usr/bin/env python3
"""c4 (STORM_t, N = 10000 x M = 524288): the list-probe kernel (K4, option sparse_probe = 1) against
the dense-on-present-blocks path (sparse_probe = 0) per load, on an arena built from the serialized
container. One JSON line per load; totals must agree."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import stormbitmaps_amd as sb
    lib = sb.load()
    M, N = 524288, 10000
    loads = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "5,104,524,1048,2097,3145,5242").split(",")]
    for d in loads:
        s = sb.Storm()
        assert s.add_synthetic(M, N, d, seed=42) == N
        data = s.serialize()
        s.free()
        ctx = sb.HipContext(0)
        if os.environ.get("STORM_PROBE_BUNDLE"):   # 1: probe_lists_kernel, 4: probe_lists_fat_kernel [r6]
            ctx.set_option("probe_bundle", int(os.environ["STORM_PROBE_BUNDLE"]))
        h = C.c_void_p()
        assert lib.storm_hip_sparse_create_serialized(ctx._h, data.ctypes.data_as(C.c_void_p), data.size, C.byref(h)) == 0
        out = C.c_uint64()
        row = {"load": d, "mean_list_len": d * 65536 // M}
        for probe, name in ((0, "dense_ms"), (1, "probe_ms")):
            ctx.set_option("sparse_probe", probe)
            for _ in range(3):
                assert lib.storm_hip_pairw_sparse(ctx._h, h, 0, 1, C.byref(out)) == 0
            ts = []
            for _ in range(8):
                t0 = time.perf_counter()
                assert lib.storm_hip_pairw_sparse(ctx._h, h, 0, 1, C.byref(out)) == 0
                ts.append(time.perf_counter() - t0)
            row[name] = round(min(ts) * 1e3, 3)
            row[name.replace("_ms", "_total")] = out.value
        assert row["dense_total"] == row["probe_total"], row
        row["speedup"] = round(row["dense_ms"] / row["probe_ms"], 2)
        print(json.dumps(row), flush=True)
        lib.storm_hip_sparse_destroy(ctx._h, h)
        ctx.close()


if __name__ == "__main__":
    main()
