#!/usr/bin/env python3
"""Randomised parity soak on the GPU: random shapes / densities / options through every all-pairs
entry of the device library, each compared bit-for-bit with the CPU oracle (small shapes) or with
the column identity and the popcount kernel (large shapes). Prints one line per case and a final
summary; exit code 1 on the first mismatch.   soak_parity.py [--seconds 240] [--seed 1]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cases", type=int, default=100000)
    args = ap.parse_args()
    import stormbitmaps_amd as sb
    from stormbitmaps_amd import synth
    from tests._orc import Oracle
    orc = Oracle()
    ctx = sb.HipContext(0)
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    n_cases = 0
    kinds = {}
    while time.time() < t_end and n_cases < args.max_cases:
        kind = rng.choice(["dense_small", "dense_small", "dense_big", "square", "matrix", "matrix_big", "wrapper", "sparse", "storm_lists"])
        seed = int(rng.integers(1, 1 << 30))
        probes = ctx.get_option("probes_build") == 1   # the alternative kernel forms exist in the tools build only
        opts = {"variant": int(rng.choice([-1, -1, 4, 5 if probes else 4, 3, 2])),
                "k2_max_run": int(rng.choice([0, 0, 128, 1, 7, 64, 4096])),
                "k2_tail_slices": int(rng.choice([3, 0, 1, 8])),
                "k2_tail_run": int(rng.choice([32, 1, 5, 64])),
                "k2_persistent": int(rng.choice([0, 0, 1 if probes else 0])),
                "k2_pitch_pad": int(rng.choice([-1, -1, 0, 128, 640])),
                "k2_shape": int(rng.choice([16, 16, 32 if probes else 16])),
                "k2_tile_shape": int(rng.choice([0, 5, 5, 2, 2, 1, 16, 32] if probes else [0, 5, 5, 2, 2, 3, 4, 32, 6, 6, 6])),
                "k2_part_slots": int(rng.choice([0, 0, 1, 2])),
                "k2_part_min_chunks": int(rng.choice([8, 8, 1, 3, 40])),
                "k2_part_narrow": int(rng.choice([1, 1, 0])),
                "k2_wave_below": int(rng.choice([400, 400, 0, 100000])),
                "probe_bundle": int(rng.choice([-1, -1, 1, 4, 4])),
                "k2_ring_sync": int(rng.choice([0, 0, 1])),
                "k2_matrix_parts": int(rng.choice([0, 0, 1])),
                "k2_strip_operands": int(rng.choice([0, 0, 5, 4, 1 if probes else 2, 2, 6, 6])),
                "k2_matrix_pad": int(rng.choice([2, 2, 1, 0, 3])),
                "k2_fold_inline": int(rng.choice([-1, -1, 0, 1])),
                "k2_stream_groups_per_cu": int(rng.choice([0, 0, 1, 2, 3, 7])),
                "k2_stream_min_piece": int(rng.choice([6, 6, 1, 30])),
                "k2_stream_min_run": int(rng.choice([2, 2, 1, 9])),
                "k2_stream_w3_1": int(rng.choice([120, 120, 100, 300])),
                "k2_stream_w3_2": int(rng.choice([60, 60, 100, 15])),
                "k2_shadow_budget_mb": int(rng.choice([98304, 98304, 1, 8])),
                "sparse_probe": int(rng.choice([-1, -1, 0, 1]))}
        for k, v in opts.items():
            ctx.set_option(k, v)
        try:
            if kind in ("dense_small", "matrix", "square"):
                N = int(rng.integers(2, 1400))
                M = int(rng.choice([64, 100, 4096, 9000, 65536, int(rng.integers(65, 150000))]))
                d = int(max(1, M * rng.choice([0.5, 0.1, 0.01, 0.9]) * rng.random()))
                mat = synth.dense_matrix_c(M, N, d, seed=seed)
                if kind == "dense_small":
                    m = ctx.matrix_from_host(mat)
                    want = orc.wrapper_diag(mat)
                    world = int(rng.choice([1, 1, 2, 3, 8]))
                    got = sum(m.pairw(r, world) for r in range(world))
                    ok = got == want
                    m.close()
                elif kind == "matrix":
                    N = min(N, 600)
                    mat = mat[:N]
                    m = ctx.matrix_from_host(mat)
                    op = int(rng.integers(0, 3))
                    got_m = m.pairw_matrix(["and", "or", "xor"][op])
                    ok = np.array_equal(got_m, np.triu(orc.tile_counts_op(mat, 0, N, 0, N, op), k=1))
                    ok = ok and m.pairw_op(["and", "or", "xor"][op]) == int(got_m.sum(dtype=np.uint64))
                    m.close()
                else:
                    na = int(rng.integers(1, N))
                    a, b = mat[:na], mat[na:]
                    ma, mb = ctx.matrix_from_host(a), ctx.matrix_from_host(b)
                    ok = ma.square(mb) == orc.wrapper_square(a, b)
                    ma.close(); mb.close()
            elif kind == "matrix_big":   # more tiles than CUs now and then: k-split last rounds, ragged edges, rectangle, bands
                import torch
                N = int(rng.integers(600, 5200))
                M = int(rng.choice([640, 4096, 9000, 8192 + 512, 65536]))
                d = int(max(1, M * rng.choice([0.5, 0.1, 0.02]) * rng.random()))
                m = ctx.matrix(N, (M + 63) // 64)
                m.fill_synthetic(M, d, seed=seed)
                op = ["and", "or", "xor"][int(rng.integers(0, 3))]
                out = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
                m.pairw_matrix_device(out.data_ptr(), N, op)
                got = out.to(torch.int64)
                ok = int(torch.triu(got, diagonal=1).sum().item()) == m.pairw_op(op)
                # a random window against the oracle
                i0 = int(rng.integers(0, N - 40)); j0 = int(rng.integers(i0, N - 40))
                sub_rows = m.download(0, N)
                want = orc.tile_counts_op(sub_rows, i0, i0 + 40, j0, j0 + 40, ["and", "or", "xor"].index(op)).astype(np.int64)
                win = got[i0:i0 + 40, j0:j0 + 40].cpu().numpy()
                mask = (np.arange(i0, i0 + 40)[:, None] < np.arange(j0, j0 + 40)[None, :])
                ok = ok and np.array_equal(win[mask], want[mask])
                del out, got
                m.close()
            elif kind == "wrapper":      # the raw-buffer wrappers: rows streamed in panels on one device (storm.h context)
                N = int(rng.integers(2, 6000))
                M = int(rng.choice([64, 700, 4096, 20000]))
                d = int(max(1, M * rng.choice([0.5, 0.05]) * rng.random()))
                mat = synth.dense_matrix_c(M, N, d, seed=seed)
                want = orc.wrapper_diag_blocked(mat, 31) if N <= 2500 else None
                got = sb.wrapper_diag(mat)
                if want is None:
                    mm = ctx.matrix_from_host(mat)
                    want = mm.column_identity()
                    mm.close()
                ok = got == want == sb.wrapper_diag_blocked(mat, 5)
            elif kind == "dense_big":
                N = int(rng.integers(1500, 9000))
                M = int(rng.choice([4096, 20000, 65536, 131072]))
                d = int(max(1, M * rng.choice([0.5, 0.05]) * rng.random()))
                m = ctx.matrix(N, (M + 63) // 64)
                m.fill_synthetic(M, d, seed=seed)
                want = m.column_identity()
                got = m.pairw()
                ctx.set_option("variant", 2)
                ok = got == want and (N > 4000 or m.pairw() == want)
                m.close()
            elif kind == "storm_lists":   # K5: the per-pair matrix of a list-only STORM_t from its lists, both kernels
                import torch
                N = int(rng.integers(2, 900))
                M = int(rng.choice([8192, 65536, 200000, 524288, 1 << 22]))
                d = int(rng.choice([1, 5, 40, 150, 400, 1500]))
                d = min(d, M // 64)
                rows = synth.positions(M, N, d, seed=seed)
                if seed % 4 == 0:
                    rows[int(rng.integers(0, N))] = rows[0][:0]
                s = sb.Storm()
                for r in rows:
                    s.add(r)
                lib = sb.load()
                op = ["and", "or", "xor"][int(rng.integers(0, 3))]
                want = orc.storm(rows).pair_counts().astype(np.int64)
                lens = np.array([len(r) for r in rows], dtype=np.int64)
                want = {"and": want, "or": lens[:, None] + lens[None, :] - want, "xor": lens[:, None] + lens[None, :] - 2 * want}[op]
                ok = True
                for kernel in (1, 2):
                    lib.STORM_hip_set_option(b"matrix_lists", 1)
                    lib.STORM_hip_set_option(b"matrix_lists_kernel", kernel)
                    out = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
                    s.pairw_matrix_device(out.data_ptr(), N, N, op)
                    ok = ok and np.array_equal(np.triu(out.cpu().numpy().astype(np.int64), k=1), np.triu(want, k=1))
                lib.STORM_hip_set_option(b"matrix_lists", -1)
                lib.STORM_hip_set_option(b"matrix_lists_kernel", 0)
                ok = ok and int(np.triu(want, k=1).sum()) == s.pairw_intersect_cardinality() if op == "and" else ok
                s.free()
            else:  # sparse container through storm.h (host library has its own context: defaults)
                N = int(rng.integers(2, 700))
                M = int(rng.choice([65536, 200000, 524288]))
                d = int(rng.choice([1, 5, 60, 3000, 5000, 40000]))
                rows = synth.positions(M, N, d, seed=seed)
                s = sb.Storm()
                for r in rows:
                    s.add(r)
                # (mixed kinds in a third of the cases: every seventh row forty times as dense)
                if seed % 3 == 0:
                    rows = [synth.positions(M, 1, min(M // 2, d * 40), seed=seed + i)[0] if i % 7 == 0 else r for i, r in enumerate(rows)]
                    s.free()
                    s = sb.Storm()
                    for r in rows:
                        s.add(r)
                o = orc.storm(rows)
                want = o.pairw()
                ok = s.pairw_intersect_cardinality() == want == s.pairw_intersect_cardinality_blocked(0)
                # the per-pair matrix of the handle against the oracle's row-pair function (rows up to 2^25 bits)
                if ok and N <= 400:
                    ok = np.array_equal(s.pairw_matrix(), o.pair_counts())
                # the same container as a serialized stream on THIS context (random sparse_probe / shards)
                import ctypes as C
                data = s.serialize()
                h = C.c_void_p()
                lib = sb.load()
                ok = ok and lib.storm_hip_sparse_create_serialized(ctx._h, data.ctypes.data_as(C.c_void_p),
                                                                   data.size, C.byref(h)) == 0
                if ok:
                    world = int(rng.choice([1, 2, 5]))
                    out, got = C.c_uint64(), 0
                    for r in range(world):
                        ok = ok and lib.storm_hip_pairw_sparse(ctx._h, h, r, world, C.byref(out)) == 0
                        got += out.value
                    ok = ok and got == want
                    lib.storm_hip_sparse_destroy(ctx._h, h)
                s.free()
        finally:
            for k, v in {"variant": -1, "k2_max_run": 0, "k2_tail_slices": 3, "k2_tail_run": 32,
                         "k2_persistent": 0, "k2_pitch_pad": -1, "k2_matrix_pad": -1, "k2_shape": 16, "k2_tile_shape": 0, "k2_part_slots": 0, "k2_part_min_chunks": 8, "k2_part_narrow": 1, "k2_wave_below": 400, "probe_bundle": -1, "k2_ring_sync": 0, "k2_matrix_parts": 0, "k2_fold_inline": -1, "k2_strip_operands": 0,
                         "k2_stream_groups_per_cu": 0, "k2_stream_min_piece": 6, "k2_stream_min_run": 2,
                         "k2_stream_w3_1": 120, "k2_stream_w3_2": 60, "k2_shadow_budget_mb": 98304, "sparse_probe": -1}.items():
                ctx.set_option(k, v)
        n_cases += 1
        kinds[kind] = kinds.get(kind, 0) + 1
        if not ok:
            print(json.dumps({"FAILED": kind, "seed": seed, "opts": opts, "N": N, "M": M, "d": d}), flush=True)
            sys.exit(1)
        if n_cases % 25 == 0:
            print(f"{n_cases} cases ok {kinds}", flush=True)
    print(json.dumps({"cases": n_cases, "by_kind": kinds, "all_ok": True, "seed": args.seed}))


if __name__ == "__main__":
    main()
