"""Round-3 GPU parity tests: the one-launch stage stream on bit operands (K2q, bitstream_kernel) that
matrices of up to 8192 rows take by default — against the CPU oracle where the CPU can afford it, the
column identity and the FP4 strips otherwise. Everything goes through the C-ABI."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import stormbitmaps_amd as sb
from stormbitmaps_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_ctx():
    ctx = sb.HipContext(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="module")
def orc():
    from tests._orc import Oracle
    return Oracle()


def _reset(ctx):
    for k, v in (("k2_strip_operands", 0), ("k2_stream_groups_per_cu", 0), ("k2_stream_min_piece", 6),
                 ("k2_stream_min_run", 2), ("k2_stream_max_rows", 8192), ("k2_stream_w3_1", 120),
                 ("k2_stream_w3_2", 60)):
        ctx.set_option(k, v)


def test_stream_kernel_matches_the_oracle(hip_ctx, orc):
    """K2q (option k2_strip_operands = 2; round 3's default up to 8192 rows, since round 4 K2b is the default at every size). Shapes around every edge of the decomposition: one block, one tile,
    2 / 3 / 4 / 5 tiles (the cyclic deal of tile pairs has an odd and an even form), ragged last blocks and
    tiles, one k-slice, a ragged last k-slice, rows of zero."""
    shapes = ((64, 2), (100, 3), (4096, 63), (4096, 64), (640, 65), (4096, 200), (4096, 256), (1000, 257),
              (8192, 511), (4160, 513), (9000, 700), (30000, 1000), (4096, 1100), (12345, 1500))
    try:
        hip_ctx.set_option("k2_strip_operands", 2)   # (round 3's default up to 8192 rows; since round 4 an option)
        for M, N in shapes:
            for d in (M // 2, max(1, M // 50)):
                mat = synth.dense_matrix_c(M, N, d, seed=N + M)
                mat[N // 2] = 0                                   # an empty row in the middle
                want = orc.wrapper_diag_blocked(mat, 31)
                m = hip_ctx.matrix_from_host(mat)
                got = [m.pairw() for _ in range(3)]               # (run-to-run: the race a skipped wait gives)
                assert got == [want] * 3, (M, N, d, got, want)
                assert hip_ctx.get_option("k2_operands_used") == 2 and hip_ctx.get_option("variant_used") == 4
                assert m.column_identity() == want
                # the stream can be sharded too (contiguous parts of its k-slice-major stage stream)
                assert sum(m.pairw(r, 3) for r in range(3)) == want
                assert hip_ctx.get_option("k2_operands_used") == 2
                for world in (2, 5):
                    assert sum(m.pairw(r, world) for r in range(world)) == want, (M, N, d, world)
                m.close()
    finally:
        _reset(hip_ctx)


@pytest.mark.parametrize("per_cu,min_piece,min_run,w1,w2",
                         [(0, 6, 2, 120, 60), (1, 1, 1, 120, 60), (2, 40, 9, 100, 100), (3, 6, 2, 300, 10),
                          (7, 6, 1, 120, 60), (16, 1, 3, 120, 60), (0, 6, 2, 10, 1000), (3, 1, 1, 1000, 1000)])
def test_stream_shaping_options_do_not_change_the_total(hip_ctx, per_cu, min_piece, min_run, w1, w2):
    """How the stage stream is cut into workgroups (shares per CU, shortest share, shortest run beside a cut,
    more shares than slots, the proportion of a CU's three shares) is tuning: every setting gives the same total. Shapes of 2..40 tiles, sizes at
    which segments are cut in the middle and continued by another workgroup."""
    try:
        hip_ctx.set_option("k2_strip_operands", 2)
        hip_ctx.set_option("k2_stream_groups_per_cu", per_cu)
        hip_ctx.set_option("k2_stream_min_piece", min_piece)
        hip_ctx.set_option("k2_stream_min_run", min_run)
        hip_ctx.set_option("k2_stream_w3_1", w1)
        hip_ctx.set_option("k2_stream_w3_2", w2)
        for M, N in ((65536, 1024), (20000, 2300), (9999, 777), (65536, 300), (2048, 5000), (512, 10000),
                     (65536, 2048)):
            m = hip_ctx.matrix(N, (M + 63) // 64)
            m.fill_synthetic(M, M // 3, seed=5)
            want = m.column_identity()
            assert m.pairw() == want, (M, N, per_cu, min_piece, min_run)
            assert sum(m.pairw(r, 3) for r in range(3)) == want
            m.close()
    finally:
        _reset(hip_ctx)


def test_stream_kernel_through_the_storm_h_containers(orc):
    """STORM_contiguous_t (storm.h) containers of mid-size take K2q on the way through
    STORM_contig_pairw_intersect_cardinality[_blocked]: same totals as the oracle's blocked loop
    (storm.c:1175-1241) and as the raw-buffer wrappers."""
    M, N, d = 65536, 1200, 20000
    mat = synth.dense_matrix_c(M, N, d, seed=77)
    want = orc.wrapper_diag_blocked(mat, 31)
    c = sb.StormContig(M)
    for r in synth.positions_from_dense(mat):
        c.add(r)
    assert c.pairw_intersect_cardinality_blocked(31) == want
    assert c.pairw_intersect_cardinality() == want
    assert sb.wrapper_diag_blocked(mat, 31) == want
    c.free()


def test_stream_kernel_at_full_mid_sizes_against_the_column_identity(hip_ctx):
    """The sizes tools/archive/midsize_pass.py reports (M = 65536, dense): totals against the size-independent
    identity sum_c C(n_c, 2), repeated (the last workgroup to arrive folds the partial sums and leaves
    slots and ticket zeroed for the next pass)."""
    try:
        hip_ctx.set_option("k2_strip_operands", 2)
        for N in (512, 1024, 2048, 4096, 8192):
            m = hip_ctx.matrix(N, 1024)
            m.fill_synthetic(65536, 32768, seed=42)
            want = m.column_identity()
            assert [m.pairw() for _ in range(5)] == [want] * 5, N
            assert hip_ctx.get_option("k2_operands_used") == 2
            m.close()
    finally:
        _reset(hip_ctx)


def test_block_columns_of_mixed_kinds_split_per_block(orc):
    """The reference dispatches per BLOCK PAIR on the kinds of the two blocks (storm.c:618-656). Here a block
    column's list blocks pair with each other in the list-probe kernel, and every pair with a bitmap block goes
    to the matrix cores (the column's bitmap rows as A rows, with the lists behind them): mostly short lists
    with a few bitmap blocks among them no longer cost a dense pass over the whole column. Rows of three
    densities in every order, several block columns, against the oracle's STORM_t restatement (intended
    semantics, defect D1 not reproduced) and the dense-everything setting."""
    import ctypes as C
    lib = sb.load()
    M = 3 * 65536 + 1000
    rng = np.random.default_rng(11)
    for n_rows, kinds in ((300, (40, 5000, 30000)), (700, (3, 3, 20000, 200)), (130, (30000, 2)), (513, (1, 60000, 700))):
        rows = []
        for r in range(n_rows):
            d = kinds[int(rng.integers(0, len(kinds)))]
            rows.append(np.unique(rng.integers(0, M, size=d, dtype=np.uint64)).astype(np.uint32))
        rows[n_rows // 3] = np.zeros(0, dtype=np.uint32)     # an empty row
        s = sb.Storm()
        for r in rows:
            s.add(r)
        want = orc.storm(rows).pairw_blocked(0)
        got = s.pairw_intersect_cardinality_blocked(0)
        assert got == want, (n_rows, kinds, got, want)
        data = s.serialize()
        ctx = sb.HipContext(0)
        h = C.c_void_p()
        assert lib.storm_hip_sparse_create_serialized(ctx._h, data.ctypes.data_as(C.c_void_p), data.size, C.byref(h)) == 0
        out = C.c_uint64()
        for probe in (-1, 0, 1):
            ctx.set_option("sparse_probe", probe)
            for world in (1, 3):
                tot = 0
                for r in range(world):
                    assert lib.storm_hip_pairw_sparse(ctx._h, h, r, world, C.byref(out)) == 0
                    tot += out.value
                assert tot == want, (n_rows, kinds, probe, world, tot, want)
        census = (C.c_uint64 * 4)()
        assert lib.storm_hip_sparse_last_census(ctx._h, C.byref(census)) == 0
        assert census[1] > 0            # list x bitmap block pairs exist: the columns are mixed
        lib.storm_hip_sparse_destroy(ctx._h, h)
        ctx.close()
        s.free()


def test_serialized_walker_applies_the_host_parser_s_rules():
    """storm_hip_sparse_create_serialized walks the headers of a serialized STORM_t itself: a stream that
    STORM_deserialize refuses (more rows than bytes, an unsorted or duplicated list, a set-bit count or block
    id that contradicts the header) is refused here too, before anything is allocated for it."""
    import ctypes as C
    import struct
    lib = sb.load()
    ctx = sb.HipContext(0)
    rows = [np.array([5, 9, 70000, 70001], dtype=np.uint32), np.array([1, 2, 3], dtype=np.uint32)]
    s = sb.Storm()
    for r in rows:
        s.add(r)
    data = s.serialize()
    s.free()
    at = 8 + 12 + 8 + 16
    unsorted = data.copy(); unsorted[at:at + 4] = np.frombuffer(struct.pack("<HH", 9, 5), dtype=np.uint8)
    duplicate = data.copy(); duplicate[at:at + 4] = np.frombuffer(struct.pack("<HH", 5, 5), dtype=np.uint8)
    wrong_count = data.copy(); wrong_count[8 + 12 + 8 + 4] = 7
    wrong_id = data.copy(); wrong_id[8 + 12 + 8 + 12] = 3
    huge = np.frombuffer(struct.pack("<IIII", 0xFFFFFFFF, 0x314D5453, 0, 0), dtype=np.uint8).copy()
    h = C.c_void_p()
    assert lib.storm_hip_sparse_create_serialized(ctx._h, data.ctypes.data_as(C.c_void_p), data.size, C.byref(h)) == 0
    out = C.c_uint64()
    assert lib.storm_hip_pairw_sparse(ctx._h, h, 0, 1, C.byref(out)) == 0 and out.value == 0
    lib.storm_hip_sparse_destroy(ctx._h, h)
    for bad in (unsorted, duplicate, wrong_count, wrong_id, huge):
        h = C.c_void_p()
        assert lib.storm_hip_sparse_create_serialized(ctx._h, bad.ctypes.data_as(C.c_void_p), bad.size, C.byref(h)) != 0
        assert not h.value
    ctx.close()


def _golden_total(M, N, d):
    import json
    import os
    root = os.path.dirname(os.path.abspath(__file__))
    for c in json.load(open(os.path.join(root, "golden", "synth_totals.json")))["dense"]:
        if (c["M"], c["N"], c["draws"]) == (M, N, d):
            return c["total"]
    raise KeyError((M, N, d))


def _benchmark_rows(args, timeout=600):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "stormbitmaps_amd", "storm_benchmark")
    res = subprocess.run([exe, *args], capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-2000:]
    # (RCCL prints a version banner on stdout when a communicator is created: keep the result rows)
    return [l.split("\t") for l in res.stdout.strip().splitlines()
            if l.count("\t") >= 15 and not l.startswith(("Samples", "#"))]


def test_native_rccl_reduce_for_c_callers_with_one_rank():
    """The C-side exchange (include/storm_hip.h storm_hip_comm_*, storm.h STORM_hip_comm_*): librccl is
    dlopen'ed, a communicator of ONE rank is created and the storm.h entry points return the all-reduced
    total — through tools/storm_benchmark.cpp --ranks 1, which forks its rank before any HIP call and hands
    the id over a pipe, exactly as it does for N ranks. Totals against the committed oracle vectors."""
    rows = _benchmark_rows(["4096", "256", "2048,40", "--ranks", "1", "--reps", "1", "--cpu-seconds", "0"])
    assert rows and all(int(r[-10]) == 1 for r in rows)          # GPUs column
    for load in (2048, 40):
        totals = {int(r[2]) for r in rows if int(r[1]) == load}
        assert totals == {_golden_total(4096, 256, load)}, (load, totals)


def test_native_rccl_reduce_over_two_gpus_when_the_box_has_them():
    """Two processes, one GPU each, shard partials all-reduced over RCCL by the C library itself
    (ncclAllReduce, count 1, ncclUint64, ncclSum). Skipped on 1-GPU boxes: RCCL refuses two ranks on one
    device."""
    n_dev = sb.load().storm_hip_device_count()
    if n_dev < 2:
        pytest.skip(f"{n_dev} GPU visible: RCCL needs one GPU per rank")
    rows = _benchmark_rows(["65536", "700", "32768,262", "--ranks", "2", "--reps", "2", "--cpu-seconds", "0"])
    assert rows and all(int(r[-10]) == 2 for r in rows)
    for load in (32768, 262):
        totals = {int(r[2]) for r in rows if int(r[1]) == load}
        assert totals == {_golden_total(65536, 700, load)}, (load, totals)


@pytest.mark.parametrize("M,N", [(65536, 700), (5000, 900), (1000, 300)])
def test_contig_add_builds_device_rows_from_positions(orc, M, N):
    """STORM_contig_add (storm.c:1031-1137) with the device copy of a row built from its POSITIONS
    (set_bits_kernel) when they are fewer bytes than its words: rows below the list cutoff (positions from
    `scalar`), between the cutoff and W (positions kept from the add), denser (words), mixed in one container;
    all-pairs calls in the middle of construction (streamed batches + the partial last one); per-pair matrix;
    the same totals as the oracle's blocked loop over the host bitmap (storm.c:1175-1241)."""
    rng = np.random.default_rng(M + N)
    W = (M + 63) // 64
    cutoff = min(200, M // 200)
    c = sb.StormContig(M)
    mat = np.zeros((N, W), dtype=np.uint64)
    for i in range(N):
        kind = int(rng.integers(0, 4))
        n = (int(rng.integers(1, max(2, cutoff))) if kind == 0 else
             int(rng.integers(cutoff, max(cutoff + 1, W))) if kind == 1 else
             int(rng.integers(W, 4 * W)) if kind == 2 else (W if rng.integers(0, 2) else max(1, cutoff)))
        pos = np.sort(rng.integers(0, M, size=max(1, n), dtype=np.uint32))   # duplicates kept, as callers send them
        assert c.add(pos) == pos.size
        for p in pos:
            mat[i, int(p) >> 6] |= np.uint64(1) << np.uint64(int(p) & 63)
        if i in (N // 3, N // 3 + 1, 2 * N // 3):
            assert c.pairw_intersect_cardinality() == orc.wrapper_diag_blocked(mat[: i + 1], 31), (M, N, i)
    want = orc.wrapper_diag_blocked(mat, 31)
    assert c.pairw_intersect_cardinality() == want
    assert c.pairw_intersect_cardinality_blocked(17) == want
    got = c.pairw_matrix("and")
    assert int(got.sum(dtype=np.uint64)) == want
    c.free()


def test_contig_add_positions_can_be_turned_off_and_give_the_same_totals():
    """STORM_HIP_ADD_POSITIONS=0 (rows always travel as words) in a child process: same total."""
    code = ("import numpy as np, stormbitmaps_amd as sb\n"
            "c = sb.StormContig(65536)\n"
            "c.add_synthetic(600, 655, seed=9)\n"
            "print(c.pairw_intersect_cardinality())\n")
    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, STORM_HIP_ADD_POSITIONS=flag, STORM_HIP_STREAM_ROWS="1")
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(int(r.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1] and outs[0] > 0


def test_stream_operand_forms_agree_and_the_shipped_library_refuses_the_tools_forms(hip_ctx):
    """k2_strip_operands: 2 (K2q, shipped) and 4 (FP4 strips, shipped) against the column identity; 1 (one item per
    workgroup) and 3 (K2w: a private ring per wave) exist in the tools build only — there they must give the same
    totals, here the option is refused."""
    from tests.conftest import shipped
    try:
        forms = shipped(hip_ctx, "k2_strip_operands", (2, 4, 1, 3))
        for M, N in ((65536, 1024), (5000, 777), (512, 300), (4096, 2300)):
            m = hip_ctx.matrix(N, (M + 63) // 64)
            m.fill_synthetic(M, M // 3, seed=11)
            want = m.column_identity()
            for ops in forms:
                hip_ctx.set_option("k2_strip_operands", ops)
                assert [m.pairw(), m.pairw()] == [want, want], (M, N, ops)
            m.close()
        if hip_ctx.get_option("probes_build") != 1:
            for ops in (1, 3):
                with pytest.raises(Exception):
                    hip_ctx.set_option("k2_strip_operands", ops)
    finally:
        _reset(hip_ctx)


def test_different_handles_from_different_threads():
    """storm.h: a handle is one thread's at a time, but different handles may be used from different threads — the
    device contexts behind them are shared and every pass takes one process-wide lock. Three threads (ctypes drops
    the GIL inside the calls): a STORM_t of 5000 rows (its arena fingerprint runs on helper threads from 4096 rows),
    a dense STORM_contiguous_t and the raw-buffer wrapper, 25 calls each, every total as in a quiet process."""
    import threading
    M = 65536
    s = sb.Storm()
    assert s.add_synthetic(M, 5000, 300, seed=3) == 5000
    c = sb.StormContig(M)
    assert c.add_synthetic(700, 20000, seed=4) == 700
    mat = synth.dense_matrix_c(M, 300, 9000, seed=5)
    want = (s.pairw_intersect_cardinality_blocked(0), c.pairw_intersect_cardinality(), sb.wrapper_diag_blocked(mat, 31))
    assert min(want) > 0
    bad = []

    def loop(fn, expect, name):
        for _ in range(25):
            got = fn()
            if got != expect:
                bad.append((name, got, expect))

    ts = [threading.Thread(target=loop, args=(lambda: s.pairw_intersect_cardinality_blocked(0), want[0], "storm_t")),
          threading.Thread(target=loop, args=(c.pairw_intersect_cardinality, want[1], "contig")),
          threading.Thread(target=loop, args=(lambda: sb.wrapper_diag_blocked(mat, 31), want[2], "wrapper"))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not bad, bad[:3]
    s.free()
    c.free()


def test_flat_array_arena_refuses_lists_that_are_not_strictly_ascending(hip_ctx):
    """storm_hip_sparse_create (the flat-array C-ABI): the list-probe kernel counts every listed element, so a list
    with a repeated or descending position must not become an arena (STORM_deserialize and the serialized walker
    refuse the same)."""
    lib = sb.load()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    for lists, ok in (([5, 9, 70, 5, 9, 71], True), ([5, 9, 9, 5, 9, 71], False), ([5, 9, 70, 9, 5, 71], False)):
        row_off = np.array([0, 1, 2], dtype=np.uint64)
        ids = np.array([0, 0], dtype=np.uint32)
        kinds = np.array([0, 0], dtype=np.uint8)
        offs = np.array([0, 3], dtype=np.uint64)
        lens = np.array([3, 3], dtype=np.uint32)
        a_lists = np.array(lists, dtype=np.uint16)
        words = np.zeros(1, dtype=np.uint64)
        h = C.c_void_p()
        rc = lib.storm_hip_sparse_create(hip_ctx._h, 2, 2, p(row_off), p(ids), p(kinds), p(offs), p(lens),
                                         p(a_lists), a_lists.size, p(words), 0, C.byref(h))
        assert (rc == 0) == ok, (lists, rc, lib.storm_hip_last_error())
        if rc == 0:
            out = C.c_uint64()
            assert lib.storm_hip_pairw_sparse(hip_ctx._h, h, 0, 1, C.byref(out)) == 0
            assert out.value == 2       # {5, 9} are shared
            lib.storm_hip_sparse_destroy(hip_ctx._h, h)
