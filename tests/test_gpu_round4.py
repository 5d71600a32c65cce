"""Round-4 GPU parity tests: K2b (strip16_bits_kernel) — the 16x16x128 strips on BIT operands, the FP4 image of
every B stage built in the LDS by the workgroup — is the default all-pairs path at every size and for every
shard. Against the CPU oracle where the CPU can afford it (incl. the headline configuration, pair by pair),
the column identity, the other operand forms and its own shards. Everything goes through the C-ABI."""
import numpy as np
import pytest

import stormbitmaps_amd as sb
from stormbitmaps_amd import synth

pytestmark = pytest.mark.gpu


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
    for k, v in (("k2_strip_operands", 0), ("k2_fold_inline", -1), ("k2_matrix_pad", -1), ("variant", -1)):
        ctx.set_option(k, v)


def test_bit_operand_strips_are_the_default_and_match_the_oracle(hip_ctx, orc):
    """Default path: K2b on one device and for every shard. Shapes around every edge of the strip
    decomposition: one block, one tile (only the diagonal phase runs), 2..6 tiles, ragged last blocks and
    tiles, one k-slice (one class pair of one chunk holds all the data), ragged last chunks (n_words not a
    multiple of 8: the second class pair of the last chunk multiplies zero padding), rows of zero, item runs
    of 1, 2 and 3 stages (shorter than the pipeline's look-ahead: the pieces issued beyond the last stage are
    never consumed)."""
    shapes = ((64, 2), (100, 3), (4096, 63), (4096, 64), (640, 65), (4096, 200), (4096, 256), (1000, 257),
              (8192, 511), (4160, 513), (9000, 700), (30000, 1000), (4096, 1100), (12345, 1500), (300, 320),
              (256, 384), (200, 448))
    try:
        for M, N in shapes:
            for d in (M // 2, max(1, M // 50)):
                mat = synth.dense_matrix_c(M, N, d, seed=N + M)
                mat[N // 2] = 0                                   # an empty row in the middle
                want = orc.wrapper_diag_blocked(mat, 31)
                m = hip_ctx.matrix_from_host(mat)
                got = [m.pairw() for _ in range(3)]               # (run to run: the race a skipped wait gives)
                assert got == [want] * 3, (M, N, d, got, want)
                assert hip_ctx.get_option("k2_operands_used") == 5 and hip_ctx.get_option("variant_used") == 4
                assert m.column_identity() == want
                for world in (2, 3, 5):
                    assert sum(m.pairw(r, world) for r in range(world)) == want, (M, N, d, world)
                assert hip_ctx.get_option("k2_operands_used") == 5
                for other in (2, 4):                              # the stage stream and the FP4 strips agree
                    hip_ctx.set_option("k2_strip_operands", other)
                    assert m.pairw() == want, (M, N, d, other)
                    assert hip_ctx.get_option("k2_operands_used") == other
                hip_ctx.set_option("k2_strip_operands", 0)
                m.close()
    finally:
        _reset(hip_ctx)


def test_bit_operand_strips_with_and_without_the_pitch_pad_and_with_the_in_kernel_fold(hip_ctx):
    """k2_matrix_pad (rows that are a multiple of 1 KiB get that many 512-byte chunks more of pitch: tuning) and k2_fold_inline
    (the last workgroup to arrive folds the partial sums and leaves slots and ticket zeroed; slower, kept as an
    option) do not change the total; repeated passes see clean slots."""
    try:
        for pad in (0, 1, 2, 3, -1):
            hip_ctx.set_option("k2_matrix_pad", pad)
            for M, N in ((65536, 1024), (8192, 3000), (20000, 2300), (1024, 5000), (524288, 300)):
                m = hip_ctx.matrix(N, (M + 63) // 64)
                dense = ((M + 63) // 64 + 63) // 64 * 64                  # words, padded to whole 512-byte chunks
                chunks = pad if pad >= 0 else (4 if dense >= 8192 else 1)   # -1 (default): by the pitch
                assert m.stride_words == dense + (64 * chunks if dense % 128 == 0 else 0)
                m.fill_synthetic(M, M // 3, seed=9)
                want = m.column_identity()
                for fold in (0, 1, 0):
                    hip_ctx.set_option("k2_fold_inline", fold)
                    assert [m.pairw() for _ in range(3)] == [want] * 3, (M, N, pad, fold)
                    assert sum(m.pairw(r, 4) for r in range(4)) == want
                m.close()
    finally:
        _reset(hip_ctx)


def test_headline_configuration_against_the_oracle_pair_by_pair(hip_ctx, orc):
    """BASELINE configs[1] at full size (N = 10000, M = 65536, 32768 draws per row, seed 42 — bench.py's
    workload): the default path's total against the CPU oracle's blocked loop over ALL 49 995 000 pairs (the
    restatement of storm.c:1175-1241 with the harness's block size, benchmark.cpp:823-824; ~3 s with the
    AVX-512 leaf, ~25 s with the scalar one), the column identity, 8-way shards and the other two operand
    forms."""
    N, M = 10000, 65536
    m = hip_ctx.matrix(N, M // 64)
    try:
        m.fill_synthetic(M, M // 2, seed=42)
        got = m.pairw()
        assert hip_ctx.get_option("k2_operands_used") == 5
        mat = m.download()
        want = orc.wrapper_diag_blocked(mat, max(5, 256000 // (M // 64 * 8)))
        assert got == want == m.column_identity()
        assert sum(m.pairw(r, 8) for r in range(8)) == want
        for other in (2, 4):
            hip_ctx.set_option("k2_strip_operands", other)
            assert m.pairw() == want
    finally:
        _reset(hip_ctx)
        m.close()


def test_rows_of_half_a_million_bits_and_more(hip_ctx):
    """c3's shape (M = 524288; fewer rows) and a row length whose pitch x 64 rows is close to the strips' 32-bit
    DMA offsets: K2b, against the column identity and the FP4 strips (k-chunked there)."""
    try:
        for M, N in ((524288, 1500), (1 << 24, 300)):
            m = hip_ctx.matrix(N, M // 64)
            m.fill_synthetic(M, M // 4, seed=11)
            want = m.column_identity()
            assert m.pairw() == want and hip_ctx.get_option("k2_operands_used") == 5
            assert sum(m.pairw(r, 3) for r in range(3)) == want
            hip_ctx.set_option("k2_strip_operands", 4)
            assert m.pairw() == want
            hip_ctx.set_option("k2_strip_operands", 0)
            m.close()
    finally:
        _reset(hip_ctx)


def test_default_path_through_the_storm_h_containers(orc):
    """STORM_contiguous_t containers (storm.h) reach K2b through STORM_contig_pairw_intersect_cardinality[_blocked]:
    same totals as the oracle's blocked loop (storm.c:1175-1241) and as the raw-buffer wrappers."""
    M, N, d = 65536, 1200, 20000
    mat = synth.dense_matrix_c(M, N, d, seed=78)
    want = orc.wrapper_diag_blocked(mat, 31)
    c = sb.StormContig(M)
    for r in synth.positions_from_dense(mat):
        c.add(r)
    assert c.pairw_intersect_cardinality_blocked(31) == want
    assert c.pairw_intersect_cardinality() == want
    assert sb.wrapper_diag_blocked(mat, 31) == want
    c.free()


def test_first_call_on_a_fresh_sparse_container_equals_the_steady_calls(orc):
    """The reference's harness times ONE call right after construction (benchmark.cpp:605-613). Here that call
    builds the device arena — since round 4 from POINTERS to the blocks where they lie in the containers
    (storm_hip_sparse_create_blocks: no host flattening; raw lists and bitmaps through a pinned ring; element
    layout by probe_fill_kernel / probe_deal_kernel on the device). First-call total == steady totals == the
    oracle's STORM_t restatement, on lists only, bitmaps only and mixed kinds, incl. empty rows, one-element
    rows, a list that fills an octant and row counts that are not multiples of 128."""
    rng = np.random.default_rng(11)
    M = 3 * 65536 + 777
    for n_rows, draws in ((300, 40), (1000, 700), (517, 3000), (260, 20000), (700, 9000)):
        rows = []
        for r in range(n_rows):
            d = draws if r % 7 else (0 if r % 14 == 0 else 1)
            if draws == 9000 and r % 3 == 0:
                d = 60000                                            # mixed kinds: some blocks become bitmaps
            v = np.unique(rng.integers(0, M, size=d, dtype=np.uint64)).astype(np.uint32)
            if r == 5:
                v = np.unique(np.concatenate([v, np.arange(8192, 16384, 2, dtype=np.uint32)]))   # a full octant
            rows.append(v)
        want = orc.storm(rows).pairw_blocked(0)
        s = sb.Storm()
        for v in rows:
            s.add(v)
        first = s.pairw_intersect_cardinality_blocked(0)
        again = [s.pairw_intersect_cardinality_blocked(0), s.pairw_intersect_cardinality()]
        assert first == want and again == [want, want], (n_rows, draws, first, again, want)
        s.free()


def test_caller_threads_on_their_own_device_slots_run_side_by_side(orc):
    """STORM_hip_set_thread_devices (storm.h): two device slots configured (the one card of the box twice), two
    caller threads each narrowed to ITS slot — one lock per slot, so neither waits for the other — each with a
    STORM_t and a STORM_contiguous_t of its own, against the oracle; then one thread widened to both slots again."""
    import ctypes as C
    import threading
    lib = sb._lib.load()
    M = 2 * 65536
    data = []
    for t in range(2):
        rows = synth.positions(M, 500, 300 + 2500 * t, seed=90 + t)
        mat = synth.dense_matrix_c(M, 700, 9000, seed=95 + t)
        data.append((rows, orc.storm(rows).pairw_blocked(0), mat, orc.wrapper_diag_blocked(mat, 0)))
    ids = (C.c_int * 2)(0, 0)
    bad = []

    def work(t):
        try:
            assert lib.STORM_hip_set_thread_devices(t, 1) == 0
            rows, want_s, mat, want_c = data[t]
            s, c = sb.Storm(), sb.StormContig(M)
            for r in rows:
                s.add(r)
            for r in synth.positions_from_dense(mat):
                c.add(r)
            for _ in range(6):
                assert s.pairw_intersect_cardinality() == want_s
                assert c.pairw_intersect_cardinality() == want_c
            assert lib.STORM_hip_set_thread_devices(0, 0) == 0       # both slots: the mirrors follow the view
            assert c.pairw_intersect_cardinality() == want_c and s.pairw_intersect_cardinality() == want_s
            s.free()
            c.free()
        except BaseException as e:     # noqa: BLE001 — handed to the main thread
            bad.append((t, repr(e)))

    try:
        assert lib.STORM_hip_set_devices(2, ids) == 0
        assert lib.STORM_hip_set_thread_devices(1, 2) == -1
        th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not bad, bad
    finally:
        one_dev = (C.c_int * 1)(0)
        assert lib.STORM_hip_set_devices(1, one_dev) == 0


def test_per_pair_matrix_of_a_sparse_container_against_the_row_pair_function(orc):
    """STORM_pairw_matrix (storm.h extension; the LD use case of README.md:165-167 on STORM_t): entry (i, j) is what
    STORM_bitmap_cont_intersect_cardinality returns for rows i, j (storm.c:790-814 — oracle: orc_storm_pair_counts),
    on list-only rows, bitmap rows, mixed kinds, empty and one-element rows; union / xor through the rows' own
    cardinalities; its sum is the all-pairs total of the same handle; the dense replica follows STORM_add."""
    rng = np.random.default_rng(17)
    M = 5 * 65536 + 300
    for n_rows, draws in ((130, 60), (300, 4000), (257, 30000), (200, None)):
        rows = []
        for r in range(n_rows):
            d = draws if draws is not None else (20, 2500, 50000, 120000)[r % 4]
            d = d if r % 9 else (0 if r % 18 == 0 else 1)
            rows.append(np.unique(rng.integers(0, M, size=d, dtype=np.uint64)).astype(np.uint32))
        want = orc.storm(rows).pair_counts()
        s = sb.Storm()
        for v in rows:
            s.add(v)
        assert s.n_rows == n_rows
        got = s.pairw_matrix()
        assert np.array_equal(got, want), (n_rows, draws)
        assert int(got.sum(dtype=np.uint64)) == s.pairw_intersect_cardinality()
        card = np.array([len(v) for v in rows], dtype=np.int64)
        union = np.triu(card[:, None] + card[None, :], k=1) - want
        assert np.array_equal(s.pairw_matrix("or"), union)
        assert np.array_equal(s.pairw_matrix("xor"), union - want)
        extra = np.unique(rng.integers(0, M, size=700, dtype=np.uint64)).astype(np.uint32)
        s.add(extra)                                                  # the replica is rebuilt for the new row
        rows.append(extra)
        assert np.array_equal(s.pairw_matrix(), orc.storm(rows).pair_counts())
        s.free()
    # three device slots (the one card three times): each writes one band of rows of the same triangle
    import ctypes as C
    lib = sb._lib.load()
    rows = [np.unique(rng.integers(0, M, size=(40, 3000, 70000)[r % 3], dtype=np.uint64)).astype(np.uint32) for r in range(700)]
    want = orc.storm(rows).pair_counts()
    s = sb.Storm()
    for v in rows:
        s.add(v)
    try:
        assert lib.STORM_hip_set_devices(3, (C.c_int * 3)(0, 0, 0)) == 0
        assert np.array_equal(s.pairw_matrix(), want)
        assert s.pairw_intersect_cardinality() == int(want.sum(dtype=np.uint64))
    finally:
        assert lib.STORM_hip_set_devices(1, (C.c_int * 1)(0)) == 0
    assert np.array_equal(s.pairw_matrix("and"), want)                # rebuilt for the one-device configuration
    s.free()
    # rows beyond 2^25 bits: refused with the reason, nothing written
    wide = sb.Storm()
    wide.add(np.array([5, (1 << 25) + 3], dtype=np.uint32))
    wide.add(np.array([5], dtype=np.uint32))
    # (the dense replica's limit. [r5] a list-only container does not need the replica: K5 writes the matrix from the lists)
    assert lib.STORM_hip_set_option(b"matrix_lists", 0) == 0
    try:
        with pytest.raises(RuntimeError, match="2\\^25"):
            wide.pairw_matrix()
    finally:
        assert lib.STORM_hip_set_option(b"matrix_lists", -1) == 0
    assert np.array_equal(wide.pairw_matrix(), np.array([[0, 1], [0, 0]], dtype=np.uint32))
    assert wide.pairw_intersect_cardinality() == 1
    wide.free()
