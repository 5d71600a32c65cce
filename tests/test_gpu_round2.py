"""GPU parity tests added in round 2 (run on the MI355X box: pytest -m gpu).

Same rules as tests/test_gpu_parity.py: the product is called through libstorm_hip.so (C-ABI,
via the ctypes mirror of storm.h), the CPU oracle is the checker only.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import stormbitmaps_amd as sb
from stormbitmaps_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
FAILED = (1 << 64) - 1


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _list_inputs(M, draws_per_row, seed):
    """Rows of mixed density with the caller-side list arrays of STORM_wrapper_diag_list
    (storm.h:95-148): n_alts[N], alt_positions (all rows' sorted set bits), alt_offsets[N]."""
    rows = []
    for i, d in enumerate(draws_per_row):
        rows.append(synth.positions(M, 1, d, seed=seed + i)[0])
    W = (M + 63) // 64
    mat = np.zeros((len(rows), W), dtype=np.uint64)
    for i, r in enumerate(rows):
        r64 = r.astype(np.uint64)
        np.bitwise_or.at(mat[i], (r64 >> np.uint64(6)).astype(np.int64), np.uint64(1) << (r64 & np.uint64(63)))
    n_alts = np.array([len(r) for r in rows], dtype=np.uint32)
    offs = np.zeros(len(rows), dtype=np.uint32)
    offs[1:] = np.cumsum(n_alts)[:-1]
    pos = np.concatenate(rows).astype(np.uint32)
    return mat, n_alts, pos, offs


@pytest.mark.parametrize("M,N", [(4096, 300), (65536, 700), (9000, 131)])
def test_wrapper_diag_list_and_list_blocked_against_oracle(lib, orc, M, N):
    """STORM_wrapper_diag_list / _diag_list_blocked (reference storm.c:190-219, :282-369) through
    libstorm_hip.so with real n_alts / alt_positions / alt_offsets, against the oracle's
    restatement of both loops. The two cutoff conventions differ (`<= cutoff` at :207, `< cutoff`
    at :309), so cutoffs equal to actual row counts are included."""
    rng = np.random.default_rng(M + N)
    draws = rng.choice([1, 3, 8, 20, 60, 200, M // 8, M // 2], size=N)
    mat, n_alts, pos, offs = _list_inputs(M, draws, seed=1000)
    W = mat.shape[1]
    f = orc.leaf(0)
    fl = C.cast(orc.lib.orc_intersect_bitmaps_scalar_list, C.c_void_p)
    truth = orc.truth_columns(mat)
    exact = sorted(set(int(x) for x in n_alts))
    for cutoff in (0, exact[0], exact[len(exact) // 2], exact[len(exact) // 2] + 1, 200, 1 << 30):
        want = orc.lib.orc_wrapper_diag_list(N, _p(mat), W, _p(n_alts), _p(pos), _p(offs), f, fl, cutoff)
        got = lib.STORM_wrapper_diag_list(N, _p(mat), W, _p(n_alts), _p(pos), _p(offs), None, None, cutoff)
        assert got == want == truth, (cutoff, got, want, truth)
        for bs in (0, 7, 31, 500):
            want_b = orc.lib.orc_wrapper_diag_list_blocked(N, _p(mat), W, _p(n_alts), _p(pos), _p(offs),
                                                           f, fl, cutoff, bs)
            got_b = lib.STORM_wrapper_diag_list_blocked(N, _p(mat), W, _p(n_alts), _p(pos), _p(offs),
                                                        None, None, cutoff, bs)
            assert got_b == want_b == truth, (cutoff, bs, got_b, want_b)
    # the library's own leaves are accepted as identity tokens, a foreign function pointer is not
    own_f = C.cast(lib.STORM_intersect_count_scalar, C.c_void_p)
    own_fl = C.cast(lib.STORM_intersect_count_scalar_list, C.c_void_p)
    assert lib.STORM_wrapper_diag_list(N, _p(mat), W, _p(n_alts), _p(pos), _p(offs), own_f, own_fl, 200) == truth
    assert lib.STORM_wrapper_diag_list(N, _p(mat), W, _p(n_alts), _p(pos), _p(offs), f, None, 200) == FAILED


def test_wrapper_diag_list_refuses_lists_that_contradict_the_bitmaps(lib):
    """The device path counts the bitmaps; it equals the reference's list path only when every list
    holds exactly its row's set bits. A list that says otherwise makes the call fail loudly."""
    mat, n_alts, pos, offs = _list_inputs(4096, [5, 9, 2000, 7], seed=7)
    args = lambda na, po: (4, _p(mat), mat.shape[1], _p(na), _p(po), _p(offs), None, None, 200)  # noqa: E731
    ok = lib.STORM_wrapper_diag_list(*args(n_alts, pos))
    assert ok != FAILED
    short = n_alts.copy(); short[1] -= 1                       # list shorter than the row's popcount
    assert lib.STORM_wrapper_diag_list(*args(short, pos)) == FAILED
    moved = pos.copy(); moved[0] ^= 1                            # a listed position whose bit is not set
    if (int(mat[0, moved[0] >> 6]) >> int(moved[0] & 63)) & 1 == 0:
        assert lib.STORM_wrapper_diag_list(*args(n_alts, moved)) == FAILED
        assert b"list" in lib.STORM_hip_error()
    # a dense row (n_alts > cutoff) is never routed to the list leaf: its list is not looked at
    junk = pos.copy(); junk[int(offs[2]):int(offs[2]) + 10] = 0
    assert lib.STORM_wrapper_diag_list(*args(n_alts, junk)) == ok


@pytest.mark.parametrize("draws", (524, 5242, 20971, 52428, 131072, 262144))
def test_sparse_container_at_full_c4_size(hip_ctx, orc, draws):
    """BASELINE config 4 at its real size: STORM_t, N = 10000 rows x M = 524288 bits, at the six
    README loads (README.md:70-77; benchmark.cpp:605-613), through STORM_add + both all-pairs entry
    points. A CPU pairwise oracle needs 30 s .. 1 h here, so the full-size total is checked
    against the column identity of the same bits on the device, and a 600-row subset (rows
    4700..5299 of the same matrix) is checked pairwise against the oracle's STORM_t restatement."""
    M, N = 524288, 10000
    m = hip_ctx.matrix(N, M // 64)
    m.fill_synthetic(M, draws, seed=42)
    want = m.column_identity()
    sub = m.download(4700, 600)
    m.close()
    s = sb.Storm()
    assert s.add_synthetic(M, N, draws, seed=42) == N
    assert s.pairw_intersect_cardinality_blocked(0) == want
    assert s.pairw_intersect_cardinality() == want          # second call: cached device arena
    s.free()
    rows = synth.positions_from_dense(sub)
    o = orc.storm(rows)
    sub_want = o.pairw_blocked(0) if draws <= 5242 or draws >= 131072 else orc.truth_columns(sub)
    s2 = sb.Storm()
    for r in rows:
        s2.add(r)
    assert s2.serialized_size() == o.serialized_size()
    assert s2.pairw_intersect_cardinality_blocked(0) == sub_want == orc.truth_columns(sub)
    s2.free()


def test_storm_t_under_in_process_multi_device_sharding(lib, orc):
    """STORM_hip_set_devices([0, 0, 0]) with a STORM_t handle: three contexts on the one GPU of
    the box, each takes a disjoint shard of the block-column work, partials added on the host."""
    M, N, d = 196608, 1500, 12690   # list- and bitmap-kind blocks mix (about 4230 draws per block)
    rows = synth.positions(M, N, d, seed=11)
    want = orc.storm(rows[:400]).pairw_blocked(0)
    s = sb.Storm()
    for r in rows[:400]:
        s.add(r)
    one = s.pairw_intersect_cardinality()
    assert one == want
    ids = (C.c_int * 3)(0, 0, 0)
    try:
        assert lib.STORM_hip_set_devices(3, ids) == 0
        assert s.pairw_intersect_cardinality() == want             # arena rebuilt for the new config
        assert s.pairw_intersect_cardinality_blocked(0) == want
        big = sb.Storm()
        assert big.add_synthetic(M, N, d, seed=11) == N
        m = sb.HipContext(0).matrix(N, M // 64)
        m.fill_synthetic(M, d, seed=11)
        assert big.pairw_intersect_cardinality() == m.column_identity()
        big.free()
    finally:
        one_dev = (C.c_int * 1)(0)
        assert lib.STORM_hip_set_devices(1, one_dev) == 0
    assert s.pairw_intersect_cardinality() == want
    s.free()


def test_strip_kernel_mfma_shapes_agree(hip_ctx, orc):
    """The default strips run on v_mfma 16x16x128 (option k2_shape = 16); the 32x32x64 form
    (k2_shape = 32) is kept as an independent operand path. Both against the oracle, incl. ragged
    row counts around the 64-row waves' own diagonal blocks, and shards."""
    try:
        for M, N, d in ((4096, 256, 2048), (1000, 131, 300), (65536, 513, 9000), (300, 65, 100), (20000, 1029, 7000)):
            mat = synth.dense_matrix_c(M, N, d, seed=N)
            want = orc.wrapper_diag_blocked(mat, 31)
            m = hip_ctx.matrix_from_host(mat)
            for shape in (16, 32):
                hip_ctx.set_option("k2_shape", shape)
                assert m.pairw() == want, (M, N, d, shape)
                assert sum(m.pairw(r, 5) for r in range(5)) == want, (M, N, d, shape)
            m.close()
    finally:
        hip_ctx.set_option("k2_shape", 16)
    assert hip_ctx.get_option("k2_shape") == 16
