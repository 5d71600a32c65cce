"""The CPU oracle against everything that pins it (no GPU needed).

1. the totals SURVEY.md recorded from the unmodified reference (tests/golden/survey_totals.json)
2. hand-computed matrices (tests/golden/tiny.json)
3. two independent truths (naive bit loop, column-count identity) on seeded random inputs
4. the committed splitmix64 vectors (tests/golden/synth_totals.json)
"""
import json
import os

import numpy as np
import pytest

from stormbitmaps_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def _dense_from_rows(M, rows):
    W = (M + 63) // 64
    mat = np.zeros((len(rows), W), dtype=np.uint64)
    for i, r in enumerate(rows):
        for v in r:
            mat[i, v // 64] |= np.uint64(1) << np.uint64(v % 64)
    return mat


@pytest.mark.parametrize("case", _load("survey_totals.json")["agree"],
                         ids=lambda c: f"M{c['M']}_N{c['N']}_d{c['draws']}")
def test_oracle_reproduces_reference_totals_recorded_by_survey(orc, case):
    rows = orc.mt_positions(case["M"], case["N"], case["draws"], 42)
    c = orc.contig(case["M"], rows)
    s = orc.storm(rows)
    mat = c.dense()
    got = {
        "contig": c.pairw(), "contig_blocked_opt": c.pairw_blocked(max(5, 256000 // (mat.shape[1] * 8))),
        "contig_blocked_7": c.pairw_blocked(7), "storm": s.pairw(), "storm_blocked_0": s.pairw_blocked(0),
        "wrapper_diag": orc.wrapper_diag(mat, 0), "wrapper_diag_blocked": orc.wrapper_diag_blocked(mat, 31, 0),
        "columns": orc.truth_columns(mat),
    }
    assert set(got.values()) == {case["total"]}, got


@pytest.mark.parametrize("case", _load("survey_totals.json")["defects"],
                         ids=lambda c: f"{c['id']}_d{c['draws']}")
def test_oracle_returns_truth_where_reference_is_defective(orc, case):
    rows = orc.mt_positions(case["M"], case["N"], case["draws"], 42)
    c = orc.contig(case["M"], rows)
    mat = c.dense()
    truth = orc.truth_columns(mat)
    assert truth == case["truth"]  # the survey's naive truth
    assert truth != case["reference_value"]  # documented divergence of the reference
    if case["container"] == "STORM_t":
        s = orc.storm(rows)
        n_list, n_bitmap = s.census()
        assert n_list > 0 and n_bitmap > 0  # D1 needs mixed block kinds
        assert s.pairw() == s.pairw_blocked(0) == truth
    else:
        assert c.cutoff() > case["draws"] or case["draws"] < 200  # rows are on the list path
        assert c.pairw() == c.pairw_list() == c.pairw_blocked(31) == c.pairw_blocked_list(5) == truth


@pytest.mark.parametrize("case", _load("tiny.json")["cases"], ids=lambda c: c["name"])
def test_oracle_hand_computed(orc, case):
    c = orc.contig(case["M"], case["rows"])
    s = orc.storm(case["rows"])
    assert c.pairw() == case["total"]
    assert c.pairw_blocked(3) == case["total"]
    assert s.pairw() == case["total"]
    assert s.pairw_blocked(0) == case["total"]
    nonempty = [r for r in case["rows"] if len(r)]
    assert orc.truth_naive(_dense_from_rows(case["M"], nonempty)) == case["total"]
    # per-pair AND / OR / XOR counts against plain set arithmetic (all rows, empty ones included)
    dense = _dense_from_rows(case["M"], case["rows"])
    n = len(case["rows"])
    for op, name in enumerate(("and", "or", "xor")):
        got = orc.tile_counts_op(dense, 0, n, 0, n, op)
        for i, j, want in case["pair_counts"][name]:
            assert got[i, j] == want == got[j, i], (name, i, j)
        assert orc.truth_naive_op(dense, op) == sum(x[2] for x in case["pair_counts"][name])


def test_leaves_agree_on_random_words(orc):
    rng = np.random.default_rng(7)
    for n in (0, 1, 3, 4, 7, 8, 15, 16, 17, 63, 64, 1024, 1031):
        a = rng.integers(0, 2**64, size=n, dtype=np.uint64)
        b = rng.integers(0, 2**64, size=n, dtype=np.uint64)
        want = int(sum(bin(int(x) & int(y)).count("1") for x, y in zip(a, b)))
        pa, pb = a.ctypes.data, b.ctypes.data
        assert orc.lib.orc_intersect_count_scalar(pa, pb, n) == want
        assert orc.lib.orc_intersect_count_avx2(pa, pb, n) == want
        assert orc.lib.orc_intersect_count_avx512(pa, pb, n) == want


@pytest.mark.parametrize("M,N,d", [(4096, 64, 2048), (1000, 37, 300), (65536, 40, 32768),
                                   (200, 5, 10), (64, 9, 64)])
def test_every_entry_point_equals_both_truths(orc, M, N, d):
    mat = synth.dense_matrix(M, N, d, seed=11)
    rows = synth.positions_from_dense(mat)
    naive = orc.truth_naive(mat)
    assert naive == orc.truth_columns(mat)
    c = orc.contig(M, rows)
    s = orc.storm(rows)
    got = [c.pairw(), c.pairw_blocked(0), c.pairw_blocked(3), c.pairw_blocked(1000),
           s.pairw(), s.pairw_blocked(0), s.pairw_blocked(6)]
    for kind in (0, 1, 2, 3):
        if orc.leaf(kind):
            got += [orc.wrapper_diag(mat, kind), orc.wrapper_diag_blocked(mat, 0, kind),
                    orc.wrapper_diag_blocked(mat, 5, kind)]
    assert set(got) == {naive}


def test_committed_synth_vectors_small_ones(orc):
    g = _load("synth_totals.json")
    for case in g["dense"]:
        if case["N"] * case["N"] * ((case["M"] + 63) // 64) > 6e8:
            continue
        mat = synth.dense_matrix(case["M"], case["N"], case["draws"], seed=g["seed"])
        assert orc.wrapper_diag_blocked(mat, 31) == case["total"], case["name"]
    for case in g["sparse"]:
        if case["draws"] > 6000:
            continue
        rows = synth.positions(case["M"], case["N"], case["draws"], seed=g["seed"])
        s = orc.storm(rows)
        assert s.pairw_blocked(0) == case["total"], case["name"]
        assert s.census() == (case["list_blocks"], case["bitmap_blocks"])
        assert s.serialized_size() == case["serialized_size"]


def test_wrapper_square_and_additivity(orc):
    mat = synth.dense_matrix(4096, 50, 1500, seed=3)
    a, b = mat[:20], mat[20:]
    sq = orc.wrapper_square(a, b)
    want = sum(int(orc.tile_counts(mat, i, i + 1, 20, 50).sum()) for i in range(20))
    assert sq == want
    assert orc.wrapper_diag(mat) == orc.wrapper_diag(a) + orc.wrapper_diag(b) + sq


def test_list_wrappers_cutoff_semantics(orc):
    # diag_list treats n_alts <= cutoff as sparse (storm.c:207), list_blocked n_alts < cutoff (:309);
    # either way the count is the exact one.
    M, N = 2048, 30
    mat = np.concatenate([synth.dense_matrix(M, 15, 8, seed=5), synth.dense_matrix(M, 15, 900, seed=6)])
    rows = synth.positions_from_dense(mat)
    n_alts = np.array([len(r) for r in rows], dtype=np.uint32)
    offs = np.zeros(N, dtype=np.uint32)
    offs[1:] = np.cumsum(n_alts)[:-1]
    pos = np.concatenate(rows).astype(np.uint32)
    truth = orc.truth_naive(mat)
    f = orc.leaf(0)
    fl = orc.lib.orc_intersect_bitmaps_scalar_list
    import ctypes as C
    fl_ptr = C.cast(fl, C.c_void_p)
    p = lambda a: a.ctypes.data  # noqa: E731
    for cutoff in (0, 8, 9, 10_000):
        assert orc.lib.orc_wrapper_diag_list(N, p(mat), mat.shape[1], p(n_alts), p(pos), p(offs), f, fl_ptr, cutoff) == truth
        for bs in (0, 4, 7, 64):
            assert orc.lib.orc_wrapper_diag_list_blocked(N, p(mat), mat.shape[1], p(n_alts), p(pos), p(offs), f, fl_ptr, cutoff, bs) == truth


def test_vector16_and_vector32(orc):
    rng = np.random.default_rng(1)
    for la, lb in ((0, 5), (5, 0), (1, 1), (8, 8), (9, 17), (100, 3000), (4095, 4095)):
        a = np.unique(rng.integers(0, 65536, size=la, dtype=np.uint16)).astype(np.uint16)
        b = np.unique(rng.integers(0, 65536, size=lb, dtype=np.uint16)).astype(np.uint16)
        want = len(np.intersect1d(a, b))
        pa = a.ctypes.data if a.size else None
        pb = b.ctypes.data if b.size else None
        assert orc.lib.orc_intersect_vector16_cardinality(pa, pb, a.size, b.size) == want
    a = np.array([0, 2, 3, 7, 9], dtype=np.uint32)
    b = np.array([1, 2, 7, 8, 9, 11], dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    n = orc.lib.orc_intersect_vector32_unsafe(a.ctypes.data, b.ctypes.data, a.size, b.size, out.ctypes.data)
    assert n == 6 and out[:6].tolist() == [1, 1, 3, 2, 4, 4]  # (idx in a, idx in b) pairs
    assert orc.lib.orc_intersect_vector32_unsafe(a.ctypes.data, b.ctypes.data, a.size, b.size, None) == 0
    assert orc.lib.orc_intersect_vector32_unsafe(a.ctypes.data, b.ctypes.data, 0, b.size, out.ctypes.data) == 0


def test_container_conventions(orc):
    L = orc.lib
    # NULL handle -> (uint64)-1 (storm.c:1150,1176,878,898); list variants -2 before any add (:1245)
    assert L.orc_contig_pairw_intersect_cardinality(None) == 2**64 - 1
    assert L.orc_contig_pairw_intersect_cardinality_blocked(None, 5) == 2**64 - 1
    assert L.orc_storm_pairw_intersect_cardinality(None) == 2**64 - 1
    assert L.orc_storm_pairw_intersect_cardinality_blocked(None, 0) == 2**64 - 1
    c = orc.contig(4096, [])
    assert c.pairw_list() == 2**64 - 2
    assert L.orc_contig_add(None, None, 0) == -1 and L.orc_contig_add(c.h, None, 3) == -2
    v = np.array([1, 5, 5, 9], dtype=np.uint32)
    assert L.orc_contig_add(c.h, v.ctypes.data, 0) == 0 and L.orc_contig_n_rows(c.h) == 0  # storm.c:1034
    assert L.orc_contig_add(c.h, v.ctypes.data, 4) == 4 and L.orc_contig_n_rows(c.h) == 1
    assert c.cutoff() == 20  # min(200, 4096/200), storm.c:1016
    assert orc.contig(65536, []).cutoff() == 200
    s = orc.storm([])
    assert L.orc_storm_add(s.h, v.ctypes.data, 0) == 1 and L.orc_storm_n_rows(s.h) == 1  # storm.c:864
    assert s.serialized_size() == 8 + 12  # one empty row: 3 u32 (storm.c:392) + 2 u32 (:970)
    s.add([3, 70000])
    # two list blocks of one value each: (2 + 16) * 2 + 4 * 2 + 12
    assert s.serialized_size() == 20 + (2 + 16) * 2 + 8 + 12


def test_storm_pair_counts_are_the_dense_and_counts(orc):
    """orc_storm_pair_counts (the row-pair function of storm.c:790-814, every pair written out) against the dense
    tile truth over the same bits, on list blocks, bitmap blocks and both mixed kinds; its sum is the all-pairs total."""
    rng = np.random.default_rng(5)
    M = 3 * 65536
    rows = []
    for r in range(60):
        d = (30, 900, 9000, 60000)[r % 4]
        rows.append(np.unique(rng.integers(0, M, size=d, dtype=np.uint64)).astype(np.uint32))
    rows[7] = np.zeros(0, dtype=np.uint32)
    s = orc.storm(rows)
    got = s.pair_counts()
    want = np.triu(orc.tile_counts(_dense_from_rows(M, rows), 0, 60, 0, 60), k=1)
    assert np.array_equal(got, want)
    assert int(got.sum(dtype=np.uint64)) == s.pairw() == s.pairw_blocked(0)
    assert np.array_equal(s.pair_counts(10, 20), want[10:20])
