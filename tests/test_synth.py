"""The synthetic-input generator: numpy restatement == C generator, and the per-row recipe."""
import numpy as np
import pytest

from stormbitmaps_amd import synth


@pytest.mark.parametrize("M,N,d,seed", [(4096, 33, 2048, 42), (1000, 7, 300, 1), (65536, 5, 32768, 42),
                                        (70, 3, 200, 9), (524288, 2, 1000, 42), (64, 4, 1, 0)])
def test_numpy_equals_c_generator(lib, M, N, d, seed):
    a = synth.dense_matrix(M, N, d, seed=seed)
    b = synth.dense_matrix_c(M, N, d, seed=seed)
    assert a.shape == b.shape == (N, (M + 63) // 64)
    assert np.array_equal(a, b)
    # rows are independent of how many rows are generated around them
    c = synth.dense_matrix_c(M, 1, d, seed=seed, row0=N - 1)
    assert np.array_equal(c[0], b[N - 1])


def test_positions_are_sorted_unique_and_match_bits(lib):
    M, N, d = 5000, 11, 700
    mat = synth.dense_matrix(M, N, d, seed=42)
    rows = synth.positions(M, N, d, seed=42)
    scratch = np.zeros((M + 63) // 64, dtype=np.uint64)
    out = np.zeros(d, dtype=np.uint32)
    for i, r in enumerate(rows):
        assert np.all(np.diff(r.astype(np.int64)) > 0) and r.max() < M and len(r) <= d
        n = lib.storm_synth_positions(out.ctypes.data, scratch.ctypes.data, M, i, d, 42)
        assert np.array_equal(out[:n], r)
    assert [len(x) for x in synth.positions_from_dense(mat)] == [len(r) for r in rows]


def test_first_draws_known_answer():
    # splitmix64(seed=42): first output 0xBDD732262FEB6E95; position = (z * M) >> 64
    z = synth._mix(np.uint64(42) + synth.GOLDEN)
    assert int(z) == 0xBDD732262FEB6E95
    assert int(synth.draws_for_rows(1 << 16, 0, 1, 1, 42)[0, 0]) == (0xBDD732262FEB6E95 * (1 << 16)) >> 64
