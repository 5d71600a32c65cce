"""Round-6 GPU parity tests. Everything goes through the C-ABI; the oracle is the checker only."""
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


def _reset_tiles(ctx):
    for k, v in (("k2_tile_shape", 0), ("k2_wave_below", 400), ("k2_part_slots", 0), ("k2_part_min_chunks", 8),
                 ("k2_part_cost_diag", 80)):
        ctx.set_option(k, v)


WAVE_SHAPES = ((4096, 256, 2048), (640, 65, 200), (1000, 257, 300), (9000, 700, 3000), (300, 130, 100), (512, 300, 100),
               (520, 31, 100), (70000, 1029, 20000), (1536, 1301, 500), (64, 97, 20), (131072 + 64, 190, 40000))


def test_tile128_kernel_against_the_oracle(hip_ctx, orc):
    """tile128_kernel (K2h, k2_tile_shape = 6: 128 x 128 tiles, rows staged through the LDS, k-parts whose sums meet inside
    the launch — windows in register layout, a ticket per tile, the last part adds and writes) against the oracle's per-pair
    counts (storm.c:1199-1238 with the leaf's result kept per pair): triangle, AND / OR / XOR, rows of zero, ragged last
    tiles (1 .. 127 rows beyond a multiple of 128), row lengths that are not whole 512-bit chunks or whole trips of four
    chunks, one chunk, one tile; with whole tiles, with one and two segments per CU and with parts down to one chunk."""
    try:
        hip_ctx.set_option("k2_tile_shape", 6)
        for M, N, d in WAVE_SHAPES:
            mat = synth.dense_matrix_c(M, N, d, seed=N + M)
            mat[N // 3] = 0
            m = hip_ctx.matrix_from_host(mat)
            want = {"and": np.triu(orc.tile_counts(mat, 0, N, 0, N), k=1).astype(np.uint32)}
            rc = m.row_counts().astype(np.uint32)
            s = rc[:, None] + rc[None, :]
            want["or"] = np.triu(s - want["and"], k=1)
            want["xor"] = np.triu(s - 2 * want["and"], k=1)
            for slots, min_chunks in ((0, 8), (1, 1), (2, 1), (2, 3)):
                hip_ctx.set_option("k2_part_slots", slots)
                hip_ctx.set_option("k2_part_min_chunks", min_chunks)
                for op in ("and", "or", "xor"):
                    for rep in range(2):   # (again: the tickets of the call before must have been left at zero)
                        got = m.pairw_matrix(op)
                        assert hip_ctx.get_option("k2_tile_shape_used") == 6
                        assert np.array_equal(want[op], got), (M, N, slots, min_chunks, op, rep,
                                                               np.argwhere(want[op] != got)[:3].tolist())
            m.close()
    finally:
        _reset_tiles(hip_ctx)


def test_tile128_kernel_bands_rectangles_and_the_automatic_rule(hip_ctx, orc):
    """K2h on a band of the triangle left in device memory (rows that start inside a tile), on the rectangle of two matrices
    (STORM_wrapper_square's shape, storm.c:153-171), and as what k2_tile_shape = 0 chooses for matrices of few 256 x 256 tiles —
    against the oracle on sampled tiles and against tilebits8_kernel entry by entry."""
    import torch
    try:
        for M, N, d in ((9000, 700, 3000), (4096, 2309, 1500), (1536, 4700, 500)):
            mat = synth.dense_matrix_c(M, N, d, seed=N + M)
            m = hip_ctx.matrix_from_host(mat)
            hip_ctx.set_option("k2_tile_shape", 2)
            ref = m.pairw_matrix("and")
            assert np.array_equal(ref[:200, :N], np.triu(orc.tile_counts(mat, 0, 200, 0, N), k=1)[:200].astype(np.uint32))
            for wt in (1, 2):
                hip_ctx.set_option("k2_tile_shape", 6)
                hip_ctx.set_option("k2_part_slots", wt)
                got = m.pairw_matrix("and")
                assert np.array_equal(ref, got), (M, N, wt, np.argwhere(ref != got)[:3].tolist())
                band = torch.zeros((300, N + 5), dtype=torch.int32, device="cuda:0")
                m.pairw_matrix_band_device(band.data_ptr(), N + 5, 333, 300, "xor")
                rcn = m.row_counts().astype(np.int64)
                want = np.triu((rcn[:, None] + rcn[None, :] - 2 * ref.astype(np.int64)), k=1)[333:633]
                assert np.array_equal(np.triu(band.cpu().numpy().astype(np.int64)[:, :N], k=334), want), (M, N, wt)
                na = N // 3
                ma, mb = hip_ctx.matrix_from_host(mat[:na]), hip_ctx.matrix_from_host(mat[na:])
                got = ma.square_matrix(mb, "and")
                assert hip_ctx.get_option("k2_tile_shape_used") == 6
                full = ref + ref.T
                assert np.array_equal(got, full[:na, na:]), (M, N, wt)
                if N <= 800:
                    assert np.array_equal(got, orc.tile_counts(mat, 0, na, na, N).astype(np.uint32))
                ma.close()
                mb.close()
            # the automatic rule: few tiles -> K2w, and the same numbers
            _reset_tiles(hip_ctx)
            got = m.pairw_matrix("or")
            hip_ctx.set_option("k2_tile_shape", 2)
            assert np.array_equal(got, m.pairw_matrix("or"))
            m.close()
        _reset_tiles(hip_ctx)
        m = hip_ctx.matrix(2048, 1024)
        m.fill_synthetic(65536, 20000, seed=5)
        out = torch.zeros((2048, 2048), dtype=torch.int32, device="cuda:0")
        m.pairw_matrix_device(out.data_ptr(), 2048, "and")
        assert hip_ctx.get_option("k2_tile_shape_used") == 6
        assert int(out.to(torch.int64).sum().item()) == m.pairw() == m.column_identity()
        m.close()
    finally:
        _reset_tiles(hip_ctx)
