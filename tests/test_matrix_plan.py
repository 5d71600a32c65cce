"""CPU tests of the host-only planner of the materialised-output kernel for matrices of few tiles (K2h,
storm_hip_matrix_plan): the items must cover every pair of the output and every chunk of every tile exactly once — the
planner cuts the reference loop storm.c:1199-1238 (per-pair results kept) into work for the CUs."""
import numpy as np
import pytest

from stormbitmaps_amd import dist


def _check_cover(plan, n_chunks, tiles_expected):
    tiles = {}
    for I, J, c0, n, tile, part, n_parts, narrow in plan.tolist():
        tiles.setdefault(tile, []).append((part, c0, n, n_parts, I, J, narrow))
    assert len(tiles) == len(tiles_expected)
    seen = set()
    for tile, parts in tiles.items():
        parts.sort()
        assert [p[0] for p in parts] == list(range(len(parts)))
        assert all(p[3] == len(parts) for p in parts)
        assert len({(p[4], p[5]) for p in parts}) == 1
        seen.add((parts[0][4], parts[0][5]))
        pos = 0
        for _, c0, n, _, _, _, narrow in parts:
            assert c0 == pos and n >= 1
            assert n * 512 < (1 << 24)            # f32 accumulators stay exact
            assert not narrow or (len(parts) > 1 and n <= 127)   # 16-bit windows only below 2^16 bits of k
            pos += n
        assert pos == n_chunks
    assert seen == tiles_expected


@pytest.mark.parametrize("n_rows", [1, 2, 127, 128, 129, 256, 1000, 1024, 1536, 2048, 3072, 4096, 5000, 6144])
@pytest.mark.parametrize("slots", [0, 1, 2])
def test_triangle_plans_cover_every_tile_and_chunk_once(n_rows, slots):
    for n_words, min_chunks in ((1024, 8), (1, 8), (70, 1), (1094, 3), (8192, 8)):
        plan = dist.matrix_plan(n_rows, n_words, slots_per_cu=slots, min_chunks=min_chunks)
        nt = (n_rows + 127) // 128
        want = {(i, j) for i in range(nt) for j in range(i, nt)}
        _check_cover(plan, (n_words + 7) // 8, want)
        # longest first: the dispatcher hands the short items to the slots that end first
        assert (np.diff(plan[:, 3].astype(np.int64)) <= 0).all()


def test_bands_rectangles_long_rows_and_the_balance_of_the_cut():
    # a band of the triangle: the tile rows that hold the band, against every later tile column
    plan = dist.matrix_plan(3000, 512, band_row0=300, band_rows=700)
    _check_cover(plan, 64, {(i, j) for i in range(300 // 128, (1000 + 127) // 128) for j in range(i, (3000 + 127) // 128)})
    # rectangle: B's tiles count on behind A's rows padded to 256
    plan = dist.matrix_plan(700, 512, n_rows_b=300)
    _check_cover(plan, 64, {(i, 6 + j) for i in range(6) for j in range(3)})
    # rows of 2^25 bits: every tile is cut (an item stays below 2^24 bits), windows of 32-bit counts
    plan = dist.matrix_plan(256, 1 << 19)
    _check_cover(plan, 1 << 16, {(0, 0), (0, 1), (1, 1)})
    assert (plan[:, 6] >= 3).all() and (plan[:, 7] == 0).all()
    # fewer tiles than CUs: as many items as the chip has slots, none much longer than its share
    for n_rows in (1024, 1536, 2048):
        plan = dist.matrix_plan(n_rows, 1024)
        nt = (n_rows + 127) // 128
        share = nt * (nt + 1) // 2 * 128 / len(plan)
        assert len(plan) in (256, 512) and plan[:, 3].max() <= 1.3 * share + 1
    # more: whole rounds of tiles stay whole, the rest is cut into one part per slot
    plan = dist.matrix_plan(4096, 1024)
    assert (plan[:, 6] == 1).sum() == 512 and len(plan) - 512 <= 512
