"""The multi-rank path on CPU (gloo, world_size 2) with the ownership rule the DEFAULT device
path uses: every rank asks libstorm_hip.so for its strip work list (storm_hip_strip_plan —
host-only, the same planner the launch uses), the lists of all ranks must tile (pair, k-slice)
space exactly once and balance, and the all-reduced partial totals must equal the single-rank
total. The per-item partials here come from the ORACLE (checker) because the product kernels
need a GPU; what is under test is the ownership rule and the all-reduce plumbing of
stormbitmaps_amd/dist.py, which bench.py uses unchanged with backend "nccl" (RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from stormbitmaps_amd import dist as sdist
from stormbitmaps_amd import synth

A_TILE, B_BLOCK = 256, 64
# (form, pair_space): the default path (K2b, class-pair slices) with both ownership modes, and the FP4-shadow strips
MODES = ((1, 0), (1, 1), (0, 0))


def _n_kslices(n_words, form=1):
    """slices that hold data (padding is never multiplied): form 0: 256 consecutive bits; form 1: a class pair of a 512-bit chunk"""
    return (n_words + 3) // 4 if form == 0 else 2 * ((n_words + 7) // 8)


def _cover(n_rows, n_words, world, form=1, pair_space=0):
    """cover[ks][i, j] = how many items of all ranks count pair (i, j) in k-slice ks; per-rank cost."""
    pad = (n_rows + A_TILE - 1) // A_TILE * A_TILE
    ks_n = _n_kslices(n_words, form)
    cover = np.zeros((ks_n, pad, pad), dtype=np.uint8)
    cost = []
    for r in range(world):
        items = sdist.strip_plan(n_rows, n_words, r, world, form, pair_space)
        c = 0
        for a0, diag, j0, j1, ks in items.tolist():
            if diag:
                cover[ks, a0:a0 + A_TILE, a0:a0 + A_TILE] += np.triu(np.ones((A_TILE, A_TILE), np.uint8), k=1)
            cover[ks, a0:a0 + A_TILE, j0 * B_BLOCK:min(j1 * B_BLOCK, pad)] += 1
            c += (j1 - j0) + 4 * diag
        cost.append(c)
    return cover, cost


@pytest.mark.parametrize("n_rows,n_words,worlds", [
    (700, 32, (1, 2, 3, 8)),      # 8 k-slices: whole slices + leftover mix
    (1500, 4, (1, 2, 3, 5, 8)),   # ONE k-slice: fewer slices than ranks -> pure pair-space split
    (300, 100, (2, 3, 7)),        # 25 slices, 7 ranks: 3 whole each + 4 leftover
])
@pytest.mark.parametrize("form,pair_space", MODES)
def test_strip_plans_of_all_ranks_tile_the_work_exactly_once(n_rows, n_words, worlds, form, pair_space):
    pad = (n_rows + A_TILE - 1) // A_TILE * A_TILE
    want = np.zeros((pad, pad), dtype=np.uint8)
    want[:n_rows, :] = np.triu(np.ones((n_rows, pad), np.uint8), k=1)   # B blocks run to the padded edge
    for world in worlds:
        cover, cost = _cover(n_rows, n_words, world, form, pair_space)
        assert cover.shape[0] == _n_kslices(n_words, form)
        for ks in range(cover.shape[0]):
            # every real pair i < j < n_rows exactly once; padded (all-zero) rows may be visited, never twice
            assert np.array_equal(cover[ks][:n_rows, :n_rows], want[:n_rows, :n_rows]), (world, ks)
            assert cover[ks].max() <= 1
        if not pair_space:   # (pair mode cuts every slice into the same items: same stage count by construction)
            assert sum(cost) == _cover(n_rows, n_words, 1, form, pair_space)[1][0] or world == 1


@pytest.mark.parametrize("form,pair_space", MODES)
def test_headline_shape_balances_within_three_percent_for_any_world(form, pair_space):
    """c2 (N = 10000, W = 1024: 256 k-slices): stage counts per rank for G = 2..8."""
    for world in (2, 3, 4, 5, 6, 7, 8):
        cost = []
        for r in range(world):
            it = sdist.strip_plan(10000, 1024, r, world, form, pair_space)
            cost.append(int((it[:, 3] - it[:, 2]).sum() + 4 * it[:, 1].sum()))
        assert max(cost) <= 1.03 * (sum(cost) / world), (world, cost)
    # a matrix with a single k-slice (form 1: one chunk = two class-pair slices) still splits 8 ways (pair-space sharding)
    cost = []
    for r in range(8):
        it = sdist.strip_plan(10000, 4, r, 8, form, pair_space)
        cost.append(int((it[:, 3] - it[:, 2]).sum() + 4 * it[:, 1].sum()))
    assert min(cost) > 0 and max(cost) <= 1.03 * (sum(cost) / 8), cost


def _block_cover(n_rows, n_words, world, form, pair_space, **opts):
    """Coverage at (k-slice, A tile, B block) granularity — cheap enough for the shapes where the automatic run
    length differs from shape to shape: cover[ks, tile, block] and diag[ks, tile] summed over all ranks' lists,
    and the run length every rank reports."""
    n_tiles = (n_rows + A_TILE - 1) // A_TILE
    n_blocks = n_tiles * (A_TILE // B_BLOCK)
    ks_n = _n_kslices(n_words, form)
    cover = np.zeros((ks_n, n_tiles, n_blocks), dtype=np.int32)
    diag = np.zeros((ks_n, n_tiles), dtype=np.int32)
    runs = []
    for r in range(world):
        items, run = sdist.strip_plan(n_rows, n_words, r, world, form, pair_space, return_run=True, **opts)
        runs.append(run)
        for a0, dg, j0, j1, ks in items.tolist():
            cover[ks, a0 // A_TILE, j0:j1] += 1
            diag[ks, a0 // A_TILE] += dg
    return cover, diag, runs


@pytest.mark.parametrize("n_rows", [6144, 8192])
@pytest.mark.parametrize("pair_space,opts", [
    (1, {}),                    # every slice dealt along the pair space, automatic run length
    (0, {"tail_run": 96}),      # default ownership, leftover slices cut at min(run, tail_run) with a long tail_run
    (1, {"n_cus": 64}),         # a smaller device: other makespans, same rule
])
def test_every_rank_cuts_the_dealt_slices_at_the_same_run_length(n_rows, pair_space, opts):
    """the automatic run length was chosen from the calling rank's own list — N = 6144 at world 3
    gave ranks [64, 96, 96], N = 8192 [96, 128, 128] — while the pair-space deal only partitions the work if every rank
    cuts the slices identically. The choice is now made from all ranks' lists (slowest rank decides)."""
    world, n_words = 3, 1024 + 8 * (1 - pair_space)   # default mode: 258 slices = 3 x 84 whole + 6 leftover... units of 4
    cover, diag, runs = _block_cover(n_rows, n_words, world, 1, pair_space, **opts)
    assert len(set(runs)) == 1 and runs[0] in (64, 96, 128), runs
    n_tiles = cover.shape[1]
    per = A_TILE // B_BLOCK
    want = np.zeros(cover.shape[1:], dtype=np.int32)
    for t in range(n_tiles):
        want[t, (t + 1) * per:] = 1
    assert np.array_equal(diag, np.ones_like(diag))
    for ks in range(cover.shape[0]):
        assert np.array_equal(cover[ks], want), ks


def test_plan_with_explicit_options_matches_the_defaults_and_a_fixed_run():
    """storm_hip_strip_plan2 == storm_hip_strip_plan3 with the context's defaults; a fixed max_run is honoured."""
    a = sdist.strip_plan(3000, 64, 1, 2)
    b, run = sdist.strip_plan(3000, 64, 1, 2, max_run=0, tail_run=32, tail_slices=3, lpt_rounds=6, n_cus=256, return_run=True)
    assert np.array_equal(a, b) and run in (64, 96, 128)
    c, run = sdist.strip_plan(3000, 64, 0, 1, max_run=16, return_run=True)
    assert run == 16 and int((c[:, 3] - c[:, 2]).max()) <= 16


def _item_total(orc, mat, item, form=1):
    """Oracle partial of one strip item: pairs (A tile x B blocks [+ own triangle]) on k-slice ks."""
    a0, diag, j0, j1, ks = (int(x) for x in item)
    n = mat.shape[0]
    sl = sdist.slice_columns(mat, ks, form)
    if sl.shape[1] == 0:
        return 0
    a1 = min(a0 + A_TILE, n)
    total = 0
    if diag and a1 - a0 > 1:
        total += int(np.triu(orc.tile_counts(sl, a0, a1, a0, a1), k=1).sum())
    b0, b1 = min(j0 * B_BLOCK, n), min(j1 * B_BLOCK, n)
    if b1 > b0 and a1 > a0:
        total += int(orc.tile_counts(sl, a0, a1, b0, b1).sum())
    return total


def _worker(rank, world, port, n_rows, n_bits, want, q, form=1, pair_space=0):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tests._orc import Oracle
    orc = Oracle()
    d = sdist.init_process_group("gloo")
    mat = synth.dense_matrix(n_bits, n_rows, 700, seed=42)

    def partial(r, w):
        return sum(_item_total(orc, mat, it, form) for it in sdist.strip_plan(n_rows, mat.shape[1], r, w, form, pair_space))

    total = sdist.sharded_pairw(partial)
    q.put((rank, total, partial(rank, world)))
    d.barrier()
    d.destroy_process_group()
    assert total == want


@pytest.mark.parametrize("n_bits,form,pair_space", [
    (2048, 1, 0),   # default path (K2b): 8 class-pair slices: whole slices only
    (700, 1, 0),    # 4 class-pair slices (one of them over the ragged last chunk): leftover slices along the pair space
    (2048, 1, 1),   # every slice along the pair space (k2_shard_pairs)
    (700, 0, 0),    # FP4-shadow strips: 3 slices of 256 consecutive bits
])
def test_two_rank_gloo_allreduce_equals_single_rank(orc, n_bits, form, pair_space):
    n_rows = 700
    mat = synth.dense_matrix(n_bits, n_rows, 700, seed=42)
    want = orc.wrapper_diag(mat)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rows, n_bits, want, q, form, pair_space)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == want
    assert res[0][2] + res[1][2] == want and res[0][2] != want


# ---- the one-launch stage stream on bit operands (K2q): matrices of up to 8192 rows on one device ----
def _stream_cover(n_rows, n_words, world, n_cus):
    """cover[ks][p, q] (p <= q, 64-row blocks) = how often the segments of all ranks' workgroups multiply
    the unordered block pair {p, q} in k-slice ks (512 bits); stages per workgroup of every rank."""
    nb = (n_rows + 63) // 64
    ks_n = (n_words + 7) // 8
    cover = np.zeros((ks_n, nb + 4, nb + 4), dtype=np.int32)
    loads = []
    for r in range(world):
        segs, groups = sdist.stream_plan(n_rows, n_words, r, world, n_cus)
        per_wg = np.zeros(max(groups, 1), dtype=np.int64)
        for wg, a_blk, ks, b_first, n_b, range_nb, diag, stages in segs.tolist():
            assert range_nb == nb and stages == 4 + n_b and wg < groups
            per_wg[wg] += stages
            tile = [a_blk + i for i in range(4)]
            if diag:
                for i, p in enumerate(tile):          # wave i: its own block (half weight twice = once) ...
                    cover[ks, p, p] += 1
                    for q in tile[i + 1:]:            # ... and the tile's later blocks
                        cover[ks, p, q] += 1
            for i in range(n_b):
                q = (b_first + i) % range_nb
                for p in tile:
                    cover[ks, min(p, q), max(p, q)] += 1
        loads.append(per_wg)
    return cover, loads


@pytest.mark.parametrize("n_rows,n_words,worlds,n_cus", [
    (65, 10, (1, 2), 256),          # two blocks, one tile, a ragged second k-slice
    (700, 32, (1, 2, 3, 8), 256),   # 3 tiles (odd): every tile takes one tile behind it
    (1024, 128, (1, 3), 256),       # 4 tiles (even): the opposite tile alternates with the k-slice
    (1100, 20, (1, 2, 5), 4),       # 5 tiles, last one ragged; a 4-CU device cuts segments in the middle
    (2048, 64, (1, 8), 16),         # 8 tiles, many cuts: continued segments bring A in without multiplying
    (2300, 8, (1, 3), 2),           # ONE k-slice
    (4000, 64, (1, 2, 3), 2),       # many rounds on a 2-CU device: whole segments as shares + a cut last round
    (2600, 200, (1, 4), 1),         # the same with ragged tiles and a ragged last k-slice
])
def test_stream_plans_of_all_ranks_cover_every_block_pair_exactly_once(n_rows, n_words, worlds, n_cus):
    nb = (n_rows + 63) // 64
    for world in worlds:
        cover, loads = _stream_cover(n_rows, n_words, world, n_cus)
        want = np.triu(np.ones((nb, nb), np.int32))              # every unordered pair incl. a block with itself
        for ks in range(cover.shape[0]):
            assert np.array_equal(cover[ks][:nb, :nb], want), (world, ks)
            # blocks beyond the matrix (the zero rows up to the tile's 256) may be touched by a tile, never twice
            assert cover[ks].max() <= 1
        # the ranks' shares are contiguous parts of one stream: equal to within the snapping of a cut
        totals = [int(l.sum()) for l in loads]
        assert max(totals) - min(totals) <= 4 * 64 + 16, totals


def test_stream_plan_shares_at_the_sizes_the_bench_reports():
    """N = 512 ... 8192 at M = 65536 on 256 CUs: whole rounds of the chip's 768 workgroup slots (or 256 / 512 for
    short streams). Short shares and the shares of a multi-round deal are equal to within a few stages; ONE round
    of long shares (from 16 stages) is dealt 100 : 120 : 60 by dispatch round (workgroup w / 256), for the SIMD
    arbiter's oldest-first issue (build_bitstream)."""
    for n_rows in (512, 1024, 1536, 2048, 4096, 6144, 8192):
        segs, groups = sdist.stream_plan(n_rows, 1024, 0, 1, 256)
        assert groups % 256 == 0 and groups >= 256
        per_wg = np.bincount(segs[:, 0], weights=segs[:, 7], minlength=groups)
        assert per_wg.min() > 0
        if groups == 768 and per_wg.mean() >= 20:
            r = [per_wg[k * 256:(k + 1) * 256].mean() for k in range(3)]
            assert abs(r[1] / r[0] - 1.2) < 0.08 and abs(r[2] / r[0] - 0.6) < 0.08, (n_rows, r)
            for k in range(3):
                part = per_wg[k * 256:(k + 1) * 256]
                assert part.max() <= part.mean() * 1.15 + 6, (n_rows, k, part.min(), part.mean(), part.max())
        else:
            assert per_wg.max() <= per_wg.mean() * 1.3 + 6, (n_rows, groups, per_wg.min(), per_wg.mean(), per_wg.max())
