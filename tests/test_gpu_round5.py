"""Round-5 GPU parity tests. Everything goes through the C-ABI; the oracle is the checker only."""
import ctypes as C

import numpy as np
import pytest

import stormbitmaps_amd as sb
from stormbitmaps_amd import dist as sdist
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
    for k, v in (("k2_strip_operands", 0), ("k2_fold_inline", -1), ("k2_matrix_pad", -1), ("variant", -1),
                 ("k2_shard_pairs", 0), ("k2_max_run", 0), ("k2_tail_run", 32)):
        ctx.set_option(k, v)


def test_pair_space_shards_sum_to_the_total_where_the_automatic_run_length_differs_by_rank(hip_ctx):
    """with k2_shard_pairs = 1 every rank replays the same longest-first deal of every slice's
    items, so all ranks must cut the slices at the same run length. The automatic choice (k2_max_run = 0) used to be
    made from the calling rank's own list — N = 6144 at world 3 chose [64, 96, 96], N = 8192 [96, 128, 128]: pairs
    dropped or counted twice, silently. The sum of all ranks' partials against the one-device total and the column
    identity (storm.c:1199-1238 is the loop being sharded); the planner reports ONE run length for the world and the
    device launches exactly the planner's item count."""
    try:
        for N in (6144, 8192):
            M = 65536
            m = hip_ctx.matrix(N, M // 64)
            m.fill_synthetic(M, M // 3, seed=N)
            want = m.column_identity()
            assert m.pairw() == want
            for pairs, tail_run in ((1, 32), (0, 96)):
                hip_ctx.set_option("k2_shard_pairs", pairs)
                hip_ctx.set_option("k2_tail_run", tail_run)
                for world in (3, 5):
                    parts = []
                    for r in range(world):
                        parts.append(m.pairw(r, world))
                        items, run = sdist.strip_plan(N, M // 64, r, world, 1, pairs, tail_run=tail_run,
                                                      n_cus=hip_ctx.get_option("n_cus"), return_run=True)
                        assert hip_ctx.last_launch_info()["items"] == len(items), (N, pairs, world, r, run)
                    assert sum(parts) == want, (N, pairs, tail_run, world, parts, want)
            _reset(hip_ctx)
            m.close()
    finally:
        _reset(hip_ctx)


def test_raw_buffer_wrappers_stream_their_rows_in_panels(hip_ctx, orc):
    """STORM_wrapper_diag[_blocked] (storm.c:132-150, :222-279) get the caller's matrix anew on every call. On one device the
    rows travel in panels of whole tiles while the panels before are multiplied (storm_hip_pairw_dense_upload: the pairs
    whose later row lies in a panel, its tiles stationary). Against the oracle's blocked loop on shapes around the panel
    edges (rows not a multiple of 256, fewer tiles than panels, one word per row, rows below the streaming threshold), and
    with the caller's buffer CHANGED between calls — nothing of the previous call may be reused."""
    for M, N, d in ((4096, 2048, 1500), (4096, 2050, 1500), (1024, 5000, 400), (64, 9000, 20), (8192, 2559, 3000),
                    (700, 4097, 300), (4096, 300, 1500)):
        mat = synth.dense_matrix_c(M, N, d, seed=M + N)
        want = orc.wrapper_diag_blocked(mat, 31)
        assert sb.wrapper_diag(mat) == want, (M, N)
        assert sb.wrapper_diag_blocked(mat, 7) == want
        mat[N // 2] = 0            # the caller edits its buffer in place ...
        mat[5, :] = mat[N - 1, :]
        want2 = orc.wrapper_diag_blocked(mat, 31)
        assert sb.wrapper_diag(mat) == want2, (M, N)
        smaller = np.ascontiguousarray(mat[: N - 300])   # ... and comes back with fewer rows
        assert sb.wrapper_diag(smaller) == orc.wrapper_diag_blocked(smaller, 31)
    # the headline shape: the device-built matrix downloaded, through the wrapper, against the resident pass
    N, M = 10000, 65536
    m = hip_ctx.matrix(N, M // 64)
    m.fill_synthetic(M, M // 2, seed=42)
    host = m.download()
    assert sb.wrapper_diag_blocked(host, 31) == m.pairw() == m.column_identity()
    m.close()


def test_per_pair_matrix_left_in_device_memory(orc):
    """STORM_pairw_matrix_device / STORM_contig_pairw_matrix_device (extensions): the triangle of STORM_pairw_matrix left in
    device memory — entry (i, j), i < j, is what STORM_bitmap_cont_intersect_cardinality returns for rows i and j
    (storm.c:790-814) — against the host-output entry points and the oracle's row-pair function written out, for both
    tile kernels (STORM_hip_set_option: tilebits8_kernel and tilering_kernel), lists, bitmaps and mixed kinds."""
    import torch
    lib = sb.load()
    assert lib.STORM_hip_set_option(b"matrix_lists", 0) == 0   # (the tile kernels on the dense replica are the subject here)
    try:
        for M, N, d in ((200000, 300, 40), (131072, 257, 9000), (65536, 600, 5000)):
            rows = [np.unique(np.random.default_rng(N + i).integers(0, M, size=d if i % 3 else d // 50 + 1)).astype(np.uint32)
                    for i in range(N)]
            s, c = sb.Storm(), sb.StormContig(M)
            for r in rows:
                s.add(r)
                c.add(r)
            want = s.pairw_matrix("and")
            if N <= 300:
                assert np.array_equal(want, orc.storm(rows).pair_counts())
            for shape in (2, 5, 2):
                assert lib.STORM_hip_set_option(b"k2_tile_shape", shape) == 0
                dev = torch.zeros((N + 3, N + 8), dtype=torch.int32, device="cuda:0")
                s.pairw_matrix_device(dev.data_ptr(), N + 3, N + 8)
                got = np.triu(dev.cpu().numpy().astype(np.uint32)[:N, :N], k=1)
                assert np.array_equal(got, want), (M, N, shape)
                dev.zero_()
                c.pairw_matrix_device(dev.data_ptr(), N + 3, N + 8, "xor")
                got = np.triu(dev.cpu().numpy().astype(np.uint32)[:N, :N], k=1)
                assert np.array_equal(got, c.pairw_matrix("xor")), (M, N, shape)
            assert lib.STORM_pairw_matrix_device(s._h, 0, None, N, N) == -2
            assert lib.STORM_pairw_matrix_device(s._h, 0, C.c_void_p(dev.data_ptr()), N - 1, N) == -4
            s.free()
            c.free()
    finally:
        lib.STORM_hip_set_option(b"k2_tile_shape", 0)
        lib.STORM_hip_set_option(b"matrix_lists", -1)


def _last_pass():
    out = (C.c_uint64 * 4)()
    assert sb.load().STORM_hip_last_pass(out) == 0
    return [int(x) for x in out]


def test_per_pair_matrix_of_a_list_only_container_straight_from_its_lists(orc):
    """K5 (storm_hip_lists.hip): STORM_pairw_matrix_device on a container whose blocks are all lists never builds the dense
    replica — lists_hash_kernel (rows of up to ~190 positions) or lists_matrix_kernel (windows) join the rows' positions
    per output tile in the LDS. Entry (i, j), i < j, must be what STORM_bitmap_cont_intersect_cardinality returns for rows
    i and j (storm.c:790-814: block-id merge :75-106, two lists meet in STORM_intersect_vector16_cardinality :4-73), OR / XOR
    from the row lengths; entries i >= j stay untouched. Both kernels forced in turn, then the automatic choice; shapes with
    empty rows, a ragged last group, one position per row, rows in far-apart blocks, 2^22 positions; a container that grows
    between calls; a container with a bitmap block (not eligible: the tile kernels run)."""
    import torch
    lib = sb.load()
    RAN_LISTS, RAN_TILES = 64, 128
    try:
        for M, N, d in ((65536, 2, 5), (65536, 65, 40), (200000, 300, 60), (524288, 257, 524), (1 << 22, 130, 300),
                        (8193, 321, 7), (70000, 1000, 1), (3000000, 513, 150)):
            rows = synth.positions(M, N, d, seed=N + d)
            rows[N // 3] = rows[N // 3][:0]
            s = sb.Storm()
            for r in rows:
                s.add(r)
            want = orc.storm(rows).pair_counts().astype(np.int64)
            lens = np.array([len(r) for r in rows], dtype=np.int64)
            upper = np.triu(np.ones((N, N), dtype=bool), k=1)
            for op in ("and", "or", "xor"):
                ref = {"and": want, "or": lens[:, None] + lens[None, :] - want,
                       "xor": lens[:, None] + lens[None, :] - 2 * want}[op]
                for lists, kernel in ((1, 1), (1, 2), (-1, 0), (0, 0)):
                    assert lib.STORM_hip_set_option(b"matrix_lists", lists) == 0
                    assert lib.STORM_hip_set_option(b"matrix_lists_kernel", kernel) == 0
                    dev = torch.full((N + 1, N + 5), -7, dtype=torch.int32, device="cuda:0")
                    s.pairw_matrix_device(dev.data_ptr(), N + 1, N + 5, op)
                    got = dev.cpu().numpy()[:N, :N]
                    assert np.array_equal(got[upper], ref[upper]), (M, N, d, op, lists, kernel)
                    assert (got[~upper] == -7).all() and (dev.cpu().numpy()[N:] == -7).all(), (M, N, d, op, lists, kernel)
                    ran = _last_pass()
                    assert ran[0] == (RAN_TILES if lists == 0 else RAN_LISTS), (ran, lists, kernel)
                    if lists == 1:   # (the hash kernel's groups shrink with the row length; the window kernel's are 64 rows)
                        assert ran[3] == 64 if kernel == 1 else ran[3] in (8, 16, 32, 64), ran
                    # the host-output entry point takes the same path (whole rows: zeros at i >= j)
                    host = s.pairw_matrix(op)
                    assert np.array_equal(host.astype(np.int64), np.triu(ref, k=1)), (M, N, d, op, lists, kernel)
                    assert _last_pass()[0] == (RAN_TILES if lists == 0 else RAN_LISTS)
            # the container grows: the lists are rebuilt from the container as it is now
            assert lib.STORM_hip_set_option(b"matrix_lists", 1) == 0
            extra = synth.positions(M, 3, d, seed=7)
            for r in extra:
                s.add(r)
            rows2 = list(rows) + list(extra)
            N2 = N + 3
            dev = torch.zeros((N2, N2), dtype=torch.int32, device="cuda:0")
            s.pairw_matrix_device(dev.data_ptr(), N2, N2)
            assert np.array_equal(np.triu(dev.cpu().numpy(), k=1), np.triu(orc.storm(rows2).pair_counts().astype(np.int32), k=1)), (M, N)
            assert _last_pass()[0] == RAN_LISTS
            s.free()
        # one dense row among the lists (a bitmap block): not eligible, whatever the option says
        rows = synth.positions(65536, 100, 50, seed=3)
        rows[17] = np.arange(0, 65536, 3, dtype=np.uint32)
        s = sb.Storm()
        for r in rows:
            s.add(r)
        dev = torch.zeros((100, 100), dtype=torch.int32, device="cuda:0")
        s.pairw_matrix_device(dev.data_ptr(), 100, 100)
        assert _last_pass()[0] == RAN_TILES
        assert np.array_equal(np.triu(dev.cpu().numpy(), k=1), np.triu(orc.storm(rows).pair_counts().astype(np.int32), k=1))
        s.free()
    finally:
        lib.STORM_hip_set_option(b"matrix_lists", -1)
        lib.STORM_hip_set_option(b"matrix_lists_kernel", 0)


def test_lists_path_at_the_readme_storm_shape_sums_to_the_all_pairs_total():
    """BASELINE c4's shape (N = 10000, M = 524288) at its two sparsest loads: the matrix written from the lists must add up to
    STORM_pairw_intersect_cardinality of the same handle (a different kernel family: the list-probe kernel), for the kernel the
    automatic rule picks and for the other one; a sampled window of it equals the dense replica's tile kernels."""
    import torch
    lib = sb.load()
    N, M = 10000, 524288
    try:
        dev = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
        ref = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
        for d in (104, 524):
            s = sb.Storm()
            assert s.add_synthetic(M, N, d, seed=42) == N
            total = s.pairw_intersect_cardinality()
            for lists, kernel in ((-1, 0), (1, 1), (1, 2)):
                assert lib.STORM_hip_set_option(b"matrix_lists", lists) == 0
                assert lib.STORM_hip_set_option(b"matrix_lists_kernel", kernel) == 0
                dev.zero_()
                s.pairw_matrix_device(dev.data_ptr(), N, N)
                assert _last_pass()[0] == 64
                assert int(dev.to(torch.int64).sum().item()) == total, (d, lists, kernel)
            assert lib.STORM_hip_set_option(b"matrix_lists", 0) == 0
            s.pairw_matrix_device(ref.data_ptr(), N, N)
            assert bool((dev == ref).all().item()), d
            s.free()
    finally:
        lib.STORM_hip_set_option(b"matrix_lists", -1)
        lib.STORM_hip_set_option(b"matrix_lists_kernel", 0)


def test_ring_tile_kernel_against_the_oracle_and_the_default_kernel(hip_ctx, orc):
    """tilering_kernel (k2_tile_shape = 5: both operands as FP4 images in the LDS, 16x16x128 MFMAs, SIMD partners half a stage
    apart; barrier and counter synchronisation) against the oracle's per-pair counts (storm.c:1199-1238 with the leaf's result
    kept per pair) and tilebits8_kernel: triangle, AND / OR / XOR, rectangle, bands, rows of zero, ragged row blocks (1 .. 255
    rows beyond a multiple of 256), row lengths that are not whole 512-bit chunks or whole chunk PAIRS, one tile, k-split last
    rounds (more tiles than CUs)."""
    import torch
    try:
        for M, N, d in ((4096, 256, 2048), (640, 65, 200), (1000, 257, 300), (9000, 700, 3000), (300, 130, 100), (512, 300, 100),
                        (520, 300, 100), (70000, 1029, 20000), (1536, 4700, 500)):
            mat = synth.dense_matrix_c(M, N, d, seed=N + M)
            mat[N // 3] = 0
            m = hip_ctx.matrix_from_host(mat)
            for op in ("and", "or", "xor"):
                hip_ctx.set_option("k2_tile_shape", 2)
                ref = m.pairw_matrix(op)
                hip_ctx.set_option("k2_tile_shape", 5)
                for sync in (0, 1):
                    hip_ctx.set_option("k2_ring_sync", sync)
                    got = m.pairw_matrix(op)
                    assert np.array_equal(ref, got), (M, N, op, sync, np.argwhere(ref != got)[:3].tolist())
                hip_ctx.set_option("k2_ring_sync", 0)
                # the k-parts into windows of their own + reduce_parts_kernel instead of atomics (option k2_matrix_parts), both kernels
                hip_ctx.set_option("k2_matrix_parts", 1)
                for shape in (5, 2):
                    hip_ctx.set_option("k2_tile_shape", shape)
                    got = m.pairw_matrix(op)
                    assert np.array_equal(ref, got), (M, N, op, "windows", shape, np.argwhere(ref != got)[:3].tolist())
                hip_ctx.set_option("k2_matrix_parts", 0)
                hip_ctx.set_option("k2_tile_shape", 5)
                if N <= 300 and op == "and":
                    assert np.array_equal(got, np.triu(orc.tile_counts(mat, 0, N, 0, N), k=1).astype(np.uint32))
            if N >= 600:   # a band of the triangle into device memory, and the rectangle of the two halves
                hip_ctx.set_option("k2_tile_shape", 5)
                band = torch.zeros((300, N), dtype=torch.int32, device="cuda:0")
                m.pairw_matrix_band_device(band.data_ptr(), N, 256, 300)
                assert np.array_equal(np.triu(band.cpu().numpy().astype(np.uint32), k=257)[:, :],
                                      np.triu(ref_and(m, hip_ctx)[256:556], k=257))
                na = N // 2
                ma, mb = hip_ctx.matrix_from_host(mat[:na]), hip_ctx.matrix_from_host(mat[na:])
                got = ma.square_matrix(mb, "and")
                hip_ctx.set_option("k2_tile_shape", 2)
                assert np.array_equal(got, ma.square_matrix(mb, "and"))
                if N <= 1100:
                    assert np.array_equal(got, orc.tile_counts(mat, 0, na, na, N).astype(np.uint32))
                ma.close()
                mb.close()
            m.close()
    finally:
        hip_ctx.set_option("k2_tile_shape", 0)
        hip_ctx.set_option("k2_ring_sync", 0)
        hip_ctx.set_option("k2_matrix_parts", 0)


def ref_and(m, ctx):
    ctx.set_option("k2_tile_shape", 2)
    out = m.pairw_matrix("and")
    ctx.set_option("k2_tile_shape", 5)
    return out
