"""The multi-rank path on CPU (gloo, world_size 2): shards partition the segment list and the
all-reduced partial totals equal the single-rank total. The per-segment partials here come
from the ORACLE (checker) because the product kernels need a GPU; what is under test is the
host-side sharding + all-reduce plumbing of stormbitmaps_amd/dist.py, which bench.py uses
unchanged with backend "nccl" (RCCL)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from stormbitmaps_amd import dist as sdist
from stormbitmaps_amd import synth

A_BLOCK, SEG_ROWS = 128, 256


def segments(n_rows):
    """Same construction as ensure_segments() in csrc/storm_hip.hip: full segments, then diagonals."""
    full, diag = [], []
    for a0 in range(0, n_rows, A_BLOCK):
        a_end = min(a0 + A_BLOCK, n_rows)
        if a_end - a0 > 1:
            diag.append((a0, a_end, a0, a_end))
        for j in range(a0 + A_BLOCK, n_rows, SEG_ROWS):
            full.append((a0, a_end, j, min(j + SEG_ROWS, n_rows)))
    return full, diag


def _segment_total(orc, mat, seg):
    a0, a_end, j_lo, j_hi = seg
    t = orc.tile_counts(mat, a0, a_end, j_lo, j_hi)
    if j_lo == a0:
        t = np.triu(t, k=1)
    return int(t.sum())


def _worker(rank, world, port, n_rows, want, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tests._orc import Oracle
    orc = Oracle()
    d = sdist.init_process_group("gloo")
    mat = synth.dense_matrix(2048, n_rows, 700, seed=42)
    full, diag = segments(n_rows)

    def partial(r, w):
        mine = [full[i] for i in sdist.shard_segments(len(full), r, w)] + \
               [diag[i] for i in sdist.shard_segments(len(diag), r, w)]
        return sum(_segment_total(orc, mat, s) for s in mine)

    total = sdist.sharded_pairw(partial)
    q.put((rank, total, partial(rank, world)))
    d.barrier()
    d.destroy_process_group()
    assert total == want


def test_two_rank_gloo_allreduce_equals_single_rank(orc):
    n_rows = 700
    mat = synth.dense_matrix(2048, n_rows, 700, seed=42)
    want = orc.wrapper_diag(mat)
    full, diag = segments(n_rows)
    # the shards partition the segment list for any world size
    for w in (1, 2, 3, 8):
        for lst in (full, diag):
            got = sorted(i for r in range(w) for i in sdist.shard_segments(len(lst), r, w))
            assert got == list(range(len(lst)))
    # ...and the segments tile the strict upper triangle exactly once
    cover = np.zeros((n_rows, n_rows), dtype=np.int32)
    for a0, a_end, j_lo, j_hi in full + diag:
        if j_lo == a0:
            cover[a0:a_end, j_lo:j_hi] += np.triu(np.ones((a_end - a0, j_hi - j_lo), dtype=np.int32), k=1)
        else:
            cover[a0:a_end, j_lo:j_hi] += 1
    assert np.array_equal(cover, np.triu(np.ones((n_rows, n_rows), dtype=np.int32), k=1))

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rows, want, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == want
    assert res[0][2] + res[1][2] == want and res[0][2] != want
