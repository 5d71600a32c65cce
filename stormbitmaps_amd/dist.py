"""One process per GPU: shard the all-pairs work, all-reduce the 8-byte total.

X is replicated on every GPU and rank r of G multiplies its share of the work of the default
(matrix-core strip) path — ``storm_hip_pairw_dense(..., shard_rank, shard_count)``. The share is
two-level (``storm_hip_strip_plan`` in include/storm_hip.h):

  * whole k-slices (256 bits of every row), in units of 4 (one 128-byte line of the bit matrix):
    slice ks goes to rank (ks / 4) % G, so a rank expands and multiplies 1/G of the columns against
    the whole pair space;
  * the leftover slices (fewer than 4 G) are cut along the PAIR space (A tile x run of B blocks),
    longest item first onto the least loaded rank.

The only inter-GPU exchange is one uint64 sum: ``torch.distributed.all_reduce`` on a 1-element
int64 tensor — backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
Integer addition makes the result independent of G. (The popcount fallback kernel, used only for
rows beyond the strips' reach, shards its segment list cyclically instead.)
"""
from __future__ import annotations

import os
from typing import Callable, Tuple


def rank_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process default)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def free_port() -> int:
    """A TCP port nobody listens on right now (fixed ports collide with concurrent jobs on one
    host and with sockets still in TIME_WAIT)."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def init_process_group(backend: str):
    import torch.distributed as dist
    if not dist.is_initialized():
        rank, world, _ = rank_world()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # Only a 1-process group can pick its own port (nobody else has to find it); a
            # launcher (torch.distributed.run, bench.py's own) always exports one for world > 1.
            if world > 1:
                raise RuntimeError("MASTER_PORT is not set: start the ranks with torch.distributed.run "
                                   "or `python bench.py --gpus N`, which pick a free rendezvous port")
            os.environ["MASTER_PORT"] = str(free_port())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def strip_plan(n_rows: int, n_words: int, rank: int, world: int, form: int = 1, pair_space: int = 0, *,
               max_run: int = 0, tail_run: int = 32, tail_slices: int = 3, lpt_rounds: int = 6, n_cus: int = 256,
               return_run: bool = False):
    """The work items rank `rank` of `world` multiplies, as an [n, 5] uint32 array of {a_row0, diag, j0, j1, ks}
    (see storm_hip_strip_plan3 in include/storm_hip.h). form 1 = the default path (K2b: slice ks = class pair
    ks & 1 of the 512-bit chunk ks / 2), form 0 = the FP4-shadow strips (slice ks = 256 consecutive bits);
    pair_space 1 = every slice cut along the pair space (option k2_shard_pairs). The keyword options are the
    context options of the same names (defaults = a fresh context on a 256-CU device): the list returned is the
    list such a context launches — one function derives the shaping for both. max_run 0 = automatic, the same
    for every rank of a world; return_run=True also returns the run length chosen.
    Host-only: computed by libstorm_hip.so without touching a device."""
    import ctypes as C

    import numpy as np

    from . import _lib
    lib = _lib.load()
    n = C.c_uint64(0)
    run = C.c_int(0)
    args = (n_rows, n_words, rank, world, form, pair_space, max_run, tail_run, tail_slices, lpt_rounds, n_cus)
    _lib.check(lib.storm_hip_strip_plan3(*args, None, 0, C.byref(n), C.byref(run)), "storm_hip_strip_plan3")
    out = np.zeros((int(n.value), 5), dtype=np.uint32)
    if n.value:
        _lib.check(lib.storm_hip_strip_plan3(*args, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n), C.byref(run)),
                   "storm_hip_strip_plan3")
    return (out, int(run.value)) if return_run else out


def slice_columns(mat, ks: int, form: int = 1):
    """The bits of k-slice `ks` of a bit matrix [rows, words] as a matrix of the same row count (what an item of
    strip_plan(form) multiplies): form 0 = words [4 ks, 4 ks + 4); form 1 = the words [8 c, 8 c + 8) of chunk
    c = ks // 2 masked to the class pair ks & 1 (bits b with (b % 4) // 2 == ks & 1)."""
    import numpy as np
    if form == 0:
        return np.ascontiguousarray(mat[:, 4 * ks:4 * ks + 4])
    c = ks // 2
    mask = np.uint64(0xCCCCCCCCCCCCCCCC if ks & 1 else 0x3333333333333333)
    return np.ascontiguousarray(mat[:, 8 * c:8 * c + 8] & mask)


def stream_plan(n_rows: int, n_words: int, rank: int = 0, world: int = 1, n_cus: int = 256):
    """The segments rank `rank` of `world` walks on the one-launch stage stream (K2q, matrices of up to 8192
    rows), as an [n, 8] uint32 array of {workgroup, a_blk, ks, b_first, n_b, range_nb, diag, stages} (see
    storm_hip_stream_plan in include/storm_hip.h), and the number of workgroups. Host-only."""
    import ctypes as C

    import numpy as np

    from . import _lib
    lib = _lib.load()
    n, g = C.c_uint64(0), C.c_uint32(0)
    _lib.check(lib.storm_hip_stream_plan(n_rows, n_words, rank, world, n_cus, None, 0, C.byref(n), C.byref(g)),
               "storm_hip_stream_plan")
    out = np.zeros((int(n.value), 8), dtype=np.uint32)
    if n.value:
        _lib.check(lib.storm_hip_stream_plan(n_rows, n_words, rank, world, n_cus, out.ctypes.data_as(C.c_void_p),
                                             n.value, C.byref(n), C.byref(g)), "storm_hip_stream_plan")
    return out, int(g.value)


def matrix_plan(n_rows_a: int, n_words: int, n_rows_b: int = 0, band_row0: int = 0, band_rows: int = 0, n_cus: int = 256,
                slots_per_cu: int = 0, min_chunks: int = 8, diag_cost_pct: int = 80):
    """The items the materialised-output kernel for matrices of few tiles (K2h) launches, as an [n, 8] uint32 array of
    {I, J, first chunk, chunks, tile, part, n_parts, narrow} (see storm_hip_matrix_plan in include/storm_hip.h). Host-only."""
    import ctypes as C

    import numpy as np

    from . import _lib
    lib = _lib.load()
    n = C.c_uint64(0)
    args = (n_rows_a, n_rows_b, n_words, band_row0, band_rows, n_cus, slots_per_cu, min_chunks, diag_cost_pct)
    _lib.check(lib.storm_hip_matrix_plan(*args, None, 0, C.byref(n)), "storm_hip_matrix_plan")
    out = np.zeros((int(n.value), 8), dtype=np.uint32)
    if n.value:
        _lib.check(lib.storm_hip_matrix_plan(*args, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)),
                   "storm_hip_matrix_plan")
    return out


def allreduce_total(partial: int, device=None) -> int:
    """Sum the per-rank partial totals. Totals are < 2^63 for every supported shape
    (N^2/2 * M < 2^63), so the int64 transport is exact."""
    import torch
    import torch.distributed as dist
    if partial >= 1 << 63:
        raise OverflowError("partial total does not fit the int64 all-reduce transport")
    t = torch.tensor([partial], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def sharded_pairw(partial_fn: Callable[[int, int], int], device=None) -> int:
    """partial_fn(rank, world) -> this rank's partial; returns the all-reduced total."""
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    return allreduce_total(partial_fn(rank, world), device=device)
