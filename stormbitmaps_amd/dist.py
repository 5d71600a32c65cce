"""One process per GPU: shard the pair space, all-reduce the 8-byte total.

The pair space N(N-1)/2 is cut into segments (A block x run of B rows) and rank r of G takes
every G-th segment (storm_hip_pairw_dense's shard arguments). X is replicated on every GPU,
so the only inter-GPU exchange is one uint64 sum: ``torch.distributed.all_reduce`` on a
1-element int64 tensor — backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU
tests. Integer addition makes the result independent of G.
"""
from __future__ import annotations

import os
from typing import Callable, Tuple


def rank_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process default)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend: str):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        rank, world, _ = rank_world()
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def shard_segments(n_segments: int, rank: int, world: int):
    """Indices of the segments rank `rank` owns (cyclic, as ensure_segments() in storm_hip.hip)."""
    return range(rank, n_segments, world)


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
