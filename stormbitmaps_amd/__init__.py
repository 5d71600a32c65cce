"""stormbitmaps_amd — MI355X-native all-pairs AND+popcount (XX^T upper triangle) behind the
StormBitmaps ``storm.h`` API.

Layout
    csrc/            hand-written gfx950 HIP kernels, the C-ABI shim (include/storm_hip.h) and
                     the C host side of the storm.h containers  -> libstorm_hip.so
    _lib.py          ctypes loader (fails loudly when the library or the GPU is missing)
    api.py           Python mirror of the storm.h interface + thin device-level handles
    synth.py         deterministic synthetic inputs (splitmix64), numpy restatement
    dist.py          one-process-per-GPU sharding + RCCL all-reduce of the 8-byte total
"""
from .api import (HipContext, HipMatrix, Storm, StormContig, wrapper_diag,  # noqa: F401
                  wrapper_diag_blocked, wrapper_square)
from ._lib import StormHipError, load  # noqa: F401

__all__ = ["HipContext", "HipMatrix", "Storm", "StormContig", "StormHipError", "load",
           "wrapper_diag", "wrapper_diag_blocked", "wrapper_square"]
