"""Deterministic synthetic inputs — numpy restatement of include/storm_synth.h.

Per row: ``draws`` values uniform on [0, M) with replacement, distinct values kept, sorted —
the recipe of the reference harness (benchmark.cpp:762-772) with a reproducible counter-based
splitmix64 stream instead of std::random_device.
"""
from __future__ import annotations

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def draws_for_rows(n_bits: int, row0: int, n_rows: int, draws: int, seed: int) -> np.ndarray:
    """[n_rows, draws] uint32 positions, in draw order (duplicates included)."""
    with np.errstate(over="ignore"):
        n = (np.arange(row0, row0 + n_rows, dtype=np.uint64)[:, None] * np.uint64(draws)
             + np.arange(1, draws + 1, dtype=np.uint64)[None, :])
        z = _mix(np.uint64(seed) + n * GOLDEN)
        hi, lo = z >> np.uint64(32), z & np.uint64(0xFFFFFFFF)
        m = np.uint64(n_bits)
        pos = (hi * m + ((lo * m) >> np.uint64(32))) >> np.uint64(32)  # (z * M) >> 64
    return pos.astype(np.uint32)


def dense_matrix(n_bits: int, n_rows: int, draws: int, seed: int = 42, row0: int = 0,
                 chunk_rows: int = 128) -> np.ndarray:
    """[n_rows, ceil(M/64)] uint64 bitmap rows (bit v -> word v//64, bit v%64)."""
    n_words = (n_bits + 63) // 64
    out = np.zeros((n_rows, n_words), dtype=np.uint64)
    if draws == 0:
        return out
    for r0 in range(0, n_rows, chunk_rows):
        nr = min(chunk_rows, n_rows - r0)
        pos = draws_for_rows(n_bits, row0 + r0, nr, draws, seed)
        bits = np.zeros((nr, n_words * 64), dtype=np.uint8)
        bits[np.arange(nr)[:, None], pos] = 1
        out[r0:r0 + nr] = np.packbits(bits, axis=1, bitorder="little").view(np.uint64)
    return out


def dense_matrix_c(n_bits: int, n_rows: int, draws: int, seed: int = 42,
                   row0: int = 0) -> np.ndarray:
    """Same matrix from the C generator in libstorm_hip.so (storm_synth_fill_dense) — fast
    enough for the full benchmark shapes; tests/test_synth.py checks it equals dense_matrix."""
    from . import _lib
    lib = _lib.load()
    n_words = (n_bits + 63) // 64
    out = np.zeros((n_rows, n_words), dtype=np.uint64)
    if n_rows:
        lib.storm_synth_fill_dense(out.ctypes.data, n_words, n_bits, row0, n_rows, draws, seed)
    return out


def positions(n_bits: int, n_rows: int, draws: int, seed: int = 42, row0: int = 0):
    """List of sorted distinct uint32 position arrays, one per row."""
    rows = []
    for r0 in range(0, n_rows, 256):
        nr = min(256, n_rows - r0)
        if draws == 0:
            rows.extend(np.zeros(0, dtype=np.uint32) for _ in range(nr))
            continue
        pos = draws_for_rows(n_bits, row0 + r0, nr, draws, seed)
        rows.extend(np.unique(pos[i]) for i in range(nr))
    return rows


def positions_from_dense(mat: np.ndarray):
    """Sorted set-bit positions of every row of a dense uint64 matrix."""
    bits = np.unpackbits(mat.view(np.uint8), axis=1, bitorder="little")
    return [np.flatnonzero(bits[i]).astype(np.uint32) for i in range(mat.shape[0])]
