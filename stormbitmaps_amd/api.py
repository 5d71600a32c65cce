"""Python mirror of the storm.h interface (same names, argument meaning and error behaviour as
the reference's C API) plus thin handles over the device-level C-ABI of include/storm_hip.h.

Everything here calls into libstorm_hip.so; no arithmetic happens in Python.
Reference lines cited are /root/reference/storm.h and storm.c of mklarqvist/StormBitmaps.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import StormHipError, check

ALL_PAIRS_FAILED = (1 << 64) - 1  # (uint64_t)-1, storm.c:878,898,1150,1176


def _u32(values) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(values, dtype=np.uint32))


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _all_pairs(value: int, what: str) -> int:
    if value == ALL_PAIRS_FAILED:
        lib = _lib.load()
        msg = lib.STORM_hip_error()
        raise StormHipError(f"{what}: device path failed: {msg.decode() if msg else '?'}")
    return int(value)


# ------------------------------------------------------------------------------------------
# storm.h containers
# ------------------------------------------------------------------------------------------
class StormContig:
    """STORM_contiguous_t (storm.h:188-200, :235-242): dense row-major bitmap matrix."""

    def __init__(self, vector_length: int):
        self._lib = _lib.load()
        self._h = self._lib.STORM_contig_new(vector_length)  # storm.c:1001
        if not self._h:
            raise MemoryError("STORM_contig_new")
        self.vector_length = vector_length

    def add(self, values) -> int:
        """STORM_contig_add (storm.c:1031): sorted positions of one row; returns n_values,
        0 for an empty row (no row appended)."""
        v = _u32(values)
        return int(self._lib.STORM_contig_add(self._h, _ptr(v) if v.size else _ptr(np.zeros(1, np.uint32)),
                                              v.size))

    def add_synthetic(self, n_rows: int, draws: int, seed: int = 42, row0: int = 0) -> int:
        """Rows of the deterministic benchmark matrix (include/storm_synth.h), added in C."""
        return int(self._lib.storm_synth_fill_contig(self._h, self.vector_length, row0, n_rows,
                                                     draws, seed))

    def clear(self) -> int:
        return int(self._lib.STORM_contig_clear(self._h))  # storm.c:1139

    def pairw_intersect_cardinality(self) -> int:
        return _all_pairs(self._lib.STORM_contig_pairw_intersect_cardinality(self._h),
                          "STORM_contig_pairw_intersect_cardinality")  # storm.c:1149

    def pairw_intersect_cardinality_blocked(self, bsize: int = 0) -> int:
        return _all_pairs(
            self._lib.STORM_contig_pairw_intersect_cardinality_blocked(self._h, bsize),
            "STORM_contig_pairw_intersect_cardinality_blocked")  # storm.c:1175

    @property
    def n_rows(self) -> int:
        """Rows the handle holds (STORM_contig_n_rows; an empty add appends none, storm.c:1034)."""
        return int(self._lib.STORM_contig_n_rows(self._h))

    def pairw_matrix(self, op: str = "and") -> np.ndarray:
        """STORM_contig_pairw_matrix (extension): [n_rows, n_rows] uint32 per-pair counts, i < j.
        The buffer is sized from the handle's own row count and its extent is passed down, so a
        miscount cannot overrun it (the C entry point returns -4 instead)."""
        n = self.n_rows
        out = np.zeros((n, n), dtype=np.uint32)
        rc = int(self._lib.STORM_contig_pairw_matrix(self._h, {"and": 0, "or": 1, "xor": 2}[op],
                                                     _ptr(out), n, n))
        if rc != 0:
            raise RuntimeError(f"STORM_contig_pairw_matrix -> {rc}: "
                               f"{self._lib.STORM_hip_error().decode()}")
        return out

    def pairw_matrix_device(self, d_out: int, out_rows: int, out_ld: int, op: str = "and") -> None:
        """STORM_contig_pairw_matrix_device (extension): the same triangle left in device memory at address d_out."""
        rc = int(self._lib.STORM_contig_pairw_matrix_device(self._h, {"and": 0, "or": 1, "xor": 2}[op], C.c_void_p(d_out),
                                                            out_rows, out_ld))
        if rc != 0:
            raise RuntimeError(f"STORM_contig_pairw_matrix_device -> {rc}: {self._lib.STORM_hip_error().decode()}")

    def hip_invalidate(self) -> None:
        """STORM_contig_hip_invalidate: forget the device copy after an in-place edit of the
        handle's public buffers (storm.h extension)."""
        self._lib.STORM_contig_hip_invalidate(self._h)

    def pairw_intersect_cardinality_list(self) -> int:
        return int(self._lib.STORM_contig_pairw_intersect_cardinality_list(self._h))  # :1243

    def pairw_intersect_cardinality_blocked_list(self, bsize: int = 0) -> int:
        return int(self._lib.STORM_contig_pairw_intersect_cardinality_blocked_list(self._h,
                                                                                  bsize))  # :1265

    def free(self) -> None:
        if self._h:
            self._lib.STORM_contig_free(self._h)  # storm.c:1020
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Storm:
    """STORM_t (storm.h:175-178, :225-232): rows of 65536-bit blocks, list or bitmap kind."""

    def __init__(self):
        self._lib = _lib.load()
        self._h = self._lib.STORM_new()  # storm.c:827
        if not self._h:
            raise MemoryError("STORM_new")

    def add(self, values) -> int:
        v = _u32(values)
        return int(self._lib.STORM_add(self._h, _ptr(v) if v.size else _ptr(np.zeros(1, np.uint32)),
                                       v.size))  # storm.c:844

    def add_synthetic(self, n_bits: int, n_rows: int, draws: int, seed: int = 42,
                      row0: int = 0) -> int:
        return int(self._lib.storm_synth_fill_storm(self._h, n_bits, row0, n_rows, draws, seed))

    def clear(self) -> int:
        return int(self._lib.STORM_clear(self._h))  # storm.c:868

    @property
    def n_rows(self) -> int:
        """Rows added so far (STORM_n_rows)."""
        return int(self._lib.STORM_n_rows(self._h))

    def pairw_matrix(self, op: str = "and") -> np.ndarray:
        """STORM_pairw_matrix (extension): [n_rows, n_rows] uint32, entry (i, j), i < j = what
        STORM_bitmap_cont_intersect_cardinality gives for rows i and j (storm.c:790-814); "or" / "xor": the union /
        symmetric-difference counts."""
        n = self.n_rows
        out = np.zeros((n, n), dtype=np.uint32)
        rc = int(self._lib.STORM_pairw_matrix(self._h, {"and": 0, "or": 1, "xor": 2}[op], _ptr(out), n, n))
        if rc != 0:
            raise RuntimeError(f"STORM_pairw_matrix -> {rc}: {self._lib.STORM_hip_error().decode()}")
        return out

    def pairw_matrix_device(self, d_out: int, out_rows: int, out_ld: int, op: str = "and") -> None:
        """STORM_pairw_matrix_device (extension): the same triangle left in device memory at address d_out."""
        rc = int(self._lib.STORM_pairw_matrix_device(self._h, {"and": 0, "or": 1, "xor": 2}[op], C.c_void_p(d_out),
                                                     out_rows, out_ld))
        if rc != 0:
            raise RuntimeError(f"STORM_pairw_matrix_device -> {rc}: {self._lib.STORM_hip_error().decode()}")

    def serialized_size(self) -> int:
        return int(self._lib.STORM_serialized_size(self._h))  # storm.c:963

    def hip_invalidate(self) -> None:
        """STORM_hip_invalidate (storm.h extension)."""
        self._lib.STORM_hip_invalidate(self._h)

    def serialize(self) -> np.ndarray:
        """STORM_serialize: exactly STORM_serialized_size(h) bytes (uint8 array)."""
        n = self.serialized_size()
        buf = np.zeros(n + (n & 1), dtype=np.uint8)
        got = int(self._lib.STORM_serialize(self._h, _ptr(buf), n))
        if got != n:
            raise RuntimeError(f"STORM_serialize wrote {got} of {n} bytes")
        return buf[:n]

    @classmethod
    def deserialize(cls, data) -> "Storm":
        """STORM_deserialize; ValueError on a malformed stream."""
        buf = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._h = self._lib.STORM_deserialize(_ptr(buf), buf.size)
        if not self._h:
            raise ValueError("STORM_deserialize: malformed stream")
        return self

    @staticmethod
    def serialized_pairw_intersect_cardinality(data) -> int:
        """All-pairs total of a serialized container, arena built on the device from the bytes."""
        buf = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        lib = _lib.load()
        return _all_pairs(lib.STORM_serialized_pairw_intersect_cardinality(_ptr(buf), buf.size),
                          "STORM_serialized_pairw_intersect_cardinality")

    def pairw_intersect_cardinality(self) -> int:
        return _all_pairs(self._lib.STORM_pairw_intersect_cardinality(self._h),
                          "STORM_pairw_intersect_cardinality")  # storm.c:877

    def pairw_intersect_cardinality_blocked(self, bsize: int = 0) -> int:
        return _all_pairs(self._lib.STORM_pairw_intersect_cardinality_blocked(self._h, bsize),
                          "STORM_pairw_intersect_cardinality_blocked")  # storm.c:897

    def free(self) -> None:
        if self._h:
            self._lib.STORM_free(self._h)  # storm.c:836
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def wrapper_diag(vals: np.ndarray) -> int:
    """STORM_wrapper_diag (storm.c:132): all pairs of the rows of a host uint64 matrix."""
    lib = _lib.load()
    v = np.ascontiguousarray(vals, dtype=np.uint64)
    return _all_pairs(lib.STORM_wrapper_diag(v.shape[0], _ptr(v), v.shape[1], None),
                      "STORM_wrapper_diag")


def wrapper_diag_blocked(vals: np.ndarray, block_size: int = 0) -> int:
    """STORM_wrapper_diag_blocked (storm.c:222)."""
    lib = _lib.load()
    v = np.ascontiguousarray(vals, dtype=np.uint64)
    return _all_pairs(
        lib.STORM_wrapper_diag_blocked(v.shape[0], _ptr(v), v.shape[1], None, block_size),
        "STORM_wrapper_diag_blocked")


def wrapper_square(vals1: np.ndarray, vals2: np.ndarray) -> int:
    """STORM_wrapper_square (storm.c:153): every row of vals1 against every row of vals2."""
    lib = _lib.load()
    a = np.ascontiguousarray(vals1, dtype=np.uint64)
    b = np.ascontiguousarray(vals2, dtype=np.uint64)
    if a.shape[1] != b.shape[1]:
        raise ValueError("row widths differ")
    return _all_pairs(lib.STORM_wrapper_square(a.shape[0], _ptr(a), b.shape[0], _ptr(b),
                                               a.shape[1], None), "STORM_wrapper_square")


# ------------------------------------------------------------------------------------------
# device-level handles (include/storm_hip.h)
# ------------------------------------------------------------------------------------------
class HipContext:
    """storm_hip_ctx_t: one MI355X + stream + workspace."""

    def __init__(self, device: int = 0, stream: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.storm_hip_ctx_create(device, C.c_void_p(stream), C.byref(h)),
              "storm_hip_ctx_create")
        self._h = h
        self.device = device

    def set_stream(self, stream: int) -> None:
        check(self._lib.storm_hip_ctx_set_stream(self._h, C.c_void_p(stream)),
              "storm_hip_ctx_set_stream")

    def synchronize(self) -> None:
        check(self._lib.storm_hip_ctx_synchronize(self._h), "storm_hip_ctx_synchronize")

    def set_option(self, key: str, value: int) -> None:
        check(self._lib.storm_hip_ctx_set_option(self._h, key.encode(), value),
              f"storm_hip_ctx_set_option({key})")

    def get_option(self, key: str) -> int:
        return int(self._lib.storm_hip_ctx_get_option(self._h, key.encode()))

    def kernel_time(self):
        """(summed ms, launches) of the dominant kernel since the last call; needs the
        "time_kernels" option (storm_hip.h)."""
        ms, n = C.c_double(0), C.c_uint64(0)
        check(self._lib.storm_hip_kernel_time(self._h, C.byref(ms), C.byref(n)),
              "storm_hip_kernel_time")
        return float(ms.value), int(n.value)

    def last_launch_info(self) -> dict:
        out = (C.c_uint64 * 4)()
        check(self._lib.storm_hip_last_launch_info(self._h, C.byref(out)),
              "storm_hip_last_launch_info")
        return {"items": out[0], "chunks_per_item": out[1], "word_pairs_executed": out[2],
                "segments": out[3]}

    def last_pass_report(self) -> dict:
        """What the last all-pairs pass ran (storm_hip.h: storm_hip_last_pass_report)."""
        out = (C.c_uint64 * 4)()
        check(self._lib.storm_hip_last_pass_report(self._h, C.byref(out)), "storm_hip_last_pass_report")
        names = {1: "pairw_dense_kernel", 2: "pairw_fp4_kernel", 4: "strip16_fp4_kernel", 8: "bitstream_kernel",
                 16: "strip16_bits_kernel", 32: "probe_lists_kernel"}
        return {"kernels": [n for b, n in names.items() if out[0] & b], "dense_word_pairs": int(out[1]),
                "probe_lookups": int(out[2]), "rows_per_lookup": int(out[3])}

    def matrix(self, n_rows: int, n_words: int) -> "HipMatrix":
        return HipMatrix(self, n_rows, n_words)

    def matrix_from_host(self, vals: np.ndarray) -> "HipMatrix":
        v = np.ascontiguousarray(vals, dtype=np.uint64)
        m = HipMatrix(self, v.shape[0], v.shape[1])
        m.upload(v)
        return m

    def close(self) -> None:
        if self._h:
            self._lib.storm_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipMatrix:
    """storm_hip_matrix_t: dense uint64 bitmap rows resident in HBM (padded layout)."""

    def __init__(self, ctx: HipContext, n_rows: int, n_words: int):
        self.ctx = ctx
        self._lib = ctx._lib
        h = C.c_void_p()
        check(self._lib.storm_hip_matrix_create(ctx._h, n_rows, n_words, C.byref(h)),
              "storm_hip_matrix_create")
        self._h = h
        self.n_rows, self.n_words = n_rows, n_words

    # -- data movement
    def upload(self, vals: np.ndarray, row0: int = 0) -> None:
        v = np.ascontiguousarray(vals, dtype=np.uint64)
        check(self._lib.storm_hip_matrix_upload(self.ctx._h, self._h, row0, v.shape[0], _ptr(v),
                                                v.shape[1]), "storm_hip_matrix_upload")

    def import_device(self, data_ptr: int, n_rows: int, stride_words: int, row0: int = 0) -> None:
        check(self._lib.storm_hip_matrix_import(self.ctx._h, self._h, row0, n_rows,
                                                C.c_void_p(data_ptr), stride_words),
              "storm_hip_matrix_import")

    def download(self, row0: int = 0, n_rows: Optional[int] = None) -> np.ndarray:
        n = self.n_rows - row0 if n_rows is None else n_rows
        out = np.zeros((n, self.n_words), dtype=np.uint64)
        check(self._lib.storm_hip_matrix_download(self.ctx._h, self._h, row0, n, _ptr(out),
                                                  self.n_words), "storm_hip_matrix_download")
        return out

    def set_rows_from_positions(self, rows: Sequence[Iterable[int]], row0: int = 0) -> None:
        arrs = [_u32(r) for r in rows]
        offs = np.zeros(len(arrs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([a.size for a in arrs], dtype=np.uint64)
        pos = np.concatenate(arrs) if arrs and offs[-1] else np.zeros(1, dtype=np.uint32)
        check(self._lib.storm_hip_matrix_set_rows_from_positions(
            self.ctx._h, self._h, row0, len(arrs), _ptr(offs), _ptr(pos)),
            "storm_hip_matrix_set_rows_from_positions")

    def fill_synthetic(self, n_bits: int, draws: int, seed: int = 42) -> None:
        check(self._lib.storm_hip_matrix_fill_synthetic(self.ctx._h, self._h, n_bits, draws,
                                                        seed), "storm_hip_matrix_fill_synthetic")

    def clear(self) -> None:
        check(self._lib.storm_hip_matrix_clear(self.ctx._h, self._h), "storm_hip_matrix_clear")

    @property
    def device_ptr(self) -> int:
        return int(self._lib.storm_hip_matrix_device_ptr(self._h) or 0)

    @property
    def stride_words(self) -> int:
        return int(self._lib.storm_hip_matrix_stride_words(self._h))

    # -- the hot path
    def pairw(self, shard_rank: int = 0, shard_count: int = 1) -> int:
        out = C.c_uint64()
        check(self._lib.storm_hip_pairw_dense(self.ctx._h, self._h, shard_rank, shard_count,
                                              C.byref(out)), "storm_hip_pairw_dense")
        return int(out.value)

    def pairw_launch(self, d_total_ptr: int, shard_rank: int = 0, shard_count: int = 1) -> None:
        """Asynchronous on the context's stream; d_total_ptr = device pointer to one uint64."""
        check(self._lib.storm_hip_pairw_dense_launch(self.ctx._h, self._h, shard_rank,
                                                     shard_count, C.c_void_p(d_total_ptr)),
              "storm_hip_pairw_dense_launch")

    def square(self, other: "HipMatrix") -> int:
        out = C.c_uint64()
        check(self._lib.storm_hip_square_dense(self.ctx._h, self._h, other._h, C.byref(out)),
              "storm_hip_square_dense")
        return int(out.value)

    def tile_counts(self, i0: int, i1: int, j0: int, j1: int) -> np.ndarray:
        out = np.zeros((i1 - i0, j1 - j0), dtype=np.uint32)
        check(self._lib.storm_hip_tile_counts(self.ctx._h, self._h, i0, i1, j0, j1, _ptr(out)),
              "storm_hip_tile_counts")
        return out

    OPS = {"and": 0, "or": 1, "xor": 2}

    def pairw_matrix(self, op: str = "and") -> np.ndarray:
        """[n_rows, n_rows] uint32, entry (i, j) = popcount(row_i OP row_j) for i < j, else 0."""
        out = np.zeros((self.n_rows, self.n_rows), dtype=np.uint32)
        check(self._lib.storm_hip_pairw_matrix(self.ctx._h, self._h, self.OPS[op], _ptr(out)),
              "storm_hip_pairw_matrix")
        return out

    def pairw_matrix_device(self, d_out: int, ld: int, op: str = "and") -> None:
        """Same, into a device buffer (address `d_out`, n_rows x ld uint32); synchronous."""
        check(self._lib.storm_hip_pairw_matrix_device(self.ctx._h, self._h, self.OPS[op],
                                                      C.c_void_p(d_out), ld),
              "storm_hip_pairw_matrix_device")

    def square_matrix(self, other: "HipMatrix", op: str = "and") -> np.ndarray:
        """[self.n_rows, other.n_rows] uint32: popcount(row_i(self) OP row_j(other)) for all i, j."""
        out = np.zeros((self.n_rows, other.n_rows), dtype=np.uint32)
        check(self._lib.storm_hip_square_matrix(self.ctx._h, self._h, other._h, self.OPS[op], _ptr(out)),
              "storm_hip_square_matrix")
        return out

    def pairw_matrix_band_device(self, d_out: int, ld: int, row0: int, n_band_rows: int,
                                 op: str = "and") -> None:
        """Rows [row0, row0 + n_band_rows) of the triangle into a device buffer (n_band_rows x ld)."""
        check(self._lib.storm_hip_pairw_matrix_band_device(self.ctx._h, self._h, self.OPS[op], row0,
                                                           n_band_rows, C.c_void_p(d_out), ld),
              "storm_hip_pairw_matrix_band_device")

    def row_counts(self) -> np.ndarray:
        out = np.zeros(self.n_rows, dtype=np.uint32)
        check(self._lib.storm_hip_row_counts(self.ctx._h, self._h, _ptr(out)),
              "storm_hip_row_counts")
        return out

    def pairw_op(self, op: str) -> int:
        """sum_{i<j} popcount(row_i OP row_j), OP in and / or / xor."""
        total = C.c_uint64(0)
        check(self._lib.storm_hip_pairw_dense_op(self.ctx._h, self._h, self.OPS[op],
                                                 C.byref(total)), "storm_hip_pairw_dense_op")
        return int(total.value)

    def column_identity(self) -> int:
        out = C.c_uint64()
        check(self._lib.storm_hip_column_identity(self.ctx._h, self._h, C.byref(out)),
              "storm_hip_column_identity")
        return int(out.value)

    def close(self) -> None:
        if self._h:
            self._lib.storm_hip_matrix_destroy(self.ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
