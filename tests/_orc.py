"""ctypes binding of the CPU oracle (oracle/liborc.so + oracle/libmtgen.so).

TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. Builds the two libraries with `make -C oracle` when they are missing.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
u64, u32, vp = C.c_uint64, C.c_uint32, C.c_void_p


def build():
    need = [os.path.join(ORACLE_DIR, n) for n in ("liborc.so", "libmtgen.so")]
    src = [os.path.join(ORACLE_DIR, n) for n in ("storm_oracle.c", "storm_oracle.h",
                                                 "mt19937_inputs.cpp")]
    stale = any(not os.path.exists(p) for p in need) or \
        min(os.path.getmtime(p) for p in need) < max(os.path.getmtime(p) for p in src)
    if stale:
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


def _p(a):
    return None if a is None else a.ctypes.data_as(vp)


class Oracle:
    def __init__(self):
        build()
        self.lib = lib = C.CDLL(os.path.join(ORACLE_DIR, "liborc.so"))
        self.mt = C.CDLL(os.path.join(ORACLE_DIR, "libmtgen.so"))
        self.mt.mtgen_positions.restype = u64
        self.mt.mtgen_positions.argtypes = [u32, u32, u32, u32, vp, vp, u64]
        sig = {
            "orc_contig_new": (vp, [C.c_size_t]), "orc_contig_free": (None, [vp]),
            "orc_contig_add": (C.c_int, [vp, vp, u32]), "orc_contig_clear": (C.c_int, [vp]),
            "orc_contig_n_rows": (u64, [vp]), "orc_contig_n_words": (u32, [vp]),
            "orc_contig_scalar_cutoff": (u32, [vp]), "orc_contig_data": (vp, [vp]),
            "orc_contig_set_leaf": (None, [vp, vp]),
            "orc_contig_pairw_intersect_cardinality": (u64, [vp]),
            "orc_contig_pairw_intersect_cardinality_blocked": (u64, [vp, u32]),
            "orc_contig_pairw_intersect_cardinality_list": (u64, [vp]),
            "orc_contig_pairw_intersect_cardinality_blocked_list": (u64, [vp, u32]),
            "orc_storm_new": (vp, []), "orc_storm_free": (None, [vp]),
            "orc_storm_add": (C.c_int, [vp, vp, u32]), "orc_storm_clear": (C.c_int, [vp]),
            "orc_storm_n_rows": (u64, [vp]), "orc_storm_serialized_size": (u64, [vp]),
            "orc_storm_pairw_intersect_cardinality": (u64, [vp]),
            "orc_storm_pairw_intersect_cardinality_blocked": (u64, [vp, u32]),
            "orc_storm_block_census": (None, [vp, vp]),
            "orc_truth_naive_dense": (u64, [vp, u64, u64]),
            "orc_truth_column_count": (u64, [vp, u64, u64]),
            "orc_tile_counts": (None, [vp, u64, u64, u64, u64, u64, vp]),
            "orc_tile_counts_op": (None, [vp, u64, u64, u64, u64, u64, C.c_int, vp]),
            "orc_truth_naive_dense_op": (u64, [vp, u64, u64, C.c_int]),
            "orc_storm_pair_counts": (C.c_int, [vp, u64, u64, vp, u64]),
            "orc_wrapper_diag": (u64, [u32, vp, u32, vp]),
            "orc_wrapper_diag_blocked": (u64, [u32, vp, u32, vp, u32]),
            "orc_wrapper_square": (u64, [u32, vp, u32, vp, u32, vp]),
            "orc_wrapper_diag_list": (u64, [u32, vp, u32, vp, vp, vp, vp, vp, u32]),
            "orc_wrapper_diag_list_blocked": (u64, [u32, vp, u32, vp, vp, vp, vp, vp, u32, u32]),
            "orc_get_intersect_count_func_kind": (vp, [C.c_int]),
            "orc_get_intersect_count_func": (vp, [C.c_size_t]),
            "orc_best_leaf_kind": (C.c_int, []), "orc_leaf_name": (C.c_char_p, [C.c_int]),
            "orc_intersect_count_scalar": (u64, [vp, vp, C.c_size_t]),
            "orc_intersect_count_avx2": (u64, [vp, vp, C.c_size_t]),
            "orc_intersect_count_avx512": (u64, [vp, vp, C.c_size_t]),
            "orc_intersect_vector16_cardinality": (u64, [vp, vp, u32, u32]),
            "orc_intersect_vector32_unsafe": (u64, [vp, vp, u32, u32, vp]),
            "orc_intersect_bitmaps_scalar_list": (u64, [vp, vp, vp, vp, u32, u32]),
            "orc_time_blocked": (C.c_double, [vp, u32, u32, C.c_int, u32, C.POINTER(u64)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args

    # ---- leaves
    def leaf(self, kind=-1):
        return self.lib.orc_get_intersect_count_func_kind(kind)

    def leaf_name(self, kind=-1):
        return self.lib.orc_leaf_name(kind).decode()

    # ---- truths on dense matrices
    def truth_naive(self, mat: np.ndarray) -> int:
        m = np.ascontiguousarray(mat, dtype=np.uint64)
        return int(self.lib.orc_truth_naive_dense(_p(m), m.shape[0], m.shape[1]))

    def truth_columns(self, mat: np.ndarray) -> int:
        m = np.ascontiguousarray(mat, dtype=np.uint64)
        return int(self.lib.orc_truth_column_count(_p(m), m.shape[0], m.shape[1]))

    def tile_counts(self, mat, i0, i1, j0, j1) -> np.ndarray:
        m = np.ascontiguousarray(mat, dtype=np.uint64)
        out = np.zeros((i1 - i0, j1 - j0), dtype=np.uint32)
        self.lib.orc_tile_counts(_p(m), m.shape[1], i0, i1, j0, j1, _p(out))
        return out

    def tile_counts_op(self, mat, i0, i1, j0, j1, op) -> np.ndarray:
        m = np.ascontiguousarray(mat, dtype=np.uint64)
        out = np.zeros((i1 - i0, j1 - j0), dtype=np.uint32)
        self.lib.orc_tile_counts_op(_p(m), m.shape[1], i0, i1, j0, j1, op, _p(out))
        return out

    def truth_naive_op(self, mat: np.ndarray, op: int) -> int:
        m = np.ascontiguousarray(mat, dtype=np.uint64)
        return int(self.lib.orc_truth_naive_dense_op(_p(m), m.shape[0], m.shape[1], op))

    # ---- raw wrappers
    def wrapper_diag(self, mat, kind=-1) -> int:
        m = np.ascontiguousarray(mat, dtype=np.uint64)
        return int(self.lib.orc_wrapper_diag(m.shape[0], _p(m), m.shape[1], self.leaf(kind)))

    def wrapper_diag_blocked(self, mat, bsize, kind=-1) -> int:
        m = np.ascontiguousarray(mat, dtype=np.uint64)
        return int(self.lib.orc_wrapper_diag_blocked(m.shape[0], _p(m), m.shape[1],
                                                     self.leaf(kind), bsize))

    def wrapper_square(self, a, b, kind=-1) -> int:
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        return int(self.lib.orc_wrapper_square(a.shape[0], _p(a), b.shape[0], _p(b), a.shape[1],
                                               self.leaf(kind)))

    def time_blocked(self, mat, kind, bsize):
        m = np.ascontiguousarray(mat, dtype=np.uint64)
        tot = u64()
        secs = self.lib.orc_time_blocked(_p(m), m.shape[0], m.shape[1], kind, bsize,
                                         C.byref(tot))
        return float(secs), int(tot.value)

    # ---- containers
    def contig(self, n_bits, rows):
        return OrcContig(self, n_bits, rows)

    def storm(self, rows):
        return OrcStorm(self, rows)

    # ---- the survey's mt19937 inputs
    def mt_positions(self, M, N, draws, seed=42):
        offs = np.zeros(N + 1, dtype=np.uint64)
        pos = np.zeros(max(1, N * min(draws, M)), dtype=np.uint32)
        used = self.mt.mtgen_positions(M, N, draws, seed, _p(offs), _p(pos), pos.size)
        assert used != (1 << 64) - 1
        return [pos[int(offs[j]):int(offs[j + 1])].copy() for j in range(N)]


class OrcContig:
    def __init__(self, orc, n_bits, rows=()):
        self.o, self.lib = orc, orc.lib
        self.h = self.lib.orc_contig_new(n_bits)
        for r in rows:
            self.add(r)

    def add(self, values):
        v = np.ascontiguousarray(values, dtype=np.uint32)
        return self.lib.orc_contig_add(self.h, _p(v if v.size else np.zeros(1, np.uint32)), v.size)

    def dense(self) -> np.ndarray:
        n, w = self.lib.orc_contig_n_rows(self.h), self.lib.orc_contig_n_words(self.h)
        if n == 0:
            return np.zeros((0, w), dtype=np.uint64)
        buf = (C.c_uint64 * (n * w)).from_address(self.lib.orc_contig_data(self.h))
        return np.frombuffer(buf, dtype=np.uint64).reshape(n, w).copy()

    def pairw(self): return int(self.lib.orc_contig_pairw_intersect_cardinality(self.h))
    def pairw_blocked(self, b): return int(self.lib.orc_contig_pairw_intersect_cardinality_blocked(self.h, b))
    def pairw_list(self): return int(self.lib.orc_contig_pairw_intersect_cardinality_list(self.h))
    def pairw_blocked_list(self, b): return int(self.lib.orc_contig_pairw_intersect_cardinality_blocked_list(self.h, b))
    def cutoff(self): return int(self.lib.orc_contig_scalar_cutoff(self.h))

    def __del__(self):
        try:
            self.lib.orc_contig_free(self.h)
        except Exception:
            pass


class OrcStorm:
    def __init__(self, orc, rows=()):
        self.o, self.lib = orc, orc.lib
        self.h = self.lib.orc_storm_new()
        for r in rows:
            self.add(r)

    def add(self, values):
        v = np.ascontiguousarray(values, dtype=np.uint32)
        return self.lib.orc_storm_add(self.h, _p(v if v.size else np.zeros(1, np.uint32)), v.size)

    def pairw(self): return int(self.lib.orc_storm_pairw_intersect_cardinality(self.h))
    def pairw_blocked(self, b=0): return int(self.lib.orc_storm_pairw_intersect_cardinality_blocked(self.h, b))
    def serialized_size(self): return int(self.lib.orc_storm_serialized_size(self.h))

    def pair_counts(self, i0=0, i1=None) -> np.ndarray:
        """[i1 - i0, n_rows] uint32: the row-pair function of storm.c:790-814 for rows i0 <= i < i1, i < j."""
        n = int(self.lib.orc_storm_n_rows(self.h))
        i1 = n if i1 is None else i1
        out = np.zeros((i1 - i0, n), dtype=np.uint32)
        assert self.lib.orc_storm_pair_counts(self.h, i0, i1, _p(out), n) == 0
        return out

    def census(self):
        out = np.zeros(2, dtype=np.uint64)
        self.lib.orc_storm_block_census(self.h, _p(out))
        return int(out[0]), int(out[1])

    def __del__(self):
        try:
            self.lib.orc_storm_free(self.h)
        except Exception:
            pass
