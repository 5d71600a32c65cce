"""The drop-in boundary without a GPU: the library loads, exports every symbol the headers in
include/ declare, keeps the reference's host-side conventions, and refuses to compute without
a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from stormbitmaps_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def _declared_functions():
    names = set()
    for rel in ("storm.h", "storm_hip.h", "storm_synth.h", os.path.join("libalgebra", "libalgebra.h")):
        src = open(os.path.join(INC, rel)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
        # drop static inline definitions (header-only helpers) and typedef'd function pointers
        src = re.sub(r"static\s+inline[^{;]*\{.*?\n\}", "", src, flags=re.S)
        src = re.sub(r"typedef[^;]*;", "", src, flags=re.S)
        for m in re.finditer(r"\b((?:STORM|storm_hip|storm_synth)_\w+)\s*\(", src):
            names.add(m.group(1))
    names -= {"STORM_ALIGN"}
    return sorted(names)


def test_library_exports_every_declared_symbol(lib):
    declared = _declared_functions()
    assert len(declared) > 80
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding table covers the device C-ABI completely
    unbound = [n for n in declared if n.startswith("storm_hip_") and n not in _lib.SIGNATURES]
    assert not unbound, unbound


def test_library_exports_nothing_but_the_c_abi():
    """a drop-in library exports STORM_* / storm_hip_* / storm_synth_* and nothing else — no C++
    internals, no kernel handles (-fvisibility=hidden + csrc/exports.map) — and everything it exports is declared in a
    header of include/ (or is a variable a header declares)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    names = [line.split()[-1] for line in out.splitlines() if line.strip()]
    assert len(names) > 80
    foreign = [n for n in names if not re.match(r"(STORM|storm_hip|storm_synth)_\w+$", n)]
    assert not foreign, foreign[:10]
    headers = "".join(open(os.path.join(INC, rel)).read()
                      for rel in ("storm.h", "storm_hip.h", "storm_synth.h", os.path.join("libalgebra", "libalgebra.h")))
    undeclared = [n for n in names if not re.search(r"\b%s\b" % re.escape(n), headers)]
    assert not undeclared, undeclared


def test_reference_symbols_of_survey_8b_are_present(lib):
    # the 41 functions storm.c defines, by family
    fams = {
        "STORM_contig_": ["new", "free", "add", "clear", "pairw_intersect_cardinality",
                          "pairw_intersect_cardinality_blocked", "pairw_intersect_cardinality_list",
                          "pairw_intersect_cardinality_blocked_list"],
        "STORM_": ["new", "free", "add", "clear", "pairw_intersect_cardinality",
                   "pairw_intersect_cardinality_blocked", "serialized_size"],
        "STORM_wrapper_": ["diag", "diag_blocked", "square", "diag_list", "diag_list_blocked"],
        "STORM_intersect_": ["vector16_cardinality", "vector32_unsafe", "bitmaps_scalar_list"],
        "STORM_bitmap_": ["new", "init", "free", "add", "add_with_scalar", "add_scalar_only",
                          "intersect_cardinality", "intersect_cardinality_func", "clear",
                          "serialized_size"],
        "STORM_bitmap_cont_": ["new", "init", "free", "add", "clear", "intersect_cardinality",
                               "intersect_cardinality_premade", "serialized_size"],
    }
    names = [p + s for p, ss in fams.items() for s in ss]
    assert len(names) == 41
    assert all(hasattr(lib, n) for n in names)
    # libalgebra surface the callers use
    for n in ("STORM_get_intersect_count_func", "STORM_get_alignment", "STORM_aligned_malloc",
              "STORM_aligned_free", "STORM_get_cpuid", "STORM_intersect_count_scalar",
              "STORM_intersect_count_scalar_list"):
        assert hasattr(lib, n)


def _no_gpu(lib):
    return lib.storm_hip_device_count() == 0


def test_no_cpu_fallback_without_device(lib):
    if not _no_gpu(lib):
        pytest.skip("a GPU is visible; the loud-failure path is exercised on the CPU container")
    h = C.c_void_p()
    assert lib.storm_hip_ctx_create(0, None, C.byref(h)) == -2  # STORM_HIP_ENODEV
    assert b"no CPU fallback" in lib.storm_hip_last_error()
    c = lib.STORM_contig_new(4096)
    for i in range(3):
        v = np.array([1 + i, 7, 100], dtype=np.uint32)
        assert lib.STORM_contig_add(c, v.ctypes.data, 3) == 3
    assert lib.STORM_contig_pairw_intersect_cardinality(c) == 2**64 - 1
    assert lib.STORM_contig_pairw_intersect_cardinality_blocked(c, 31) == 2**64 - 1
    mat = synth.dense_matrix(256, 4, 30)
    assert lib.STORM_wrapper_diag(4, mat.ctypes.data, mat.shape[1], None) == 2**64 - 1
    s = lib.STORM_new()
    for i in range(3):
        v = np.array([1 + i, 7, 100], dtype=np.uint32)
        assert lib.STORM_add(s, v.ctypes.data, 3) == 1
    assert lib.STORM_pairw_intersect_cardinality(s) == 2**64 - 1
    lib.STORM_contig_free(c)
    lib.STORM_free(s)


def test_host_conventions_match_reference(lib):
    assert lib.STORM_contig_pairw_intersect_cardinality(None) == 2**64 - 1   # storm.c:1150
    assert lib.STORM_pairw_intersect_cardinality_blocked(None, 0) == 2**64 - 1  # storm.c:898
    c = lib.STORM_contig_new(4096)
    assert lib.STORM_contig_pairw_intersect_cardinality_list(c) == 2**64 - 2  # storm.c:1245
    v = np.array([3, 9, 9, 4000], dtype=np.uint32)
    assert lib.STORM_contig_add(None, v.ctypes.data, 4) == -1
    assert lib.STORM_contig_add(c, None, 4) == -2
    assert lib.STORM_contig_add(c, v.ctypes.data, 0) == 0
    assert lib.STORM_contig_add(c, v.ctypes.data, 4) == 4
    assert lib.STORM_contig_clear(c) == 1 and lib.STORM_contig_clear(None) == -1
    # fewer than two rows: nothing to pair, no device needed
    assert lib.STORM_contig_pairw_intersect_cardinality(c) == 0
    lib.STORM_contig_free(c)
    assert lib.STORM_add(None, v.ctypes.data, 4) == -1
    assert lib.STORM_clear(None) == -1 and lib.STORM_serialized_size(None) == 0
    assert lib.STORM_get_alignment() == 64


def test_struct_layout_keeps_reference_members_first():
    # compile a tiny C program against include/storm.h and check the public members' offsets
    import subprocess
    import tempfile
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "storm.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n",
    offsetof(STORM_contiguous_t, data), offsetof(STORM_contiguous_t, scalar),
    offsetof(STORM_contiguous_t, n_scalar), offsetof(STORM_contiguous_t, bitmaps),
    offsetof(STORM_contiguous_t, n_data), offsetof(STORM_contiguous_t, vector_length),
    offsetof(STORM_contiguous_t, intsec_func), offsetof(STORM_contiguous_t, scalar_cutoff));
  printf("%zu %zu %zu\n", offsetof(STORM_t, conts), offsetof(STORM_t, n_conts), offsetof(STORM_t, m_conts));
  printf("%zu %zu\n", sizeof(STORM_bitmap_t), offsetof(STORM_bitmap_cont_t, block_ids));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", INC, src, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    assert [int(x) for x in out[:8]] == [0, 8, 16, 24, 32, 64, 80, 92]
    assert [int(x) for x in out[8:11]] == [0, 8, 12]
    assert int(out[11]) % 64 == 0 and int(out[12]) == 8


def test_one_pair_host_helpers_equal_oracle(lib, orc):
    rng = np.random.default_rng(5)
    a = np.unique(rng.integers(0, 65536, size=900, dtype=np.uint16)).astype(np.uint16)
    b = np.unique(rng.integers(0, 65536, size=4000, dtype=np.uint16)).astype(np.uint16)
    assert lib.STORM_intersect_vector16_cardinality(a.ctypes.data, b.ctypes.data, a.size, b.size) == \
        orc.lib.orc_intersect_vector16_cardinality(a.ctypes.data, b.ctypes.data, a.size, b.size)
    x = rng.integers(0, 2**64, size=1024, dtype=np.uint64)
    y = rng.integers(0, 2**64, size=1024, dtype=np.uint64)
    assert lib.STORM_intersect_count_scalar(x.ctypes.data, y.ctypes.data, 1024) == \
        orc.lib.orc_intersect_count_scalar(x.ctypes.data, y.ctypes.data, 1024)
    l1 = np.unique(rng.integers(0, 65536, size=50)).astype(np.uint32)
    l2 = np.unique(rng.integers(0, 65536, size=70)).astype(np.uint32)
    assert lib.STORM_intersect_bitmaps_scalar_list(x.ctypes.data, y.ctypes.data, l1.ctypes.data, l2.ctypes.data, l1.size, l2.size) == \
        orc.lib.orc_intersect_bitmaps_scalar_list(x.ctypes.data, y.ctypes.data, l1.ctypes.data, l2.ctypes.data, l1.size, l2.size)


def test_storm_t_host_build_matches_oracle_sizes(lib, orc):
    # construction is host work: block kinds and byte counts must match the restated reference
    for M, N, d in ((524288, 40, 524), (65536, 60, 4230), (524288, 12, 131072), (524288, 30, 1)):
        rows = synth.positions(M, N, d, seed=42)
        s = lib.STORM_new()
        for r in rows:
            assert lib.STORM_add(s, r.ctypes.data, r.size) == 1
        assert lib.STORM_serialized_size(s) == orc.storm(rows).serialized_size()
        assert lib.STORM_clear(s) == 1
        assert lib.STORM_serialized_size(s) == 8
        lib.STORM_free(s)


def test_benchmark_cli_is_built_and_refuses_bad_arguments():
    import subprocess
    exe = os.path.join(ROOT, "stormbitmaps_amd", "storm_benchmark")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode != 0 and "Usage" in res.stderr
    res = subprocess.run([exe, "0", "10"], capture_output=True, text=True)
    assert res.returncode != 0 and "non-positive" in res.stderr


def test_simd_leaves_of_the_libalgebra_surface_match_the_scalar_leaf(lib, orc):
    """STORM_intersect_count_sse4 / _avx2 / _avx512 (benchmark.cpp:961,1013,1031; host functions of the product,
    stormbitmaps_amd/csrc/storm_leaves.c): every one the host CPU supports (STORM_get_cpuid, as the reference's
    harness gates them, benchmark.cpp:949-1022) equals the scalar leaf and the oracle's leaf on lengths around
    every vector width and round boundary, aligned and unaligned."""
    import ctypes as C
    rng = np.random.default_rng(7)
    a = rng.integers(0, 1 << 63, size=2100, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=2100, dtype=np.uint64)
    b = rng.integers(0, 1 << 63, size=2100, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=2100, dtype=np.uint64)
    cpuid = lib.STORM_get_cpuid()
    leaves = [("STORM_intersect_count_scalar", 0), ("STORM_intersect_count_sse4", 1),
              ("STORM_intersect_count_avx2", 2), ("STORM_intersect_count_avx512", 4)]
    ran = 0
    for name, bit in leaves:
        assert hasattr(lib, name), name
        if bit and not (cpuid & bit):
            continue
        f = getattr(lib, name)
        f.restype = C.c_uint64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        for off in (0, 1, 3):
            for n in list(range(0, 70)) + [127, 128, 129, 255, 256, 257, 511, 512, 513, 1024, 2047]:
                want = int(sum(bin(int(x) & int(y)).count("1") for x, y in zip(a[off:off + n], b[off:off + n]))) if n < 70 \
                    else int(np.unpackbits((a[off:off + n] & b[off:off + n]).view(np.uint8)).sum())
                got = int(f(a[off:].ctypes.data, b[off:].ctypes.data, n))
                assert got == want, (name, off, n, got, want)
        ran += 1
    assert ran >= 1


def test_options_are_validated_before_they_are_remembered(lib):
    """STORM_hip_set_option used to remember an unknown key or a value out of range when no context existed yet
    (and 0 was returned); the contexts made later dropped it silently. storm_hip_option_check validates without a context."""
    lib.storm_hip_option_check.restype = C.c_int
    lib.storm_hip_option_check.argtypes = [C.c_char_p, C.c_int64]
    assert lib.storm_hip_option_check(b"k2_tile_shape", 6) == 0
    assert lib.storm_hip_option_check(b"k2_tile_shape", 99) != 0
    assert lib.storm_hip_option_check(b"k2_tile_shap", 2) != 0
    assert b"unknown" in lib.storm_hip_last_error().lower() or lib.storm_hip_last_error()
    lib.STORM_hip_set_option.restype = C.c_int
    lib.STORM_hip_set_option.argtypes = [C.c_char_p, C.c_int64]
    assert lib.STORM_hip_set_option(b"k2_tile_shap", 2) == -1
    assert lib.STORM_hip_set_option(b"k2_part_slots", 7) == -1
    assert lib.STORM_hip_set_option(b"k2_part_slots", 0) == 0
