"""ctypes loader for libstorm_hip.so (the C-ABI of include/storm_hip.h and include/storm.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C stormbitmaps_amd/csrc``.
There is no Python or CPU fallback: if the shared object is missing, or no gfx950 device is
usable at call time, the failure is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# STORM_HIP_LIB: another build of the same library (the tools' probes build, `make probes`)
LIB_PATH = os.environ.get("STORM_HIP_LIB") or os.path.join(_HERE, "libstorm_hip.so")

u64, u32, u16, u8, i64 = C.c_uint64, C.c_uint32, C.c_uint16, C.c_uint8, C.c_int64
vp, cp, sz = C.c_void_p, C.c_char_p, C.c_size_t
P = C.POINTER

# name -> (restype, argtypes): every symbol include/storm_hip.h, include/storm_synth.h and the
# entry points of include/storm.h that the Python mirror uses.
SIGNATURES = {
    # storm_hip.h
    "storm_hip_last_error": (cp, []),
    "storm_hip_device_count": (C.c_int, []),
    "storm_hip_device_arch": (C.c_int, [C.c_int, cp, sz]),
    "storm_hip_device_pci_bus_id": (C.c_int, [C.c_int, cp, sz]),
    "storm_hip_ctx_create": (C.c_int, [C.c_int, vp, P(vp)]),
    "storm_hip_ctx_set_stream": (C.c_int, [vp, vp]),
    "storm_hip_ctx_synchronize": (C.c_int, [vp]),
    "storm_hip_ctx_destroy": (None, [vp]),
    "storm_hip_matrix_create": (C.c_int, [vp, u64, u32, P(vp)]),
    "storm_hip_matrix_upload": (C.c_int, [vp, vp, u64, u64, vp, u64]),
    "storm_hip_matrix_resize": (C.c_int, [vp, vp, u64]),
    "storm_hip_matrix_import": (C.c_int, [vp, vp, u64, u64, vp, u64]),
    "storm_hip_matrix_download": (C.c_int, [vp, vp, u64, u64, vp, u64]),
    "storm_hip_matrix_set_rows_from_positions": (C.c_int, [vp, vp, u64, u64, vp, vp]),
    "storm_hip_matrix_fill_synthetic": (C.c_int, [vp, vp, u64, u32, u64]),
    "storm_hip_matrix_clear": (C.c_int, [vp, vp]),
    "storm_hip_matrix_destroy": (None, [vp, vp]),
    "storm_hip_matrix_rows": (u64, [vp]),
    "storm_hip_matrix_words": (u32, [vp]),
    "storm_hip_matrix_stride_words": (u64, [vp]),
    "storm_hip_matrix_device_ptr": (vp, [vp]),
    "storm_hip_pairw_dense_launch": (C.c_int, [vp, vp, u32, u32, vp]),
    "storm_hip_pairw_dense": (C.c_int, [vp, vp, u32, u32, P(u64)]),
    "storm_hip_pairw_dense_upload": (C.c_int, [vp, vp, vp, u64, P(u64)]),
    "storm_hip_pairw_dense_begin": (C.c_int, [vp, vp, u32, u32]),
    "storm_hip_pairw_dense_end": (C.c_int, [vp, P(u64)]),
    "storm_hip_square_dense": (C.c_int, [vp, vp, vp, P(u64)]),
    "storm_hip_tile_counts": (C.c_int, [vp, vp, u64, u64, u64, u64, vp]),
    "storm_hip_column_identity": (C.c_int, [vp, vp, P(u64)]),
    "storm_hip_pairw_matrix_device": (C.c_int, [vp, vp, C.c_int, vp, u64]),
    "storm_hip_pairw_matrix": (C.c_int, [vp, vp, C.c_int, vp]),
    "storm_hip_row_counts": (C.c_int, [vp, vp, vp]),
    "storm_hip_pairw_matrix_band_device": (C.c_int, [vp, vp, C.c_int, u64, u64, vp, u64]),
    "storm_hip_pairw_matrix_band": (C.c_int, [vp, vp, C.c_int, u64, u64, vp, u64]),
    "storm_hip_pairw_matrix_band_begin": (C.c_int, [vp, vp, C.c_int, u64, u64, vp, u64]),
    "storm_hip_pairw_matrix_band_end": (C.c_int, [vp]),
    "storm_hip_strip_plan": (C.c_int, [u64, u32, u32, u32, vp, u64, vp]),
    "storm_hip_strip_plan2": (C.c_int, [u64, u32, u32, u32, C.c_int, C.c_int, vp, u64, vp]),
    "storm_hip_strip_plan3": (C.c_int, [u64, u32, u32, u32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u32,
                                        vp, u64, vp, vp]),
    "storm_hip_stream_plan": (C.c_int, [u64, u32, u32, u32, u32, vp, u64, vp, vp]),
    "storm_hip_matrix_plan": (C.c_int, [u64, u64, u32, u64, u64, u32, C.c_int, C.c_int, C.c_int, vp, u64, vp]),
    "storm_hip_square_matrix_device": (C.c_int, [vp, vp, vp, C.c_int, vp, u64]),
    "storm_hip_square_matrix": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "storm_hip_kernel_time": (C.c_int, [vp, P(C.c_double), P(u64)]),
    "storm_hip_debug_strip_trace": (C.c_int, [vp, vp, u64, P(u64)]),
    "storm_hip_pairw_dense_op": (C.c_int, [vp, vp, C.c_int, P(u64)]),
    "storm_hip_ctx_set_option": (C.c_int, [vp, cp, i64]),
    "storm_hip_option_check": (C.c_int, [cp, i64]),
    "storm_hip_ctx_reserve_staging": (C.c_int, [vp]),
    "storm_hip_stage_create": (C.c_int, [vp, vp]),
    "storm_hip_stage_add": (C.c_int, [vp, vp, vp, vp]),
    "storm_hip_stage_add_list": (C.c_int, [vp, vp, vp, u32, vp]),
    "storm_hip_stage_count": (u64, [vp]),
    "storm_hip_stage_destroy": (None, [vp, vp]),
    "storm_hip_sparse_create_blocks_staged": (C.c_int, [vp, u64, u64, vp, vp, vp, vp, vp, vp, vp, vp]),
    "storm_hip_ctx_get_option": (i64, [vp, cp]),
    "storm_hip_last_launch_info": (C.c_int, [vp, P(u64 * 4)]),
    "storm_hip_comm_unique_id": (C.c_int, [vp]),
    "storm_hip_comm_init_rank": (C.c_int, [vp, vp, u32, u32, P(vp)]),
    "storm_hip_comm_allreduce_u64": (C.c_int, [vp, vp, P(u64)]),
    "storm_hip_comm_allreduce_u64s": (C.c_int, [vp, vp, P(u64), C.c_uint32]),
    "storm_hip_last_pass_report": (C.c_int, [vp, P(u64)]),
    "storm_hip_comm_allreduce_result": (C.c_int, [vp, vp, P(u64)]),
    "storm_hip_comm_rank": (u32, [vp]),
    "storm_hip_comm_world": (u32, [vp]),
    "storm_hip_comm_destroy": (None, [vp]),
    "storm_hip_sparse_create": (C.c_int, [vp, u64, u64, vp, vp, vp, vp, vp, vp, u64, vp, u64,
                                          P(vp)]),
    "storm_hip_sparse_destroy": (None, [vp, vp]),
    "storm_hip_pairw_sparse": (C.c_int, [vp, vp, u32, u32, P(u64)]),
    "storm_hip_sparse_create_serialized": (C.c_int, [vp, vp, u64, P(vp)]),
    "storm_hip_sparse_create_blocks": (C.c_int, [vp, u64, u64, vp, vp, vp, vp, vp, P(vp)]),
    "storm_hip_matrix_create_from_blocks": (C.c_int, [vp, u64, u64, vp, vp, vp, vp, vp, P(vp)]),
    "storm_hip_rowlists_create_blocks": (C.c_int, [vp, u64, u64, vp, vp, vp, vp, vp, P(vp)]),
    "storm_hip_rowlists_create_blocks_staged": (C.c_int, [vp, u64, u64, vp, vp, vp, vp, vp, vp, vp, P(vp)]),
    "storm_hip_rowlists_destroy": (None, [vp, vp]),
    "storm_hip_rowlists_worthwhile": (C.c_int, [vp, vp]),
    "storm_hip_rowlists_worthwhile_counts": (C.c_int, [vp, u64, u64, u64]),
    "storm_hip_rowlists_pairw_matrix_device": (C.c_int, [vp, vp, C.c_int, vp, u64]),
    "storm_hip_rowlists_pairw_matrix": (C.c_int, [vp, vp, C.c_int, vp, u64]),
    "storm_hip_rowlists_n_elems": (u64, [vp]),
    "storm_hip_pairw_sparse_begin": (C.c_int, [vp, vp, u32, u32]),
    "storm_hip_pairw_sparse_end": (C.c_int, [vp, P(u64)]),
    "storm_hip_sparse_last_census": (C.c_int, [vp, P(u64 * 4)]),
    # storm_synth.h
    "storm_synth_fill_row": (None, [vp, u64, u64, u32, u64]),
    "storm_synth_fill_dense": (None, [vp, u64, u64, u64, u64, u32, u64]),
    "storm_synth_positions": (u32, [vp, vp, u64, u64, u32, u64]),
    "storm_synth_fill_storm": (i64, [vp, u64, u64, u64, u32, u64]),
    "storm_synth_fill_contig": (i64, [vp, u64, u64, u64, u32, u64]),
    # storm.h (containers + all-pairs entry points)
    "STORM_contig_new": (vp, [sz]),
    "STORM_contig_free": (None, [vp]),
    "STORM_contig_add": (C.c_int, [vp, vp, u32]),
    "STORM_contig_clear": (C.c_int, [vp]),
    "STORM_contig_pairw_intersect_cardinality": (u64, [vp]),
    "STORM_contig_pairw_intersect_cardinality_blocked": (u64, [vp, u32]),
    "STORM_contig_pairw_intersect_cardinality_list": (u64, [vp]),
    "STORM_contig_pairw_intersect_cardinality_blocked_list": (u64, [vp, u32]),
    "STORM_new": (vp, []),
    "STORM_free": (None, [vp]),
    "STORM_add": (C.c_int, [vp, vp, u32]),
    "STORM_clear": (C.c_int, [vp]),
    "STORM_pairw_intersect_cardinality": (u64, [vp]),
    "STORM_pairw_intersect_cardinality_blocked": (u64, [vp, u32]),
    "STORM_serialized_size": (u64, [vp]),
    "STORM_wrapper_diag": (u64, [u32, vp, u32, vp]),
    "STORM_wrapper_diag_blocked": (u64, [u32, vp, u32, vp, u32]),
    "STORM_wrapper_square": (u64, [u32, vp, u32, vp, u32, vp]),
    "STORM_wrapper_diag_list": (u64, [u32, vp, u32, vp, vp, vp, vp, vp, u32]),
    "STORM_wrapper_diag_list_blocked": (u64, [u32, vp, u32, vp, vp, vp, vp, vp, u32, u32]),
    "STORM_intersect_vector16_cardinality": (u64, [vp, vp, u32, u32]),
    "STORM_intersect_vector32_unsafe": (u64, [vp, vp, u32, u32, vp]),
    "STORM_intersect_bitmaps_scalar_list": (u64, [vp, vp, vp, vp, u32, u32]),
    "STORM_intersect_count_scalar": (u64, [vp, vp, sz]),
    "STORM_get_intersect_count_func": (vp, [sz]),
    "STORM_get_alignment": (u32, []),
    "STORM_get_cpuid": (C.c_int, []),
    "STORM_contig_pairw_matrix": (C.c_int, [vp, C.c_int, vp, u64, u64]),
    "STORM_contig_n_rows": (u64, [vp]),
    "STORM_n_rows": (u64, [vp]),
    "STORM_pairw_matrix": (C.c_int, [vp, C.c_int, vp, u64, u64]),
    "STORM_pairw_matrix_device": (C.c_int, [vp, C.c_int, vp, u64, u64]),
    "STORM_hip_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "STORM_contig_pairw_matrix_device": (C.c_int, [vp, C.c_int, vp, u64, u64]),
    "STORM_serialize": (u64, [vp, vp, u64]),
    "STORM_deserialize": (vp, [vp, u64]),
    "STORM_serialized_pairw_intersect_cardinality": (u64, [vp, u64]),
    "STORM_hip_invalidate": (C.c_int, [vp]),
    "STORM_contig_hip_invalidate": (C.c_int, [vp]),
    "STORM_hip_set_devices": (C.c_int, [C.c_int, vp]),
    "STORM_hip_set_shard": (C.c_int, [u32, u32]),
    "STORM_hip_set_thread_devices": (C.c_int, [C.c_int, C.c_int]),
    "STORM_hip_error": (cp, []),
    "STORM_hip_shutdown": (C.c_int, []),
    "STORM_hip_comm_unique_id": (C.c_int, [vp]),
    "STORM_hip_comm_init": (C.c_int, [vp]),
    "STORM_hip_comm_finalize": (C.c_int, []),
}

_lib = None


class StormHipError(RuntimeError):
    pass


def load():
    """Load libstorm_hip.so once. torch is imported first when present so that both share one
    HIP runtime (the wheel's libamdhip64.so.7 satisfies this library's NEEDED entry)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StormHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C stormbitmaps_amd/csrc` (there is no CPU fallback)")
    try:
        import torch  # noqa: F401  (side effect: loads the HIP runtime torch ships)
    except Exception:  # pragma: no cover - torch is optional for the pure C-ABI
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    lib = load()
    msg = lib.storm_hip_last_error()
    return msg.decode() if msg else ""


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise StormHipError(f"{what} failed (code {rc}): {last_error()}")


def kernel_source_hash() -> str:
    """Fingerprint of the device sources (csrc/*.hip, csrc/*.inc + the shared internal header): what a
    profile summary under profiles/ was measured on. bench.py only quotes measured HBM traffic
    whose fingerprint equals the current one."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".inc")) or name == "storm_hip_internal.h":
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]
