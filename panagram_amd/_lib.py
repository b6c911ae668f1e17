"""ctypes binding of libpanagram_hip.so (C-ABI: include/panagram_hip.h).

The library is the product; this module only loads it and declares prototypes.
There is deliberately no fallback: a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpanagram_hip.so")

PG_ANCHOR_COLSUMS = 1
PG_ANCHOR_ROWS_ONLY = 2
PG_ANCHOR_COLUMNS_ONLY = 4


class PanagramHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libpanagram_hip error {code}: {msg}")
        self.code = code


_lib = None

# name -> (restype, argtypes); every symbol declared in include/panagram_hip.h
_u8p, _u32p, _u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
_vp, _vpp = C.c_void_p, C.POINTER(C.c_void_p)
PROTOTYPES = {
    "pg_last_error": (C.c_char_p, []),
    "pg_version": (C.c_char_p, []),
    "pg_tile_positions": (C.c_uint32, []),
    "pg_ctx_create": (C.c_int, [C.c_int, _vpp]),
    "pg_ctx_destroy": (C.c_int, [_vp]),
    "pg_ctx_trim": (C.c_int, [_vp]),
    "pg_ctx_mem_info": (C.c_int, [_vp, _u64p, _u64p]),
    "pg_host_alloc": (C.c_int, [_vp, C.c_uint64, C.POINTER(C.c_void_p)]),
    "pg_host_free": (C.c_int, [_vp, _vp]),
    "pg_ctx_set_stream": (C.c_int, [_vp, _vp, C.c_int]),
    "pg_ctx_synchronize": (C.c_int, [_vp]),
    "pg_device_alloc": (C.c_int, [_vp, C.c_uint64, _vpp]),
    "pg_device_memset": (C.c_int, [_vp, _vp, C.c_int, C.c_uint64]),
    "pg_device_free": (C.c_int, [_vp, _vp]),
    "pg_table_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_uint64, _vpp]),
    "pg_table_create_dense": (C.c_int, [_vp, C.c_int, C.c_int, C.c_uint64, C.c_double, _vpp]),
    "pg_table_bytes_for_dense": (C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_double, _u64p]),
    "pg_table_destroy": (C.c_int, [_vp]),
    "pg_table_clear": (C.c_int, [_vp]),
    "pg_table_insert_seqset": (C.c_int, [_vp, C.c_int, _vp]),
    "pg_table_insert_seqset_min": (C.c_int, [_vp, C.c_int, _vp, C.c_uint32]),
    "pg_table_update_seqset": (C.c_int, [_vp, C.c_int, _vp]),
    "pg_table_insert_keys": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_uint64]),
    "pg_table_load_kmc1": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t]),
    "pg_table_load_kmc": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t]),
    "pg_kmc_kmer_length": (C.c_int, [_vp, C.c_size_t, _u32p]),
    "pg_table_stats": (C.c_int, [_vp, _u64p, _u64p, _u64p, _u64p]),
    "pg_table_rehash": (C.c_int, [_vp, C.c_double]),
    "pg_table_spill": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "pg_table_measure_spill": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "pg_sketch_create": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "pg_sketch_add_seqset": (C.c_int, [_vp, _vp]),
    "pg_sketch_estimate": (C.c_int, [_vp, _u64p]),
    "pg_sketch_registers": (C.c_int, [_vp, _vp]),
    "pg_sketch_estimate_registers": (C.c_int, [_vp, _u64p]),
    "pg_sketch_reset": (C.c_int, [_vp]),
    "pg_table_bytes_for": (C.c_int, [C.c_int, C.c_int, C.c_uint64, _u64p]),
    "pg_sketch_destroy": (C.c_int, [_vp]),
    "pg_table_export": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_uint64, _u64p]),
    "pg_table_k": (C.c_int, [_vp]),
    "pg_table_ngenomes": (C.c_int, [_vp]),
    "pg_table_minimizer": (C.c_int, [_vp]),
    "pg_table_set_minimizer": (C.c_int, [_vp, C.c_int]),
    "pg_minimizer_length": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int]),
    "pg_minimizer_length_for": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int]),
    "pg_minimizer_length_dense": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_double]),
    "pg_table_set_coscheduled": (C.c_int, [_vp, C.c_int]),
    "pg_seqset_create": (C.c_int, [_vp, C.c_uint32, _vp, _vpp]),
    "pg_seqset_destroy": (C.c_int, [_vp]),
    "pg_seqset_load_host": (C.c_int, [_vp, C.c_uint32, _vp, C.c_uint64]),
    "pg_seqset_load_dev": (C.c_int, [_vp, C.c_uint32, _vp, C.c_uint64]),
    "pg_seqset_total_kmers": (C.c_uint64, [_vp, C.c_int]),
    "pg_seqset_from_fasta": (C.c_int, [_vp, _vp, C.c_uint64, _vpp]),
    "pg_seqset_concat": (C.c_int, [_vp, _vp, C.c_uint32, _vpp]),
    "pg_seqset_ncontigs": (C.c_uint32, [_vp]),
    "pg_seqset_unpack": (C.c_int, [_vp, C.c_uint32, _vp]),
    "pg_seqset_contig": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_char_p), _u64p]),
    "pg_seqset_describe": (C.c_int, [_vp, _vp, _vp, C.c_uint64, _u64p]),
    "pg_result_create": (C.c_int, [_vp, _vp, C.c_uint32, _vpp]),
    "pg_result_create_ex": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, _vpp]),
    "pg_result_create_rows": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, _vpp]),
    "pg_result_destroy": (C.c_int, [_vp]),
    "pg_anchor_run": (C.c_int, [_vp]),
    "pg_anchor_run_range": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "pg_result_columns_direct": (C.c_int, [_vp, C.c_uint32]),
    "pg_anchor_run_columns_range": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "pg_result_timing_reset": (C.c_int, [_vp]),
    "pg_result_timing_mean": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), _u32p]),
    "pg_result_fused_runs": (C.c_int, [_vp, _u32p]),
    "pg_result_columns_bytes_range": (C.c_uint64, [_vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "pg_result_extract_columns_range": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "pg_result_merge_columns_range": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64]),
    "pg_result_coschedule_ranges": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.c_uint32]),
    "pg_result_coschedule_classes": (C.c_int, [_vp, _vp, _vp, C.c_uint32]),
    "pg_seqset_concat_ranges": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, _vpp]),
    "pg_seqset_slice": (C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp, _vp, _vpp]),
    "pg_rows_epilogue": (C.c_int, [_vp]),
    "pg_result_timing": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "pg_result_contig_info": (C.c_int, [_vp, C.c_uint32, _u64p, _u64p, _u32p, _u32p]),
    "pg_result_coschedule": (C.c_int, [_vp, _vp, C.c_uint32]),
    "pg_result_contig_colsums": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp]),
    "pg_result_columns_bytes": (C.c_uint64, [_vp, C.c_uint32]),
    "pg_result_extract_columns": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp]),
    "pg_result_merge_columns": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32]),
    "pg_result_window_stats": (C.c_int, [_vp, C.c_uint32, C.c_int, C.c_uint32, _vp, _vp, _vp, _vp]),
    "pg_result_write_bgzf": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]),
    "pg_result_write_bgzf_range": (C.c_int, [_vp, C.c_int, C.c_uint32, C.c_uint32, C.c_char_p, C.c_char_p, C.c_int, C.c_int]),
    "pg_result_download": (C.c_int, [_vp, C.c_uint32, _vp, _vp, _vp]),
    "pg_result_contigs_small": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, _vp, C.c_uint64]),
    "pg_write_bins_tsv": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "pg_result_colsums": (C.c_int, [_vp, _vp]),
    "pg_result_device_ptrs": (C.c_int, [_vp, _vpp, _u64p, _vpp, _u64p]),
    "pg_anchor_contig": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _u64p]),
    "pg_counters_for_read": (C.c_int, [_vp, C.c_int, _vp, C.c_uint64, _vp]),
    "pg_bgzf_open": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _vpp]),
    "pg_bgzf_write": (C.c_int, [_vp, _vp, C.c_size_t]),
    "pg_bgzf_close": (C.c_int, [_vp, C.c_char_p]),
}


def _share_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HIP runtimes in one process do not
    share the device: whichever initialises second finds "no HIP GPUs".  The multi-GPU modes hand torch tensors and
    streams to this library (torch.distributed over RCCL), so both must sit on ONE runtime whatever the import
    order: when torch is installed and not yet imported, its bundled runtime is loaded first (by path, globally), and
    the dynamic linker then resolves this library's libamdhip64 dependency to that copy — as it does when torch was
    imported first.  Without torch installed the system runtime (/opt/rocm) is used."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        libdir = os.path.join(os.path.dirname(spec.origin), "lib") if spec and spec.origin else None
    except (ImportError, ValueError):
        libdir = None
    if not libdir:
        return
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m panagram_amd.build` "
            "(hipcc --offload-arch=gfx950).  panagram_amd has no CPU fallback.")
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise PanagramHipError(rc, load().pg_last_error().decode("utf-8", "replace"))
