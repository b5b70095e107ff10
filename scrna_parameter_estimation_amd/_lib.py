"""ctypes binding of libmemento_hip.so (include/memento_hip.h).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible the product path raises.
torch is imported first so that its bundled HIP runtime (same SONAME, libamdhip64.so.7) is the one
both torch and this library use -- torch tensors then serve as plain device memory for the kernels.
"""

import ctypes
import os
from ctypes import c_double, c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmemento_hip.so")

_lib = None


class HipLibraryMissing(ImportError):
    pass


class MementoHipError(RuntimeError):
    pass


class ChainTiles(ctypes.Structure):
    """include/memento_hip.h: mm_chain_tiles"""
    _fields_ = [("d_tile_chain", c_void_p), ("d_ops", c_void_p), ("d_ch_base", c_void_p), ("d_ch_K", c_void_p), ("d_ch_nobs", c_void_p),
                ("d_ch_omq", c_void_p), ("d_ch_row", c_void_p), ("d_jump", c_void_p), ("d_w_dump", c_void_p), ("kmax_dump", c_int32)]


_SIGS = {
    "mm_version": ([], ctypes.c_int),
    "mm_device_count": ([], ctypes.c_int),
    "mm_set_device": ([ctypes.c_int], ctypes.c_int),
    "mm_malloc": ([ctypes.POINTER(c_void_p), c_size_t], ctypes.c_int),
    "mm_free": ([c_void_p], ctypes.c_int),
    "mm_memset": ([c_void_p, ctypes.c_int, c_size_t, c_void_p], ctypes.c_int),
    "mm_memcpy_h2d": ([c_void_p, c_void_p, c_size_t, c_void_p], ctypes.c_int),
    "mm_memcpy_d2h": ([c_void_p, c_void_p, c_size_t, c_void_p], ctypes.c_int),
    "mm_sync": ([c_void_p], ctypes.c_int),
    "mm_timer_create": ([ctypes.POINTER(c_void_p)], ctypes.c_int),
    "mm_timer_begin": ([c_void_p, c_void_p], ctypes.c_int),
    "mm_timer_end": ([c_void_p, c_void_p], ctypes.c_int),
    "mm_timer_elapsed_ms": ([c_void_p, ctypes.POINTER(c_float)], ctypes.c_int),
    "mm_timer_destroy": ([c_void_p], ctypes.c_int),
    "mm_debug_read_probe": ([c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "mm_csr_rowsum": ([c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_csr_colcount": ([c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "mm_csr_colsplit": ([c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_csr_mapcount": ([c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_csr_mapsplit": ([c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_csr_colsum": ([c_void_p, c_void_p, c_int64, c_void_p, c_void_p], ctypes.c_int),
    "mm_sell_count": ([c_void_p] * 5 + [c_int32, c_int32, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_sell_layout": ([c_void_p, c_int32, c_int32] + [c_void_p] * 8, ctypes.c_int),
    "mm_sell_scatter": ([c_void_p] * 5 + [c_int32, c_int32] + [c_void_p] * 5, ctypes.c_int),
    "mm_sell_split_count": ([c_void_p] * 4 + [c_int32, c_int64, c_int32, c_int32] + [c_void_p] * 4, ctypes.c_int),
    "mm_sell_scatter_ranges": ([c_void_p] * 5 + [c_int32, c_int32, c_int32, c_int64] + [c_void_p] * 7, ctypes.c_int),
    "mm_moments1d_sell": ([c_void_p] * 8 + [c_int32, c_int32] + [c_void_p] * 2, ctypes.c_int),
    "mm_moments1d_reduce": ([c_void_p] * 5 + [c_int32, c_int32] + [c_void_p] * 4, ctypes.c_int),
    "mm_hist1d_sell": ([c_void_p] * 9 + [c_int32, c_int32] + [c_void_p] * 5, ctypes.c_int),
    "mm_bins_count": ([c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_bins_order": ([c_void_p] * 5 + [c_int64, c_int32, c_int32, c_int32] + [c_void_p] * 13, ctypes.c_int),
    "mm_boot1d_replay": ([c_void_p] * 6 + [c_int64] + [c_void_p] * 4 + [ctypes.POINTER(c_uint64), c_int32, c_int32, c_int64,
                         c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p], ctypes.c_int),
    "mm_boot1d_free": ([c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_uint64), c_int32, c_int32, c_int64,
                       c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "mm_pcg64_stream": ([ctypes.POINTER(c_uint64), c_int64, c_void_p, c_void_p], ctypes.c_int),
    "mm_boot1d_chain": ([c_void_p] * 6 + [c_int64, c_void_p, ctypes.POINTER(c_uint64), c_int32, c_int32, c_int64,
                        c_void_p, c_void_p, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "mm_boot1d_async": ([c_void_p] * 6 + [c_int64, ctypes.POINTER(c_uint64), c_int32, c_int32, c_int64, c_void_p, c_void_p, c_void_p,
                        c_int32, c_void_p], ctypes.c_int),
    "mm_debug_wave_clock": ([c_void_p], ctypes.c_int),
    "mm_debug_replay_arith": ([c_int32], ctypes.c_int),
    "mm_debug_replay_ring": ([c_int32], ctypes.c_int),
    "mm_debug_replay_rows_mod": ([c_int64], ctypes.c_int),
    "mm_boot1d_fast": ([c_void_p] * 6 + [c_int64] + [c_void_p] * 4 + [c_uint64, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_boot_fill_log": ([c_void_p, c_void_p, c_int64, c_int64, c_int32, ctypes.POINTER(c_double), c_int32, c_uint64,
                          c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_contrast_stats": ([c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                           c_void_p], ctypes.c_int),
    "mm_contrast_rows": ([c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p],
                         ctypes.c_int),
    "mm_valid_cols": ([c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_residualize": ([c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_cross_resampled": ([c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_uint64, c_int64, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_extract_cols": ([c_void_p] * 6 + [c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_pair_cross": ([c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32,
                       c_void_p, c_int64, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_pair_hist": ([c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                      c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_pair_bins_count": ([c_void_p] * 6 + [c_int64, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "mm_bins_order2d": ([c_void_p] * 6 + [c_int64, c_int32, c_int32, c_int32] + [c_void_p] * 15, ctypes.c_int),
    "mm_boot2d_replay": ([c_void_p] * 7 + [c_int64] + [c_void_p] * 4 + [ctypes.POINTER(c_uint64), c_int32, c_int64, c_void_p, c_void_p],
                         ctypes.c_int),
    "mm_boot2d_replay_rec": ([c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_uint64), c_int32, c_int64,
                             c_void_p, c_void_p], ctypes.c_int),
    "mm_simulate": ([c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_uint64, c_uint64, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "mm_std_normal": ([c_uint64, c_int64, c_void_p, c_void_p], ctypes.c_int),
    "mm_contract_stats": ([c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                           c_void_p, c_void_p, c_void_p], ctypes.c_int),
}

EXPORTS = ["mm_last_error"] + list(_SIGS)


def load(require_gpu=True):
    """Load the library (once).  Raises HipLibraryMissing if it was not built; MementoHipError if
    ``require_gpu`` and no HIP device is visible."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found -- build it with `python -m scrna_parameter_estimation_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        import torch  # noqa: F401  (loads libamdhip64.so.7 first; see module docstring)

        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        lib.mm_last_error.restype = ctypes.c_char_p
        lib.mm_last_error.argtypes = []
        for name, (args, res) in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = res
        _lib = lib
    if require_gpu and _lib.mm_device_count() < 1:
        raise MementoHipError("no HIP device visible: the memento HIP path needs an MI355X (no CPU fallback)")
    return _lib


def call(name, *args):
    """Call an int-returning entry point and raise on a negative status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MementoHipError(f"{name} failed ({rc}): {lib.mm_last_error().decode()}")
    return rc
