"""ctypes loader of the C-ABI library (include/kmc_hip.h).

The library is built in-tree by ``kmc_amd/build.py`` (hipcc, gfx950).  There is
no CPU fallback: if the shared object is missing or a device cannot be opened
the caller gets an exception, never a silent slow path.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# KMC_AMD_LIB selects another build of the same library (profiling builds, scripts/stamps.py)
LIB_PATH = os.environ.get('KMC_AMD_LIB') or os.path.join(HERE, 'libkmc_hip.so')

# name -> (restype, argtypes); mirrors include/kmc_hip.h one to one
_ctx = C.c_void_p
_dp = C.POINTER(C.c_double)
_i8p = C.POINTER(C.c_int8)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_i64p = C.POINTER(C.c_int64)
_fp = C.POINTER(C.c_float)

SIGNATURES = {
    'kmc_create': (C.c_int, [C.POINTER(_ctx), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'kmc_destroy': (C.c_int, [_ctx]),
    'kmc_last_error': (C.c_char_p, [_ctx]),
    'kmc_sync': (C.c_int, [_ctx]),
    'kmc_last_kernel_ms': (C.c_int, [_ctx, _fp]),
    'kmc_run_kernel_times': (C.c_int, [_ctx, _fp, C.c_int, C.POINTER(C.c_int)]),
    'kmc_set_base_table': (C.c_int, [_ctx, _dp, C.c_double]),
    'kmc_set_hex_table': (C.c_int, [_ctx, _dp]),
    'kmc_get_hex_table': (C.c_int, [_ctx, _dp]),
    'kmc_generate_base_table': (C.c_int, [_ctx, _dp, C.c_double, C.c_double, _dp, _dp]),
    'kmc_set_lattice': (C.c_int, [_ctx, _i8p]),
    'kmc_get_lattice': (C.c_int, [_ctx, _i8p]),
    'kmc_fill_random': (C.c_int, [_ctx, _dp, C.c_uint64]),
    'kmc_set_replica_offset': (C.c_int, [_ctx, C.c_uint64]),
    'kmc_rates_full': (C.c_int, [_ctx]),
    'kmc_get_keys': (C.c_int, [_ctx, _u32p]),
    'kmc_set_keys': (C.c_int, [_ctx, _u32p]),
    'kmc_get_rates': (C.c_int, [_ctx, _dp]),
    'kmc_refresh_window': (C.c_int, [_ctx, _i32p]),
    'kmc_site_probability': (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _u32p, _dp]),
    'kmc_init_displacements': (C.c_int, [_ctx]),
    'kmc_set_displacements': (C.c_int, [_ctx, _i32p]),
    'kmc_get_displacements': (C.c_int, [_ctx, _i32p]),
    'kmc_run': (C.c_int, [_ctx, C.c_int64, C.c_int, _dp, C.c_uint64, _i32p]),
    'kmc_run_counts': (C.c_int, [_ctx, _i64p, C.c_int, C.c_uint64]),
    'kmc_get_scalars': (C.c_int, [_ctx, _dp, _dp, _dp, _i64p]),
    'kmc_set_time': (C.c_int, [_ctx, _dp]),
    'kmc_set_counters': (C.c_int, [_ctx, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'kmc_get_barrier_stats': (C.c_int, [_ctx, _dp]),
    'kmc_get_status': (C.c_int, [_ctx, _i32p, _i64p, _i64p]),
    'kmc_gamma_exact': (C.c_int, [_ctx, _dp]),
    'kmc_set_debug': (C.c_int, [_ctx, C.c_int, C.c_double]),
    'kmc_diffusion': (C.c_int, [_ctx, _dp, C.c_double, _dp, _dp, _dp]),
    'kmc_diffusion_noalloc': (C.c_int, [_ctx, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    'kmc_comm_unique_id': (C.c_int, [C.c_void_p]),
    'kmc_comm_create': (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    'kmc_comm_destroy': (C.c_int, [_ctx, C.c_void_p]),
    'kmc_allreduce_stats': (C.c_int, [_ctx, C.c_void_p, _dp, C.c_int]),
    'kmc_get_distances': (C.c_int, [_ctx, C.c_double, _dp]),
    'kmc_uses_lds_path': (C.c_int, [_ctx]),
    'kmc_lds_bytes': (C.c_int, [_ctx]),
    'kmc_table_is_compact': (C.c_int, [_ctx]),
    'kmc_set_hop_kernel': (C.c_int, [_ctx, C.c_int]),
    'kmc_hop_kernel_name': (C.c_char_p, [_ctx]),
    'kmc_set_launch_groups': (C.c_int, [_ctx, C.c_int]),
    'kmc_run_span_ms': (C.c_int, [_ctx, C.c_int, _fp, C.POINTER(C.c_int)]),
    'kmc_debug_stamps': (C.c_int, [_ctx, _i64p]),
}

_lib = None


class KMCLibraryError(RuntimeError):
    pass


def load():
    """Load libkmc_hip.so (once).  Raises KMCLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KMCLibraryError(
            "%s is missing: build it with `python -m kmc_amd.build` (hipcc, gfx950). "
            "kmc_amd has no CPU fallback." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise KMCLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise KMCLibraryError("%s does not export %s" % (LIB_PATH, name)) from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
