"""ctypes binding of libroaring_hip.so (the C ABI in include/roaring_hip.h).

The library is the product: if it is missing or no HIP device is usable, importing callers get
an exception -- there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libroaring_hip.so")
if os.environ.get("RHIP_LIB_VARIANT"):  # a diagnostic build beside the product library (croaring_amd/build.py)
    LIB_PATH = os.path.join(PKG, f"libroaring_hip_{os.environ['RHIP_LIB_VARIANT']}.so")


class RoaringHipError(RuntimeError):
    pass


class ClassStats(C.Structure):
    _fields_ = [("kernel", C.c_char_p), ("items", C.c_uint64), ("bytes_in", C.c_uint64), ("bytes_out", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("matched_pairs", C.c_uint64), ("passthrough", C.c_uint64), ("bytes_in", C.c_uint64),
                ("bytes_out", C.c_uint64), ("n_bitset_pairs", C.c_uint64), ("result_containers", C.c_uint64),
                ("ms_bitset_kernel", C.c_float), ("ms_total", C.c_float)]


class Partials(C.Structure):
    _fields_ = [("n_keys", C.c_uint64), ("d_keys", C.c_void_p), ("d_words", C.c_void_p), ("max_key", C.c_uint64),
                ("capacity", C.c_uint64)]


# every symbol include/roaring_hip.h declares: (name, restype, argtypes)
_vp, _sz, _u32, _u64, _i = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int
_pp = C.POINTER(C.c_char_p)
_psz = C.POINTER(C.c_size_t)
_pu32 = C.POINTER(C.c_uint32)
_pu64 = C.POINTER(C.c_uint64)
SYMBOLS = [
    ("rhip_ctx_create", _vp, [_i]),
    ("rhip_ctx_destroy", None, [_vp]),
    ("rhip_ctx_stream", _vp, [_vp]),
    ("rhip_ctx_synchronize", _i, [_vp]),
    ("rhip_last_error", C.c_char_p, []),
    ("rhip_version", C.c_char_p, []),
    ("rhip_pool_from_portable", _vp, [_vp, _sz, _vp, _vp]),
    ("rhip_pool_from_portable64", _vp, [_vp, _sz, _vp, _vp]),
    ("rhip_pool_from_blob", _vp, [_vp, _vp, _sz, _sz, _vp, _vp, _i]),
    ("rhip_pool_from_sorted_u32", _vp, [_vp, _sz, _vp, _vp]),
    ("rhip_pool_from_sorted_u64", _vp, [_vp, _sz, _vp, _vp]),
    ("rhip_pool_to_u32", _i, [_vp, _vp, _sz, _vp]),
    ("rhip_pool_to_u64", _i, [_vp, _vp, _sz, _vp]),
    ("rhip_pool_synth_bitset", _vp, [_vp, _u32, _u32, _u64]),
    ("rhip_synth_sparse_sizes", _i, [_u64, _u64, _sz, _vp]),
    ("rhip_synth_sparse_fill", _i, [_u64, _u64, _sz, _vp, _vp]),
    ("rhip_pool_free", None, [_vp]),
    ("rhip_pool_size", _u32, [_vp]),
    ("rhip_pool_containers", _u64, [_vp]),
    ("rhip_pool_is64", _i, [_vp]),
    ("rhip_pool_payload_bytes", _u64, [_vp]),
    ("rhip_pool_arena_bytes", _u64, [_vp]),
    ("rhip_pool_payload_align", _u32, [_vp]),
    ("rhip_pool_type_counts", _i, [_vp, _vp]),
    ("rhip_pool_portable_size", _sz, [_vp, _u32]),
    ("rhip_pool_portable_serialize", _sz, [_vp, _u32, _vp]),
    ("rhip_pool_cardinalities", _i, [_vp, _vp]),
    ("rhip_pool_portable_sizes", _i, [_vp, _sz, _vp, _vp]),
    ("rhip_pool_portable_serialize_many", _sz, [_vp, _sz, _vp, _vp, _sz, _vp]),
    ("rhip_pool_from_frozen", _vp, [_vp, _vp, _sz, _sz, _vp, _vp]),
    ("rhip_pool_frozen_sizes", _i, [_vp, _sz, _vp, _vp, _vp]),
    ("rhip_pool_frozen_serialize_many", _sz, [_vp, _sz, _vp, _vp, _sz, _vp, _vp]),
    ("rhip_pairwise", _vp, [_vp, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    ("rhip_pairwise_begin", _vp, [_vp, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    ("rhip_pairwise_end", _vp, [_vp]),
    ("rhip_pairwise_multi", _vp, [_vp, _sz, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    ("rhip_pairwise_multi_begin", _vp, [_vp, _sz, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    ("rhip_pairwise_cardinality", _i, [_vp, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    ("rhip_pairlist_create", _vp, [_vp, _vp, _vp, _sz, _vp, _vp]),
    ("rhip_pairlist_all_pairs", _vp, [_vp, _vp]),
    ("rhip_pairlist_successive", _vp, [_vp, _vp]),
    ("rhip_pairlist_size", _sz, [_vp]),
    ("rhip_pairlist_pairs", _i, [_vp, _vp, _vp]),
    ("rhip_pairlist_free", None, [_vp]),
    ("rhip_pairwise_list_begin", _vp, [_vp, _sz, _vp, _vp, _vp]),
    ("rhip_pairwise_list", _vp, [_vp, _sz, _vp, _vp, _vp]),
    ("rhip_pairwise_list_cardinality", _i, [_vp, _i, _vp, _vp]),
    ("rhip_pairwise_predicate", _i, [_vp, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    ("rhip_pairwise_inplace", _i, [_vp, _i, _vp, _vp, _sz, _vp, _vp]),
    ("rhip_pool_select", _vp, [_vp, _sz, _vp, _sz, _vp, _vp]),
    ("rhip_pool_run_optimize", _vp, [_vp, _vp]),
    ("rhip_pool_remove_run_compression", _vp, [_vp, _vp]),
    ("rhip_pool_flip", _vp, [_vp, _vp, _vp, _vp]),
    ("rhip_pool_max_key", _i, [_vp, _vp]),
    ("rhip_or_many", _vp, [_vp, _vp, _sz, _vp]),
    ("rhip_xor_many", _vp, [_vp, _vp, _sz, _vp]),
    ("rhip_or_many_heap", _vp, [_vp, _vp, _sz, _vp]),
    ("rhip_many_partials", _i, [_vp, _i, _vp, _sz, _vp, C.POINTER(Partials)]),
    ("rhip_partials_free", None, [_vp, C.POINTER(Partials)]),
    ("rhip_many_finalize", _vp, [_vp, _i, _i, _u64, _vp, _vp]),
    ("rhip_many_partials_dense", _i, [_vp, _i, _vp, _sz, _vp, _u64, _u32, _vp]),
    ("rhip_many_finalize_dense", _vp, [_vp, _i, _i, _u32, _u32, _u64, _vp]),
    ("rhip_many_sharded", _i, [_vp, _vp, _i, _vp, _sz, _vp, _u64, C.POINTER(_vp)]),
    ("rhip_last_stats", _i, [_vp, C.POINTER(Stats)]),
    ("rhip_ctx_set_timing", None, [_vp, _i]),
    ("rhip_ctx_set_class_stats", None, [_vp, _i]),
    ("rhip_last_class_stats", _i, [_vp, C.POINTER(ClassStats), _i]),
    ("rhip_debug_host_clock", _i, [_vp, C.POINTER(C.c_double), _i]),
    ("rhip_debug_last_placement", _i, [_vp, C.POINTER(C.c_float), _i]),
    ("rhip_debug_join_recovered", _u64, [_vp]),
    ("rhip_debug_plan_cached", _i, [_vp]),
    ("rhip_debug_fail_allocs", None, [_i, _i]),
    ("rhip_pairlist_drop_plans", _i, [_vp]),
    ("rhip_ctx_trim", _u64, [_vp]),
]

_lib = None


def load() -> C.CDLL:
    """Load libroaring_hip.so and bind every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RoaringHipError(
            f"{LIB_PATH} is missing: build it with `python -m croaring_amd.build` (hipcc, gfx950). "
            "croaring_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)  # RTLD_LOCAL: the CRoaring-named drop-ins must not interpose on a co-loaded libroaring
    for name, res, args in SYMBOLS:
        f = getattr(lib, name)  # AttributeError if the ABI drifted
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return (load().rhip_last_error() or b"").decode()
