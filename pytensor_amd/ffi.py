"""ctypes binding of ``include/pthip.h`` (``libpthip.so``).

There is no CPU fallback: if the library is missing, or a call fails, this module
raises.  ``lib()`` loads the in-tree ``pytensor_amd/libpthip.so`` built by
``__graft_entry__.build()``.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTHIP_LIB") or os.path.join(_HERE, "libpthip.so")  # (override: A/B builds)

DTYPE_CODE = {
    "bool": 0,
    "int8": 1,
    "int16": 2,
    "int32": 3,
    "int64": 4,
    "uint8": 5,
    "float32": 6,
    "float64": 7,
    "uint16": 8,
    "uint32": 9,
    "uint64": 10,
    "float16": 11,
}
REDUCE_CODE = {
    "Add": 0,
    "Mul": 1,
    "Maximum": 2,
    "ScalarMaximum": 2,
    "Minimum": 3,
    "ScalarMinimum": 3,
    "AND": 4,
    "OR": 5,
    "XOR": 6,
}

# every symbol include/pthip.h declares: name -> (restype, argtypes)
_vp, _i64, _sz, _int, _dbl = C.c_void_p, C.c_int64, C.c_size_t, C.c_int, C.c_double
_u32 = C.c_uint32
SIGNATURES = {
    "pthip_init": (_int, [_int]),
    "pthip_device_count": (_int, [C.POINTER(_int)]),
    "pthip_device_name": (_int, [C.c_char_p, _sz]),
    "pthip_last_error": (C.c_char_p, []),
    "pthip_synchronize": (_int, []),
    "pthip_stream": (_vp, []),
    "pthip_stream_select": (_int, [_int]),
    "pthip_stream_wait": (_int, [_int, _int]),
    "pthip_arena_set_no_reuse": (_int, [_vp, _int]),
    "pthip_alloc": (_int, [_sz, C.POINTER(_vp)]),
    "pthip_free": (_int, [_vp]),
    "pthip_pool_stats": (_int, [C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]),
    "pthip_pool_trim": (_int, []),
    "pthip_host_alloc": (_int, [_sz, C.POINTER(_vp)]),
    "pthip_host_free": (_int, [_vp]),
    "pthip_h2d": (_int, [_vp, _vp, _sz]),
    "pthip_d2h": (_int, [_vp, _vp, _sz]),
    "pthip_d2d": (_int, [_vp, _vp, _sz]),
    "pthip_memset": (_int, [_vp, _int, _sz]),
    "pthip_guard_protect": (_int, [_vp, _sz, C.POINTER(_int), C.POINTER(_vp)]),
    "pthip_guard_set_edges": (_int, [_int, _vp, _sz, _vp, _sz]),
    "pthip_guard_clean": (_int, [_int]),
    "pthip_guard_release": (_int, [_int]),
    "pthip_guard_stats": (_int, [C.POINTER(_int), C.POINTER(_int)]),
    "pthip_arena_begin": (_int, [C.POINTER(_vp)]),
    "pthip_arena_end": (_int, []),
    "pthip_arena_destroy": (_int, [_vp]),
    "pthip_capture_begin": (_int, []),
    "pthip_capture_end": (_int, [C.POINTER(_vp)]),
    "pthip_graph_launch": (_int, [_vp]),
    "pthip_graph_launch_on": (_int, [_vp, _int]),
    "pthip_plan_replay": (_int, [_vp, _vp, _vp, _vp, _vp, _sz, _int]),
    "pthip_record_begin": (_int, []),
    "pthip_record_end": (_int, [C.POINTER(_vp), C.POINTER(_i64)]),
    "pthip_list_launch": (_int, [_vp, _int]),
    "pthip_list_destroy": (_int, [_vp]),
    "pthip_launch_count": (_i64, []),
    "pthip_plan_replay2": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _int]),
    "pthip_plan_replay3": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _sz, _int]),
    "pthip_plan_replay4": (_int, [_vp, _vp, _vp, _int]),
    "pthip_ticket_slot": (_int, [C.POINTER(_vp)]),
    "pthip_ticket_slots": (_int, [_int, C.POINTER(_vp)]),
    "pthip_join_signal": (_int, [_vp]),
    "pthip_join_probe": (_int, [_int, C.POINTER(_int)]),
    "pthip_graph_destroy": (_int, [_vp]),
    "pthip_event_create": (_int, [C.POINTER(_vp)]),
    "pthip_event_record": (_int, [_vp]),
    "pthip_event_synchronize": (_int, [_vp]),
    "pthip_event_elapsed_ms": (_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "pthip_event_destroy": (_int, [_vp]),
    "pthip_jit_compile": (
        _int,
        [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), _int, C.POINTER(_vp), C.POINTER(_sz), C.c_char_p, _sz],
    ),
    "pthip_buffer_free": (None, [_vp]),
    "pthip_module_load": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "pthip_module_unload": (_int, [_vp]),
    "pthip_module_get_function": (_int, [_vp, C.c_char_p, C.POINTER(_vp)]),
    "pthip_launch": (_int, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _sz]),
    "pthip_reduce_workspace": (_sz, [_int, _i64, _i64, _i64]),
    "pthip_reduce": (_int, [_int, _int, _int, _int, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _sz]),
    "pthip_gemv_workspace": (_sz, [_int, _i64, _i64, _i64, _i64]),
    "pthip_gemv": (_int, [_int, _i64, _i64, _dbl, _vp, _i64, _i64, _vp, _i64, _dbl, _vp, _i64, _vp, _vp, _sz]),
    "pthip_gemv_finish": (_int, [_int, _i64, _i64, _vp, _dbl, _dbl, _vp, _i64, _vp]),
    "pthip_gemm": (
        _int,
        [_int, _i64, _i64, _i64, _i64, _dbl, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _dbl, _vp, _i64, _i64, _i64, _vp],
    ),
    "pthip_gemm_nslabs": (_i64, [_i64, _i64, _i64, _i64]),
    "pthip_gemm_partials": (_int, [_int, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64]),
    "pthip_pack_b16": (_int, [_int, _i64, _i64, _vp, _i64, _i64, _vp]),
    "pthip_ger": (_int, [_int, _i64, _i64, _dbl, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "pthip_potrf": (_int, [_int, _int, _i64, _i64, _vp, _vp]),
    "pthip_potrf_trsv": (_int, [_int, _i64, _i64, _vp, _vp, _vp, _vp]),
    "pthip_getrf": (_int, [_int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _int]),
    "pthip_lu_factor_finish": (_int, [_int, _i64, _i64, _vp, _vp, _vp]),
    "pthip_pivots_to_perm": (_int, [_int, _int, _i64, _i64, _vp, _vp]),
    "pthip_permuted_identity": (_int, [_int, _i64, _vp, _vp]),
    "pthip_eigh": (_int, [_int, _i64, _i64, _int, _vp, _vp, _vp]),
    "pthip_symmetrize": (_int, [_int, _i64, _i64, _int, _vp, _vp]),
    "pthip_arange": (_int, [_int, _i64, _dbl, _dbl, _i64, _i64, _vp]),
    "pthip_eye": (_int, [_int, _i64, _i64, _i64, _vp]),
    "pthip_sort": (_int, [_int, _i64, _i64, _vp, _vp, _vp]),
    "pthip_nonzero": (_int, [_i64, _vp, _vp, _vp]),
    "pthip_random": (_int, [_int, _int, _i64, _vp, _vp, _int, _vp, _vp, _vp, _vp]),
    "pthip_random_categorical": (_int, [_int, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
    "pthip_searchsorted": (_int, [_int, _i64, _vp, _vp, _int, _i64, _vp, _int, _vp]),
    "pthip_convolve1d": (_int, [_int, _i64, _vp, _i64, _vp, _int, _vp]),
    "pthip_convolve2d": (_int, [_int, _i64, _i64, _vp, _i64, _i64, _vp, _int, _vp]),
    "pthip_geqrf": (_int, [_int, _i64, _i64, _i64, _vp, _vp]),
    "pthip_orgqr": (_int, [_int, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _vp]),
    "pthip_svd_rows": (_int, [_int, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp]),
    "pthip_fill_null_rows": (_int, [_int, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _i64]),
    "pthip_triu": (_int, [_int, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _int, _int]),
    "pthip_gttrf": (_int, [_int, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "pthip_gttrs": (_int, [_int, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pthip_random_multinomial": (_int, [_int, _i64, _i64, _vp, _vp, _vp, _int, _i64, _vp, _i64, _vp]),
    "pthip_trsm": (_int, [_int, _int, _int, _int, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "pthip_copy_strided": (_int, [_int, _int, C.POINTER(_i64), _vp, C.POINTER(_i64), _vp, C.POINTER(_i64)]),
    "pthip_take_rows": (_int, [_int, _i64, _i64, _vp, _i64, _i64, _vp, _vp]),
    "pthip_scatter_rows_workspace": (_sz, [_i64, _i64, _i64]),
    "pthip_scatter_rows": (_int, [_int, _int, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _sz]),
    "pthip_pack": (_int, [_int, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_i64), _vp]),
    "pthip_multi_finish": (_int, [_int, _int, C.POINTER(_int), C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_int), C.POINTER(_vp)]),
    "pthip_softmax": (_int, [_int, _int, _i64, _i64, _vp, _vp]),
    "pthip_logsumexp_rows": (_int, [_int, _i64, _i64, _vp, _vp]),
    "pthip_logsumexp_rows_max": (_i64, [_int]),
    "pthip_colstat_workspace": (_sz, [_int, _i64, _i64, _i64]),
    "pthip_logsumexp_cols": (_int, [_int, _i64, _i64, _i64, _vp, _vp, _vp, _sz]),
    "pthip_softmax_cols": (_int, [_int, _int, _i64, _i64, _i64, _vp, _vp, _vp, _sz]),
    "pthip_cumulative": (_int, [_int, _int, _i64, _i64, _i64, _vp, _vp]),
    "pthip_imatmul": (_int, [_int, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp]),
    "pthip_argmax": (_int, [_int, _i64, _i64, _vp, _vp]),
    "pthip_comm_unique_id": (_int, [_vp]),
    "pthip_comm_init": (_int, [_int, _int, _vp]),
    "pthip_comm_size": (_int, [C.POINTER(_int), C.POINTER(_int)]),
    "pthip_comm_destroy": (_int, []),
    "pthip_all_reduce": (_int, [_int, _int, _i64, _vp]),
    "pthip_check_status": (_int, [C.POINTER(_int)]),
    "pthip_status_ptr": (_vp, []),
    "pthip_set_safe_mode": (_int, [_int]),
}


class HipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load ``libpthip.so`` (fails loudly — there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                "The hip linker has no CPU fallback."
            )
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(l, name)  # AttributeError if the ABI is incomplete
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def check(rc: int):
    if rc != 0:
        raise HipError(lib().pthip_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = C.c_int(0)
    lib().pthip_device_count(C.byref(n))
    return n.value


def init(device: int = 0):
    check(lib().pthip_init(device))


def jit_compile(src: str, name: str, opts=()) -> bytes:
    """HIP source → gfx950 code object (works without a GPU)."""
    l = lib()
    code = C.c_void_p()
    size = C.c_size_t()
    log = C.create_string_buffer(16384)
    arr = (C.c_char_p * max(1, len(opts)))(*[o.encode() for o in opts])
    rc = l.pthip_jit_compile(src.encode(), name.encode(), arr, len(opts), C.byref(code), C.byref(size), log, len(log))
    if rc != 0:
        raise HipError(l.pthip_last_error().decode("utf-8", "replace"))
    try:
        return C.string_at(code.value, size.value)
    finally:
        l.pthip_buffer_free(code)


def np_dtype_code(dt) -> int:
    return DTYPE_CODE[str(np.dtype(dt))]
