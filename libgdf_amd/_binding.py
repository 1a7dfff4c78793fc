"""ctypes binding of libgdf.so / librmm.so (the C ABI of include/gdf/gdf.h and include/memory.h).

Mirrors /root/reference/libgdf/python/libgdf_cffi/wrapper.py:13-52: every ``gdf_*`` function that
returns a ``gdf_error`` is wrapped so that a non-zero code raises ``GDFError(<error name>)``; for
``GDF_CUDA_ERROR`` the message carries the runtime's own error name and string
(``gdf_cuda_last_error`` / ``gdf_cuda_error_name`` / ``gdf_cuda_error_string``).
"""
from __future__ import annotations

import ctypes as C
import os

LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")


class GDFError(Exception):
    def __init__(self, errcode, msg):
        self.errcode = errcode
        super().__init__(msg)


class RMMError(Exception):
    def __init__(self, errcode, msg):
        self.errcode = errcode
        super().__init__(msg)


class gdf_dtype_extra_info(C.Structure):
    _fields_ = [("time_unit", C.c_int)]


class gdf_column(C.Structure):
    """56-byte POD, reference include/gdf/cffi/types.h:84-92."""
    _fields_ = [
        ("data", C.c_void_p),
        ("valid", C.c_void_p),
        ("size", C.c_size_t),
        ("dtype", C.c_int),
        ("null_count", C.c_size_t),
        ("dtype_info", gdf_dtype_extra_info),
        ("col_name", C.c_char_p),
    ]


class gdf_context(C.Structure):
    """20-byte POD, reference include/gdf/cffi/types.h:161-167."""
    _fields_ = [
        ("flag_sorted", C.c_int),
        ("flag_method", C.c_int),
        ("flag_distinct", C.c_int),
        ("flag_sort_result", C.c_int),
        ("flag_sort_inplace", C.c_int),
    ]


class rmmOptions_t(C.Structure):
    _fields_ = [("allocation_mode", C.c_int), ("initial_pool_size", C.c_size_t), ("enable_logging", C.c_bool)]


_COLP = C.POINTER(gdf_column)
_COLPP = C.POINTER(_COLP)
_CTXP = C.POINTER(gdf_context)
_INTP = C.POINTER(C.c_int)

_JOIN = [_COLPP, C.c_int, _INTP, _COLPP, C.c_int, _INTP, C.c_int, C.c_int, _COLPP, _COLP, _COLP, _CTXP]
_GROUPBY = [C.c_int, _COLPP, _COLP, _COLP, _COLPP, _COLP, _CTXP]

# name -> (restype, argtypes); restype None means gdf_error (int, checked)
_PROTOTYPES = {
    "gdf_column_sizeof": (C.c_size_t, []),
    "gdf_column_view": (None, [_COLP, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "gdf_column_view_augmented": (None, [_COLP, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t]),
    "gdf_column_free": (None, [_COLP]),
    "gdf_column_concat": (None, [_COLP, _COLPP, C.c_int]),
    "get_column_byte_width": (None, [_COLP, _INTP]),
    "gdf_context_view": (None, [_CTXP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gdf_error_get_name": (C.c_char_p, [C.c_int]),
    "gdf_cuda_last_error": (C.c_int, []),
    "gdf_cuda_error_string": (C.c_char_p, [C.c_int]),
    "gdf_cuda_error_name": (C.c_char_p, [C.c_int]),
    "gdf_nvtx_range_push": (None, [C.c_char_p, C.c_int]),
    "gdf_nvtx_range_push_hex": (None, [C.c_char_p, C.c_uint]),
    "gdf_nvtx_range_pop": (None, []),
    "gdf_count_nonzero_mask": (None, [C.c_void_p, C.c_int, _INTP]),
    "gdf_validity_and": (None, [_COLP, _COLP, _COLP]),
    "gdf_inner_join": (None, _JOIN),
    "gdf_left_join": (None, _JOIN),
    "gdf_full_join": (None, _JOIN),
    "gdf_group_by_sum": (None, _GROUPBY),
    "gdf_group_by_min": (None, _GROUPBY),
    "gdf_group_by_max": (None, _GROUPBY),
    "gdf_group_by_avg": (None, _GROUPBY),
    "gdf_group_by_count": (None, _GROUPBY),
    "gdf_hash": (None, [C.c_int, _COLPP, C.c_int, _COLP]),
    "gdf_hash_partition": (None, [C.c_int, _COLPP, _INTP, C.c_int, C.c_int, _COLPP, _INTP, C.c_int]),
    "gdf_prefixsum_generic": (None, [_COLP, _COLP, C.c_int]),
    "gdf_prefixsum_i8": (None, [_COLP, _COLP, C.c_int]),
    "gdf_prefixsum_i32": (None, [_COLP, _COLP, C.c_int]),
    "gdf_prefixsum_i64": (None, [_COLP, _COLP, C.c_int]),
    "gpu_comparison_static_i8": (None, [_COLP, C.c_int8, _COLP, C.c_int]),
    "gpu_comparison_static_i16": (None, [_COLP, C.c_int16, _COLP, C.c_int]),
    "gpu_comparison_static_i32": (None, [_COLP, C.c_int32, _COLP, C.c_int]),
    "gpu_comparison_static_i64": (None, [_COLP, C.c_int64, _COLP, C.c_int]),
    "gpu_comparison_static_f32": (None, [_COLP, C.c_float, _COLP, C.c_int]),
    "gpu_comparison_static_f64": (None, [_COLP, C.c_double, _COLP, C.c_int]),
    "gpu_comparison": (None, [_COLP, _COLP, _COLP, C.c_int]),
    "gpu_apply_stencil": (None, [_COLP, _COLP, _COLP]),
    "gpu_concat": (None, [_COLP, _COLP, _COLP]),
    "gdf_ipc_parser_open": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "gdf_ipc_parser_open_recordbatches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),   # void in C; the value is ignored
    "gdf_ipc_parser_close": (C.c_int, [C.c_void_p]),                                        # void in C
    "gdf_ipc_parser_failed": (C.c_int, [C.c_void_p]),
    "gdf_ipc_parser_to_json": (C.c_char_p, [C.c_void_p]),
    "gdf_ipc_parser_get_error": (C.c_char_p, [C.c_void_p]),
    "gdf_ipc_parser_get_data": (C.c_void_p, [C.c_void_p]),
    "gdf_ipc_parser_get_data_offset": (C.c_int64, [C.c_void_p]),
    "gdf_ipc_parser_get_schema_json": (C.c_char_p, [C.c_void_p]),
    "gdf_ipc_parser_get_layout_json": (C.c_char_p, [C.c_void_p]),
    # include/gdf/gdf_amd_ext.h
    "gdf_amd_narrow_keys": (None, [_COLP, C.c_int64, C.c_int64, _COLP]),
    "gdf_amd_shuffle_partition": (None, [_COLP, C.c_int, C.c_int64, C.c_int64, C.c_int32, C.c_int, _COLP, _COLP, C.POINTER(C.c_int)]),
    "gdf_amd_shuffle_partition_stable": (None, [_COLP, C.c_int, C.c_int64, C.c_int64, C.c_int, _COLP, C.c_void_p, C.POINTER(C.c_int)]),
    "gdf_amd_join_build_create": (None, [C.POINTER(_COLP), C.c_int, C.POINTER(C.c_void_p)]),
    "gdf_amd_join_build_probe": (None, [C.c_void_p, C.c_int, C.POINTER(_COLP), C.c_int, _COLP, _COLP]),
    "gdf_amd_join_build_free": (C.c_int, [C.c_void_p]),                                       # void in C
    "gdf_amd_join_probe_begin": (None, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "gdf_amd_join_probe_add": (None, [C.c_void_p, C.POINTER(_COLP), C.c_int]),
    "gdf_amd_join_probe_finish": (None, [C.c_void_p, _COLP, _COLP]),
    "gdf_amd_fj_plan": (None, [C.c_int, C.c_int64, C.c_int64, C.c_double, _INTP, _INTP, C.POINTER(C.c_uint32)]),
    "gdf_amd_fj_send": (None, [_COLP, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, _INTP]),
    "gdf_amd_fj_build_create": (None, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_uint32, C.c_int64, C.POINTER(C.c_void_p)]),
    "gdf_amd_fj_probe_add": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int64, C.c_int64]),
    "gdf_amd_dist_inner_join": (None, [_COLP, _COLP, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _COLP, _COLP, C.c_void_p, _INTP]),
    "gdf_amd_dist_group_by": (None, [C.c_int, _COLP, _COLP, C.c_void_p, _COLP, _COLP]),
    "gdf_amd_dist_group_by_multi": (None, [C.c_int, C.c_int, C.POINTER(_COLP), _COLP, C.c_void_p, C.POINTER(_COLP), _COLP]),
    "gdf_amd_dist_shuffle_join": (None, [_COLP, _COLP, C.c_void_p, _COLP, _COLP]),
    "gdf_amd_dist_shuffle_left_join": (None, [_COLP, _COLP, C.c_void_p, _COLP, _COLP]),
    "gdf_amd_dist_shuffle_full_join": (None, [_COLP, _COLP, C.c_void_p, _COLP, _COLP]),
    "gdf_amd_dist_gather": (None, [_COLP, C.c_int, C.POINTER(_COLP), C.c_void_p, C.POINTER(_COLP)]),
    "gdf_amd_rccl_unique_id": (None, [C.c_char_p]),
    "gdf_amd_rccl_transport_create": (None, [C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
    "gdf_amd_rccl_transport_ranks": (None, [C.c_void_p, _INTP, _INTP]),
    "gdf_amd_transport_free": (C.c_int, [C.c_void_p]),                                        # void in C
    "gdf_amd_copy": (None, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "gdf_order_by": (None, [C.c_size_t, _COLP, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gdf_filter": (None, [C.c_size_t, _COLP, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                          C.POINTER(C.c_size_t)]),
}

_RMM_PROTOTYPES = {
    "rmmInitialize": (None, [C.POINTER(rmmOptions_t)]),
    "rmmFinalize": (None, []),
    "rmmGetErrorString": (C.c_char_p, [C.c_int]),
    "rmmAlloc": (None, [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]),
    "rmmRealloc": (None, [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]),
    "rmmFree": (None, [C.c_void_p, C.c_void_p]),
    "rmmGetAllocationOffset": (None, [C.POINTER(C.c_long), C.c_void_p, C.c_void_p]),
    "rmmGetInfo": (None, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p]),
    "rmmWriteLog": (None, [C.c_char_p]),
    "rmmLogSize": (C.c_size_t, []),
    "rmmGetLog": (None, [C.c_char_p, C.c_size_t]),
}

GDF_CUDA_ERROR = 1
GDF_UNSUPPORTED_METHOD = 12          # include/gdf/gdf.h gdf_error
GDF_COLUMN_SIZE_TOO_BIG = 4
GDF_INT64 = 4                        # include/gdf/gdf.h gdf_dtype


# LIBGDF_AMD_LAB=1: bind the LAB build of libgdf.so (lib/lab/, csrc/lab.h: experiment knobs compiled in and read from the
# environment) instead of the shipped one.  Only the tuning scripts under tools/gpu/ set it; the choice is made HERE, in the
# Python binding -- the shipped libgdf.so itself reads no environment variable.
# (LIBGDF_AMD_LAB=<subdirectory of lib/>: any other side-by-side build, e.g. the previous commit's kernels for an A/B on one box.)
LAB_BUILD = os.environ.get("LIBGDF_AMD_LAB", "")
LAB_BUILD = "" if LAB_BUILD == "0" else ("lab" if LAB_BUILD == "1" else LAB_BUILD)


def _load(name):
    path = os.path.join(LIB_DIR, LAB_BUILD, name) if (LAB_BUILD and name == "libgdf.so") else os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C libgdf_amd/csrc`).  libgdf_amd has no CPU fallback.")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


class _Wrapper:
    """Attribute access -> checked C call (reference wrapper.py:13-52)."""

    def __init__(self, cdll, prototypes, check):
        self._cdll = cdll
        self._prototypes = prototypes
        self._check = check
        self._cache = {}

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        fn = self._cache.get(name)
        if fn is not None:
            return fn
        try:
            cfn = getattr(self._cdll, name)
        except AttributeError:
            raise AttributeError(f"{name} is not exported by the library") from None
        restype, argtypes = self._prototypes.get(name, (None, None))
        if argtypes is not None:
            cfn.argtypes = argtypes
        if restype is None:
            cfn.restype = C.c_int
            check = self._check

            def wrapped(*args, _cfn=cfn, _name=name):
                rc = _cfn(*args)
                if rc != 0:
                    check(rc, _name)
                return rc

            fn = wrapped
        else:
            cfn.restype = restype
            fn = cfn
        self._cache[name] = fn
        return fn

    def raw(self, name):
        """The unchecked ctypes function (returns the error code instead of raising)."""
        cfn = getattr(self._cdll, name)
        restype, argtypes = self._prototypes.get(name, (None, None))
        if argtypes is not None:
            cfn.argtypes = argtypes
        cfn.restype = C.c_int if restype is None else restype
        return cfn


# LIBGDF_AMD_TESTHOOK=1 (set by tests/conftest.py and the stress / A-B tools, never by a caller of the library): load
# libgdf_testhook.so -- csrc/testhook.cpp, the registry behind force_path -- IN FRONT of libgdf.so, whose weak reference to the
# registry's lookup is bound at load time.  Without it libgdf.so has no path switch at all (include/gdf/gdf_amd_testhook.h).
TEST_HOOK = os.environ.get("LIBGDF_AMD_TESTHOOK", "") not in ("", "0")
_hook_cdll = _load("libgdf_testhook.so") if TEST_HOOK else None
_rmm_cdll = _load("librmm.so")
_gdf_cdll = _load("libgdf.so")

_gdf_cdll.gdf_error_get_name.restype = C.c_char_p
_gdf_cdll.gdf_error_get_name.argtypes = [C.c_int]
_gdf_cdll.gdf_cuda_error_name.restype = C.c_char_p
_gdf_cdll.gdf_cuda_error_name.argtypes = [C.c_int]
_gdf_cdll.gdf_cuda_error_string.restype = C.c_char_p
_gdf_cdll.gdf_cuda_error_string.argtypes = [C.c_int]
_rmm_cdll.rmmGetErrorString.restype = C.c_char_p
_rmm_cdll.rmmGetErrorString.argtypes = [C.c_int]


def _check_gdf(rc, fname):
    name = _gdf_cdll.gdf_error_get_name(rc).decode()
    if rc == GDF_CUDA_ERROR:
        code = _gdf_cdll.gdf_cuda_last_error()
        name = "CUDA ERROR. {}: {}".format(_gdf_cdll.gdf_cuda_error_name(code).decode(),
                                           _gdf_cdll.gdf_cuda_error_string(code).decode())
    raise GDFError(rc, name)


def _check_rmm(rc, fname):
    raise RMMError(rc, _rmm_cdll.rmmGetErrorString(rc).decode())


class _GdfWrapper(_Wrapper):
    """libgdf.so, plus the one name that lives in the test-hook library when a test process loaded it"""

    def __getattr__(self, name):
        if name == "gdf_amd_debug_force":
            if _hook_cdll is None:
                raise AttributeError("gdf_amd_debug_force lives in libgdf_testhook.so (test infrastructure): set LIBGDF_AMD_TESTHOOK=1 "
                                     "before importing libgdf_amd")
            fn = _hook_cdll.gdf_amd_debug_force
            fn.restype = C.c_int
            fn.argtypes = [C.c_char_p, C.c_char_p]
            return fn
        return super().__getattr__(name)


libgdf = _GdfWrapper(_gdf_cdll, _PROTOTYPES, _check_gdf)
librmm = _Wrapper(_rmm_cdll, _RMM_PROTOTYPES, _check_rmm)
