"""Column / context helpers -- the counterpart of the reference's test utilities
(/root/reference/libgdf/python/tests/utils.py:7-66): ``new_column``/``new_context`` make zeroed
structs, ``get_dtype`` maps numpy dtypes to ``gdf_dtype`` values, ``buffer_as_bits`` expands an
LSB-first validity mask.  Device buffers are torch ROCm tensors (the reference used numba device
arrays); only their ``data_ptr()`` crosses into the library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._binding import gdf_column, gdf_context, libgdf

# gdf_dtype values (include/gdf/gdf.h)
GDF_DTYPES = dict(GDF_invalid=0, GDF_INT8=1, GDF_INT16=2, GDF_INT32=3, GDF_INT64=4, GDF_FLOAT32=5, GDF_FLOAT64=6,
                  GDF_DATE32=7, GDF_DATE64=8, GDF_TIMESTAMP=9, GDF_CATEGORY=10, GDF_STRING=11, N_GDF_TYPES=12)
GDF_SORT, GDF_HASH = 0, 1
GDF_HASH_MURMUR3, GDF_HASH_IDENTITY = 0, 1
GDF_EQUALS, GDF_NOT_EQUALS, GDF_LESS_THAN, GDF_LESS_THAN_OR_EQUALS, GDF_GREATER_THAN, GDF_GREATER_THAN_OR_EQUALS = range(6)

# reference utils.py:19-28
NP_TO_GDF = {
    np.dtype(np.float64): 6, np.dtype(np.float32): 5, np.dtype(np.int64): 4, np.dtype(np.int32): 3,
    np.dtype(np.int16): 2, np.dtype(np.int8): 1, np.dtype(np.bool_): 1,
}
GDF_TO_NP = {1: np.int8, 2: np.int16, 3: np.int32, 4: np.int64, 5: np.float32, 6: np.float64,
             7: np.int32, 8: np.int64, 9: np.int64}


def get_dtype(dtype) -> int:
    return NP_TO_GDF[np.dtype(dtype)]


def new_column() -> gdf_column:
    return gdf_column()          # ctypes zero-initialises, like ffi.new('gdf_column*')


def new_context(flag_sorted=0, method=GDF_HASH, flag_distinct=0, flag_sort_result=0, flag_sort_inplace=0) -> gdf_context:
    ctx = gdf_context()
    libgdf.gdf_context_view(C.byref(ctx), flag_sorted, method, flag_distinct, flag_sort_result, flag_sort_inplace)
    return ctx


def buffer_as_bits(data: np.ndarray, nbits: int | None = None) -> list:
    """LSB-first expansion of a validity mask (reference utils.py:59-66)."""
    bits = np.unpackbits(np.asarray(data, dtype=np.uint8), bitorder="little").astype(bool)
    return list(bits[:nbits] if nbits is not None else bits)


def mask_from_bools(valid: np.ndarray) -> np.ndarray:
    """Pack a bool array into an LSB-first mask, padded to a multiple of 64 bytes (Arrow layout)."""
    packed = np.packbits(np.asarray(valid, dtype=bool), bitorder="little")
    padded = np.zeros(((len(packed) + 63) // 64) * 64 or 64, dtype=np.uint8)
    padded[: len(packed)] = packed
    return padded


_TORCH_TO_NP = None


def _torch_np_dtype(t):
    global _TORCH_TO_NP
    import torch
    if _TORCH_TO_NP is None:
        _TORCH_TO_NP = {torch.int8: np.int8, torch.int16: np.int16, torch.int32: np.int32, torch.int64: np.int64,
                        torch.float32: np.float32, torch.float64: np.float64, torch.uint8: np.uint8, torch.bool: np.bool_}
    return _TORCH_TO_NP[t.dtype]


class Column:
    """A ``gdf_column`` plus the torch tensors that own its device memory."""

    def __init__(self, data=None, valid=None, dtype: int | None = None, size: int | None = None, null_count: int = 0):
        self.data = data
        self.valid = valid
        self.c = gdf_column()
        if data is not None:
            if dtype is None:
                dtype = get_dtype(_torch_np_dtype(data))
            n = data.numel() if size is None else size
            libgdf.gdf_column_view_augmented(C.byref(self.c), data.data_ptr() if data.numel() else None,
                                             valid.data_ptr() if valid is not None else None, n, dtype, null_count)

    @property
    def ptr(self):
        return C.pointer(self.c)

    @property
    def size(self) -> int:
        return int(self.c.size)

    def to_numpy(self, n: int | None = None) -> np.ndarray:
        n = self.size if n is None else n
        return self.data[:n].cpu().numpy()

    def valid_bits(self, n: int | None = None) -> np.ndarray:
        n = self.size if n is None else n
        if self.valid is None:
            return np.ones(n, dtype=bool)
        return np.array(buffer_as_bits(self.valid.cpu().numpy(), n), dtype=bool)


def column_from_tensor(data, valid=None, dtype: int | None = None, null_count: int = 0) -> Column:
    return Column(data, valid, dtype, null_count=null_count)


def column_from_numpy(arr: np.ndarray, valid: np.ndarray | None = None, dtype: int | None = None, device="cuda") -> Column:
    """Upload a host array (and optional bool validity vector) -- test convenience."""
    import torch
    arr = np.ascontiguousarray(arr)
    t = torch.from_numpy(arr.view(np.int8) if arr.dtype == np.bool_ else arr).to(device)
    v = None
    nulls = 0
    if valid is not None:
        v = torch.from_numpy(mask_from_bools(valid)).to(device)
        nulls = int(len(valid) - np.count_nonzero(valid))
    return Column(t, v, dtype if dtype is not None else get_dtype(arr.dtype), null_count=nulls)


def column_array(cols):
    """gdf_column*[] for a list of Column objects (kept alive by the caller)."""
    arr = (C.POINTER(gdf_column) * len(cols))(*[c.ptr for c in cols])
    return arr
