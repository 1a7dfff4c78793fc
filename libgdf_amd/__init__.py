"""libgdf_amd -- Python host-side mirror of the reference's ``libgdf_cffi`` / ``librmm_cffi``.

The product is the pair of C-ABI shared libraries ``lib/libgdf.so`` and ``lib/librmm.so``
(built from ``csrc/`` for gfx950).  This package only *binds* them, the way
``/root/reference/libgdf/python/libgdf_cffi/__init__.py:14-31`` and ``wrapper.py:13-52`` do with
cffi: attribute access resolves a ``gdf_*`` symbol, a non-zero ``gdf_error`` is raised as
:class:`GDFError` carrying the error name.  cffi is not installed in this image, so the binding is
ctypes with explicit prototypes for the hot-path entry points (``include/gdf/gdf.h``).

There is deliberately NO fallback: if the libraries are missing or fail to load, importing
``libgdf_amd.libgdf`` raises -- nothing in this package computes on the CPU.

PyTorch is used by callers purely as a device-buffer provider (``tensor.data_ptr()``); import torch
before this package so that one HIP runtime (same soname, ``libamdhip64.so.7``) serves both.
"""
from __future__ import annotations

from ._binding import (  # noqa: F401
    GDFError,
    RMMError,
    gdf_column,
    gdf_context,
    libgdf,
    librmm,
    LIB_DIR,
)
from .columns import (  # noqa: F401
    GDF_DTYPES,
    NP_TO_GDF,
    Column,
    buffer_as_bits,
    column_from_tensor,
    get_dtype,
    mask_from_bools,
    new_column,
    new_context,
)
from . import api  # noqa: F401

__all__ = [
    "GDFError", "RMMError", "gdf_column", "gdf_context", "libgdf", "librmm", "LIB_DIR",
    "GDF_DTYPES", "NP_TO_GDF", "Column", "buffer_as_bits", "column_from_tensor", "get_dtype",
    "mask_from_bools", "new_column", "new_context", "api",
]
