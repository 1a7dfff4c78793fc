"""Convenience calls over the C ABI for tests, the benchmark and the multi-GPU layer.

Every function here is a few lines of argument marshalling around ONE ``gdf_*`` entry point of
libgdf.so -- the computation happens in the HIP library, never in Python.  Join index columns are
allocated by the library (reference ownership rule, join_compute_api.h:525-548); they are copied into
torch tensors with a device-to-device hipMemcpy and released with ``gdf_column_free``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._binding import GDF_UNSUPPORTED_METHOD, GDFError, gdf_column, libgdf
from .columns import (GDF_HASH, GDF_HASH_MURMUR3, GDF_SORT, GDF_TO_NP, Column, column_array, new_context)

_hip = None


def _hipMemcpyDtoD(dst_ptr: int, src_ptr: int, nbytes: int):
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so.7")     # already in the process (libgdf.so links it): same handle
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpy.restype = C.c_int
    rc = _hip.hipMemcpy(dst_ptr, src_ptr, nbytes, 3)   # hipMemcpyDeviceToDevice
    if rc != 0:
        raise RuntimeError(f"hipMemcpy failed with {rc}")


def _take_library_column(col: gdf_column, torch_dtype):
    """Copy a library-allocated column into a torch tensor and free the library's buffer."""
    import torch
    n = int(col.size)
    out = torch.empty(n, dtype=torch_dtype, device="cuda")
    if n:
        _hipMemcpyDtoD(out.data_ptr(), col.data, n * out.element_size())
    libgdf.gdf_column_free(C.byref(col))
    return out


def _int_array(values):
    return (C.c_int * len(values))(*values)


class LibraryIndexColumn:
    """A join index column still owned by the library (int32, device memory): ``numel()`` is free, ``tensor()``
    copies it into a torch tensor on first use; the library buffer is released when this object dies."""

    def __init__(self, col: gdf_column):
        self._col = col
        self._tensor = None

    def numel(self) -> int:
        return int(self._col.size) if self._tensor is None else int(self._tensor.numel())

    def tensor(self):
        import torch
        if self._tensor is None:
            self._tensor = _take_library_column(self._col, torch.int32)
            self._col = None
        return self._tensor

    def __del__(self):
        if getattr(self, "_col", None) is not None and self._col.data:
            libgdf.gdf_column_free(C.byref(self._col))
            self._col = None


def join(left, right, left_on=None, right_on=None, how="inner", method=GDF_HASH, copy=True):
    """gdf_{inner,left,full}_join over lists of Column -> (left_idx, right_idx) int32 tensors, or, with
    ``copy=False``, two :class:`LibraryIndexColumn` that leave the pairs in the library's buffers."""
    import torch
    left_on = list(range(len(left))) if left_on is None else list(left_on)
    right_on = list(range(len(right))) if right_on is None else list(right_on)
    fn = {"inner": libgdf.gdf_inner_join, "left": libgdf.gdf_left_join, "full": libgdf.gdf_full_join}[how]
    ctx = new_context(method=method)
    li, ri = gdf_column(), gdf_column()
    la, ra = column_array(left), column_array(right)
    fn(la, len(left), _int_array(left_on), ra, len(right), _int_array(right_on), len(left_on), 0, None,
       C.byref(li), C.byref(ri), C.byref(ctx))
    if not copy:
        return LibraryIndexColumn(li), LibraryIndexColumn(ri)
    return _take_library_column(li, torch.int32), _take_library_column(ri, torch.int32)


_GROUPBY = {"sum": "gdf_group_by_sum", "min": "gdf_group_by_min", "max": "gdf_group_by_max",
            "avg": "gdf_group_by_avg", "count": "gdf_group_by_count"}


def group_by(op, keys, values, out_dtype: int | None = None, sort_result=False, method=GDF_HASH, capacity=None,
             with_indices=False, distinct=False, presorted=False, with_masks=False):
    """gdf_group_by_<op> -> (list of key tensors, aggregate tensor), trimmed to the number of groups.

    Outputs are preallocated by the caller with capacity N rows, as the reference's tests do
    (tests/groupby/groupby-test.cu:127).  ``with_indices`` also passes ``out_col_indices`` (size_t row
    numbers, filled by the GDF_SORT method only) and returns it as a third int64 tensor.  ``with_masks`` gives
    every output column a validity buffer and returns the aggregate's valid bits as a bool tensor (the
    mask-aware HASH group-by, an extension over the reference).
    """
    import torch
    n = keys[0].size if capacity is None else capacity
    torch_of = {1: torch.int8, 2: torch.int16, 3: torch.int32, 4: torch.int64, 5: torch.float32, 6: torch.float64,
                7: torch.int32, 8: torch.int64, 9: torch.int64}
    def mask():
        return torch.zeros(((n + 7) // 8 + 63) // 64 * 64 or 64, dtype=torch.uint8, device="cuda") if with_masks else None
    out_keys = [Column(torch.empty(max(n, 1), dtype=torch_of[k.c.dtype], device="cuda"), mask(), k.c.dtype, size=n) for k in keys]
    if out_dtype is None:
        out_dtype = values.c.dtype
    out_agg = Column(torch.empty(max(n, 1), dtype=torch_of[out_dtype], device="cuda"), mask(), out_dtype, size=n)
    out_idx = Column(torch.full((max(n, 1),), -1, dtype=torch.int64, device="cuda"), None, 4, size=n) if with_indices else None
    ctx = new_context(method=method, flag_sort_result=1 if sort_result else 0, flag_distinct=1 if distinct else 0,
                      flag_sorted=1 if presorted else 0)
    ka, oa = column_array(keys), column_array(out_keys)
    getattr(libgdf, _GROUPBY[op])(len(keys), ka, values.ptr, out_idx.ptr if with_indices else None, oa, out_agg.ptr, C.byref(ctx))
    g = out_agg.size
    if with_masks:
        bits = np.unpackbits(out_agg.valid.cpu().numpy(), bitorder="little")[:g].astype(bool)
        assert int(out_agg.c.null_count) == int(g - bits.sum())
        for k in out_keys:
            assert np.unpackbits(k.valid.cpu().numpy(), bitorder="little")[:g].all() and int(k.c.null_count) == 0
        return [k.data[:g] for k in out_keys], out_agg.data[:g], torch.from_numpy(bits)
    if with_indices:
        return [k.data[:g] for k in out_keys], out_agg.data[:g], out_idx.data[:out_idx.size]
    return [k.data[:g] for k in out_keys], out_agg.data[:g]


def order_by(cols):
    """gdf_order_by -> int64 tensor with the sorted row permutation (the library writes size_t)."""
    import torch
    n = cols[0].size
    arr = (gdf_column * len(cols))(*[c.c for c in cols])       # the entry point takes an ARRAY of structs
    d_cols = torch.empty(len(cols), dtype=torch.int64, device="cuda")
    d_types = torch.empty(len(cols), dtype=torch.int32, device="cuda")
    d_indx = torch.empty(max(n, 1), dtype=torch.int64, device="cuda")
    libgdf.gdf_order_by(n, arr, len(cols), d_cols.data_ptr(), d_types.data_ptr(), d_indx.data_ptr())
    assert d_cols.tolist() == [int(c.c.data or 0) for c in cols] and d_types.tolist() == [int(c.c.dtype) for c in cols]
    return d_indx[:n]


def hash_rows(cols, hash_func=GDF_HASH_MURMUR3):
    import torch
    n = cols[0].size
    out = Column(torch.empty(max(n, 1), dtype=torch.int32, device="cuda"), None, 3, size=n)
    libgdf.gdf_hash(len(cols), column_array(cols), hash_func, out.ptr)
    return out.data[:n]


def hash_partition(cols, cols_to_hash, num_partitions, hash_func=GDF_HASH_MURMUR3, with_masks=False):
    """gdf_hash_partition -> (list of output Columns, offsets list)."""
    import torch
    n = cols[0].size
    outs = []
    for c in cols:
        data = torch.empty_like(c.data)
        valid = torch.zeros_like(c.valid) if (with_masks and c.valid is not None) else None
        outs.append(Column(data, valid, c.c.dtype, size=n))
    offsets = (C.c_int * num_partitions)()
    libgdf.gdf_hash_partition(len(cols), column_array(cols), _int_array(cols_to_hash), len(cols_to_hash),
                              num_partitions, column_array(outs), offsets, hash_func)
    return outs, np.frombuffer(offsets, dtype=np.int32).tolist()       # (list(ctypes array) walks 12000 offsets in Python: 0.4 ms)


def shuffle_partition(keys: Column, num_partitions, row_base=0, narrow=None):
    """gdf_amd_shuffle_partition (include/gdf/gdf_amd_ext.h) -> (keys tensor, int32 row-number tensor, offsets list).
    ``narrow=(lo, hi)`` ships int64 keys as their int32 ``gdf_amd_narrow_keys`` image."""
    import torch
    n = keys.size
    out_k = Column(torch.empty(n, dtype=torch.int32 if narrow else keys.data.dtype, device=keys.data.device))
    out_r = Column(torch.empty(n, dtype=torch.int32, device=keys.data.device))
    offsets = (C.c_int * num_partitions)()
    lo, hi = narrow if narrow else (0, 0)
    libgdf.gdf_amd_shuffle_partition(keys.ptr, 1 if narrow else 0, int(lo), int(hi), int(row_base), num_partitions,
                                     out_k.ptr, out_r.ptr, offsets)
    return out_k.data, out_r.data, list(offsets)


def shuffle_partition_stable(keys: Column, num_partitions, narrow=None):
    """gdf_amd_shuffle_partition_stable -> (keys tensor, bitmaps int64 tensor [num_partitions, ceil(n / 64)], offsets).
    Partition p holds its keys in input order; bit i of bitmap p says that input row i went there.  With ``narrow`` =
    (lo, hi) the keys travel as int32 (key - lo) and rows whose key lies outside [lo, hi] -- they can join nothing --
    stay home: they are in no partition and in no bitmap, and the returned key tensor is shorter than the input."""
    import torch
    n = keys.size
    out_k = Column(torch.empty(n, dtype=torch.int32 if narrow else keys.data.dtype, device=keys.data.device))
    words = (n + 63) // 64
    bitmaps = torch.empty((num_partitions, max(words, 1)), dtype=torch.int64, device=keys.data.device)
    offsets = (C.c_int * (num_partitions + 1))()
    lo, hi = narrow if narrow else (0, 0)
    libgdf.gdf_amd_shuffle_partition_stable(keys.ptr, 2 if narrow else 0, int(lo), int(hi), num_partitions, out_k.ptr,
                                            bitmaps.data_ptr(), offsets)
    total = int(offsets[num_partitions]) if narrow else n
    return out_k.data[:total], bitmaps[:, :words], list(offsets)[:num_partitions]


class JoinBuild:
    """gdf_amd_join_build_* (include/gdf/gdf_amd_ext.h): the build relation partitioned once, probed many times."""

    def __init__(self, build):
        self._cols = list(build)                                   # the library reads the build DATA on every probe
        self._h = C.c_void_p()
        libgdf.gdf_amd_join_build_create(column_array(self._cols), len(self._cols), C.byref(self._h))

    def probe(self, probe, how="inner", copy=True):
        """-> (probe_idx, build_idx): what ``join(probe, build, how=how)`` returns with the table on ``build``."""
        import torch
        li, ri = gdf_column(), gdf_column()
        pa = column_array(probe)
        libgdf.gdf_amd_join_build_probe(self._h, {"inner": 0, "left": 1}[how], pa, len(probe), C.byref(li), C.byref(ri))
        if not copy:
            return LibraryIndexColumn(li), LibraryIndexColumn(ri)
        return _take_library_column(li, torch.int32), _take_library_column(ri, torch.int32)

    def accumulate(self, expected_rows):
        """gdf_amd_join_probe_begin: a probe relation added slice by slice and probed once.  Raises GDFError
        (GDF_UNSUPPORTED_METHOD) when this build side / key type cannot do it: probe the slices one by one then."""
        return ProbeAccumulator(self, expected_rows)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            libgdf.gdf_amd_join_build_free(self._h)
            self._h = None

    __del__ = close


class ProbeAccumulator:
    """gdf_amd_join_probe_* (include/gdf/gdf_amd_ext.h)."""

    def __init__(self, build: JoinBuild, expected_rows):
        self._build = build                                            # keeps the build side alive
        self._h = C.c_void_p()
        libgdf.gdf_amd_join_probe_begin(build._h, int(expected_rows), C.byref(self._h))

    def add(self, probe):
        libgdf.gdf_amd_join_probe_add(self._h, column_array(probe), len(probe))

    def add_recv(self, recv_keys, recv_fill, layout, position_base):
        """gdf_amd_fj_probe_add: one receive buffer of the probe relation (queued, not waited for); its tuples are numbered
        position_base + offset in the buffer."""
        self._keep = getattr(self, "_keep", []) + [recv_keys, recv_fill]
        libgdf.gdf_amd_fj_probe_add(self._h, recv_keys.data_ptr(), recv_fill.data_ptr(), layout.cap, int(position_base),
                                    layout.world * layout.block)

    def finish(self, copy=True):
        """-> (probe_idx, build_idx); probe rows are numbered across the slices in the order they were added."""
        import torch
        h, self._h = self._h, None                                     # _finish releases the object whatever it returns
        li, ri = gdf_column(), gdf_column()
        libgdf.gdf_amd_join_probe_finish(h, C.byref(li), C.byref(ri))
        if not copy:
            return LibraryIndexColumn(li), LibraryIndexColumn(ri)
        return _take_library_column(li, torch.int32), _take_library_column(ri, torch.int32)

    def __del__(self):
        if getattr(self, "_h", None) is not None and self._h.value:     # abandoned: finish into the void to free it
            try:
                li, ri = self.finish(copy=False)
                del li, ri
            except Exception:
                pass


# ---- fused multi-GPU join (include/gdf/gdf_amd_ext.h gdf_amd_fj_*) -----------------------------------------------------
FJ_DUMP_ELEMS = 32768          # one sender tile of dump space behind the regions (csrc/join.hip FJ_TILE)


class FjLayout:
    """What gdf_amd_fj_plan derives from GLOBAL numbers: partition bits of a rank's share and the room per region."""

    def __init__(self, world, fine_bits, coarse_bits, cap):
        self.world, self.fine_bits, self.coarse_bits, self.cap = world, fine_bits, coarse_bits, cap
        self.regions_per_rank = 8 << coarse_bits
        self.block = self.regions_per_rank * cap                 # elements every sender makes for one rank
        self.nregions = world * self.regions_per_rank


def fj_plan(world, build_rows_total, rows_max, rows_per_key=1.0):
    """-> FjLayout, or None when this world size / relation size does not fit the fused path."""
    fb, c1, cap = C.c_int(), C.c_int(), C.c_uint32()
    try:
        libgdf.gdf_amd_fj_plan(int(world), int(build_rows_total), int(rows_max), float(rows_per_key), C.byref(fb), C.byref(c1), C.byref(cap))
    except GDFError as e:
        if e.errcode == GDF_UNSUPPORTED_METHOD:
            return None
        raise
    return FjLayout(int(world), fb.value, c1.value, cap.value)


class FjRows:
    """Which row sits at which position of a send buffer, kept as gdf_amd_fj_send wrote it: out_pos[i] = the position row i's key
    went to.  ``materialize()`` inverts it on demand (global ids are resolved outside the timed path) into an int32 array of
    the buffer's shape with ``row_base + i`` at position out_pos[i] and -1 elsewhere."""

    def __init__(self, pos, row_base, total):
        self.pos, self.row_base, self.total = pos, int(row_base), int(total)

    def materialize(self):
        import torch
        dev = self.pos.device
        rows = torch.full((self.total,), -1, dtype=torch.int32, device=dev)
        p = self.pos.long() & 0xffffffff                          # the int32 storage holds uint32 positions
        ok = p < self.total                                       # 0xffffffff: the row was dropped (key outside [lo, hi])
        rows[p[ok]] = (torch.arange(self.pos.numel(), dtype=torch.int32, device=dev) + self.row_base)[ok]
        return rows


def fj_send(keys: Column, lo, hi, layout: FjLayout, row_base=0):
    """gdf_amd_fj_send -> (keys buffer int32 [world * block (+ dump)], FjRows (stays with the sender), fill counters int32
    [world * regions_per_rank (+ 1)], overflowed)."""
    import torch
    dev = keys.data.device
    total = layout.world * layout.block + FJ_DUMP_ELEMS
    out_keys = torch.empty(total, dtype=torch.int32, device=dev)
    out_pos = torch.empty(max(keys.size, 1), dtype=torch.int32, device=dev)[:keys.size]
    fill = torch.empty(layout.nregions + 1, dtype=torch.int32, device=dev)
    over = C.c_int(0)
    libgdf.gdf_amd_fj_send(keys.ptr, int(lo), int(hi), layout.world, layout.coarse_bits, layout.cap, out_keys.data_ptr(),
                           out_pos.data_ptr(), fill.data_ptr(), C.byref(over))
    return out_keys, FjRows(out_pos, row_base, total), fill, bool(over.value)


class FjBuild(JoinBuild):
    """gdf_amd_fj_build_create: the build side made from a receive buffer (world blocks, sender-major) and its fill counters."""

    def __init__(self, recv_keys, recv_fill, lo, layout: FjLayout, expected_rows):
        self._cols = [recv_keys, recv_fill]                          # kept alive while the library reads them
        self._h = C.c_void_p()
        libgdf.gdf_amd_fj_build_create(recv_keys.data_ptr(), recv_fill.data_ptr(), layout.world, int(lo), layout.fine_bits, layout.coarse_bits,
                                       layout.cap, int(expected_rows), C.byref(self._h))

    def probe(self, *a, **k):
        raise NotImplementedError("a receive-buffer build side is probed through accumulate() / add_recv()")


def prefixsum(col: Column, inclusive=True):
    import torch
    out = Column(torch.empty_like(col.data), None, col.c.dtype, size=col.size)
    libgdf.gdf_prefixsum_generic(col.ptr, out.ptr, 1 if inclusive else 0)
    return out.data


def comparison(lhs: Column, rhs, op: int):
    """gpu_comparison (column rhs) or gpu_comparison_static_* (python scalar tagged with a numpy dtype)."""
    import torch
    n = lhs.size
    out = Column(torch.empty(max(n, 1), dtype=torch.int8, device="cuda"),
                 torch.zeros(((n + 7) // 8 + 63) // 64 * 64 or 64, dtype=torch.uint8, device="cuda"), 1, size=n)
    if isinstance(rhs, Column):
        libgdf.gpu_comparison(lhs.ptr, rhs.ptr, out.ptr, op)
    else:
        suffix = {np.dtype(np.int8): "i8", np.dtype(np.int16): "i16", np.dtype(np.int32): "i32",
                  np.dtype(np.int64): "i64", np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[np.asarray(rhs).dtype]
        getattr(libgdf, "gpu_comparison_static_" + suffix)(lhs.ptr, np.asarray(rhs).item(), out.ptr, op)
    return out


def apply_stencil(lhs: Column, stencil: Column):
    import torch
    n = lhs.size
    out = Column(torch.empty_like(lhs.data), torch.zeros(((n + 7) // 8 + 63) // 64 * 64 or 64, dtype=torch.uint8, device="cuda"),
                 lhs.c.dtype, size=n)
    libgdf.gpu_apply_stencil(lhs.ptr, stencil.ptr, out.ptr)
    return out


def filter_rows(cols, values):
    """gdf_filter: indices (int64 tensor) of the rows whose every column equals the matching scalar."""
    import torch
    n, k = cols[0].size, len(cols)
    col_structs = (gdf_column * k)(*[c.c for c in cols])
    d_cols = torch.zeros(k, dtype=torch.int64, device="cuda")
    d_types = torch.zeros(k, dtype=torch.int32, device="cuda")
    val_bufs = [torch.from_numpy(np.asarray([v], dtype=GDF_TO_NP[c.c.dtype])).cuda() for c, v in zip(cols, values)]
    d_vals = torch.tensor([b.data_ptr() for b in val_bufs], dtype=torch.int64, device="cuda")
    d_indx = torch.empty(max(n, 1), dtype=torch.int64, device="cuda")
    new_sz = C.c_size_t(0)
    libgdf.gdf_filter(n, col_structs, k, d_cols.data_ptr(), d_types.data_ptr(), d_vals.data_ptr(), d_indx.data_ptr(),
                      C.byref(new_sz))
    return d_indx[: new_sz.value]


# ---- gdf_amd_dist_inner_join (include/gdf/gdf_amd_ext.h): the multi-GPU join behind the C ABI -----------------------------
class gdf_amd_transport(C.Structure):
    _A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p))
    _WAIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
    _ALLRED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_int)
    _DESTROY = C.CFUNCTYPE(None, C.c_void_p)
    _A2AV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p))
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("all_to_all", _A2A), ("wait", _WAIT),
                ("all_reduce_i64", _ALLRED), ("destroy", _DESTROY), ("all_to_all_v", _A2AV)]


class gdf_amd_dist_info(C.Structure):
    _fields_ = [("world", C.c_int), ("chunks", C.c_int), ("lo", C.c_int64), ("hi", C.c_int64), ("slice_rows", C.c_int64),
                ("fine_bits_p", C.c_int), ("coarse_bits_p", C.c_int), ("cap_p", C.c_uint32),
                ("fine_bits_b", C.c_int), ("coarse_bits_b", C.c_int), ("cap_b", C.c_uint32),
                ("block_p", C.c_int64), ("block_b", C.c_int64)]


class RcclTransport:
    """gdf_amd_rccl_transport_create: the library's own RCCL communicator (ncclSend / ncclRecv groups on its own stream).  The
    128-byte unique id comes from rank 0 (``RcclTransport.unique_id()``) over whatever channel the host has -- here
    torch.distributed's object broadcast (libgdf_amd/multigpu.py)."""

    def __init__(self, unique_id: bytes, world: int, rank: int):
        self._h = C.POINTER(gdf_amd_transport)()
        libgdf.gdf_amd_rccl_transport_create(C.c_char_p(unique_id), int(world), int(rank), C.byref(self._h))
        self.world, self.rank = int(world), int(rank)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        libgdf.gdf_amd_rccl_unique_id(buf)
        return buf.raw

    @property
    def ptr(self):
        return self._h

    def communicator_ranks(self):
        """(nranks, rank) as the RCCL communicator itself reports them (ncclCommCount / ncclCommUserRank)"""
        n, r = C.c_int(0), C.c_int(-1)
        libgdf.gdf_amd_rccl_transport_ranks(self._h, C.byref(n), C.byref(r))
        return int(n.value), int(r.value)

    def close(self):
        if getattr(self, "_h", None):
            libgdf.gdf_amd_transport_free(self._h)
            self._h = None

    __del__ = close


class CallbackTransport:
    """A gdf_amd_transport whose four functions are Python callbacks over a torch.distributed group of ANY backend: the blocks
    are staged through host memory (gdf_amd_copy), so ranks that share one GPU can talk through gloo
    (tests/test_gpu_multirank_one_gpu.py) -- the same C orchestration as over RCCL, another wire.  Synchronous: all_to_all
    returns when the exchange is done."""

    def __init__(self, group=None, with_all_to_all_v=True):
        import torch
        import torch.distributed as dist
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.errors = []

        def all_to_all_v(ctx, send, send_off, recv, recv_off, ticket):
            # the optional fifth function: exact sizes per peer (point-to-point messages through host memory, any backend)
            try:
                so = [int(send_off[i]) for i in range(self.world + 1)]
                ro = [int(recv_off[i]) for i in range(self.world + 1)]
                hs = torch.empty(max(so[-1], 1), dtype=torch.uint8)
                hr = torch.empty(max(ro[-1], 1), dtype=torch.uint8)
                if so[-1]:
                    libgdf.gdf_amd_copy(hs.data_ptr(), send, so[-1], 0)
                ops = []
                for r in range(self.world):
                    ns, nr = so[r + 1] - so[r], ro[r + 1] - ro[r]
                    if r == self.rank:
                        assert ns == nr
                        hr[ro[r]:ro[r + 1]].copy_(hs[so[r]:so[r + 1]])
                        continue
                    peer = dist.get_global_rank(self.group, r) if self.group is not None else r
                    if ns:
                        ops.append(dist.P2POp(dist.isend, hs[so[r]:so[r + 1]], peer, self.group))
                    if nr:
                        ops.append(dist.P2POp(dist.irecv, hr[ro[r]:ro[r + 1]], peer, self.group))
                for w in (dist.batch_isend_irecv(ops) if ops else []):
                    w.wait()
                if ro[-1]:
                    libgdf.gdf_amd_copy(recv, hr.data_ptr(), ro[-1], 1)
                ticket[0] = None
                return 0
            except Exception as e:                 # noqa: BLE001 -- a callback must not raise into C
                self.errors.append(e)
                return 1

        def all_to_all(ctx, send, recv, bytes_per_rank, ticket):
            try:
                n = int(bytes_per_rank) * self.world
                hs = torch.empty(max(n, 1), dtype=torch.uint8)
                hr = torch.empty(max(n, 1), dtype=torch.uint8)
                if n:
                    libgdf.gdf_amd_copy(hs.data_ptr(), send, n, 0)
                    dist.all_to_all_single(hr[:n], hs[:n], group=self.group)
                    libgdf.gdf_amd_copy(recv, hr.data_ptr(), n, 1)
                ticket[0] = None
                return 0
            except Exception as e:                 # noqa: BLE001 -- a callback must not raise into C
                self.errors.append(e)
                return 1

        def wait(ctx, ticket):
            return 0

        def all_reduce(ctx, values, count, op):
            try:
                t = torch.tensor([values[i] for i in range(count)], dtype=torch.int64)
                dist.all_reduce(t, op=(dist.ReduceOp.MIN, dist.ReduceOp.MAX, dist.ReduceOp.SUM)[op], group=self.group)
                for i in range(count):
                    values[i] = int(t[i])
                return 0
            except Exception as e:                 # noqa: BLE001
                self.errors.append(e)
                return 1

        self._cbs = (gdf_amd_transport._A2A(all_to_all), gdf_amd_transport._WAIT(wait), gdf_amd_transport._ALLRED(all_reduce),
                     gdf_amd_transport._A2AV(all_to_all_v))
        self._t = gdf_amd_transport(None, self.rank, self.world, self._cbs[0], self._cbs[1], self._cbs[2],
                                    C.cast(None, gdf_amd_transport._DESTROY),
                                    self._cbs[3] if with_all_to_all_v else C.cast(None, gdf_amd_transport._A2AV))

    @property
    def ptr(self):
        return C.pointer(self._t)

    def close(self):
        pass


_AGG_OPS = {"sum": 0, "min": 1, "max": 2, "avg": 3, "count": 4}      # gdf_agg_op (include/gdf/gdf.h)


def dist_group_by(op: str, keys: Column, values: Column, transport):
    """gdf_amd_dist_group_by (COLLECTIVE over the transport's ranks) -> (keys, aggregates) tensors with THIS rank's groups, sorted
    by key: sum / min / max in the value dtype, count int64, avg float64."""
    import torch
    ok, oa = gdf_column(), gdf_column()
    libgdf.gdf_amd_dist_group_by(_AGG_OPS[op], keys.ptr, values.ptr, transport.ptr, C.byref(ok), C.byref(oa))
    errs = getattr(transport, "errors", None)
    if errs:
        raise errs.pop(0)
    tdt = {1: torch.int8, 2: torch.int16, 3: torch.int32, 4: torch.int64, 5: torch.float32, 6: torch.float64, 7: torch.int32, 8: torch.int64,
           9: torch.int64}
    return _take_library_column(ok, tdt[int(ok.dtype)]), _take_library_column(oa, tdt[int(oa.dtype)])


def dist_group_by_multi(op: str, keys, values: Column, transport):
    """gdf_amd_dist_group_by_multi (COLLECTIVE): several key columns, validity masks honoured -> (list of key tensors, aggregate tensor,
    bool tensor of the aggregate's valid bits) with THIS rank's groups in ascending key order."""
    import torch
    oks = [gdf_column() for _ in keys]
    oa = gdf_column()
    oks_arr = (C.POINTER(gdf_column) * len(keys))(*[C.pointer(k) for k in oks])
    libgdf.gdf_amd_dist_group_by_multi(_AGG_OPS[op], len(keys), column_array(list(keys)), values.ptr, transport.ptr, oks_arr, C.byref(oa))
    errs = getattr(transport, "errors", None)
    if errs:
        raise errs.pop(0)
    tdt = {1: torch.int8, 2: torch.int16, 3: torch.int32, 4: torch.int64, 5: torch.float32, 6: torch.float64, 7: torch.int32, 8: torch.int64,
           9: torch.int64}
    g = int(oa.size)
    if oa.valid:
        raw = torch.empty((g + 7) // 8, dtype=torch.uint8, device="cuda")
        if g:
            _hipMemcpyDtoD(raw.data_ptr(), oa.valid, raw.numel())
        bits = torch.from_numpy(np.unpackbits(raw.cpu().numpy(), bitorder="little")[:g].astype(bool))
        assert int(oa.null_count) == int(g - int(bits.sum()))
    else:
        bits = torch.ones(g, dtype=torch.bool)
    return [_take_library_column(k, tdt[int(k.dtype)]) for k in oks], _take_library_column(oa, tdt[int(oa.dtype)]), bits


def dist_shuffle_join(probe: Column, build: Column, transport, how="inner"):
    """gdf_amd_dist_shuffle_[left_|full_]join (COLLECTIVE): the key-shuffle join behind ONE C call -> (probe ids, build ids), int64 tensors
    of (owner rank << 40 | local row) for every pair this rank produced; -1 names the missing side of an unmatched row (left / full)."""
    import torch
    op, ob = gdf_column(), gdf_column()
    fn = {"inner": libgdf.gdf_amd_dist_shuffle_join, "left": libgdf.gdf_amd_dist_shuffle_left_join, "full": libgdf.gdf_amd_dist_shuffle_full_join}[how]
    fn(probe.ptr, build.ptr, transport.ptr, C.byref(op), C.byref(ob))
    errs = getattr(transport, "errors", None)
    if errs:
        raise errs.pop(0)
    if not op.data:
        e = torch.empty(0, dtype=torch.int64, device=probe.data.device)
        return e, e.clone()
    return _take_library_column(op, torch.int64), _take_library_column(ob, torch.int64)


def dist_gather(ids, columns, transport):
    """gdf_amd_dist_gather (COLLECTIVE): the rows that global ids (int64 tensor: owner rank << 40 | local row, or -1) name, fetched from
    the ranks that own them -> list of (values tensor, bool tensor of valid bits) per column of this rank's shard."""
    import torch
    idc = Column(ids)
    outs = [gdf_column() for _ in columns]
    outs_arr = (C.POINTER(gdf_column) * len(columns))(*[C.pointer(o) for o in outs])
    libgdf.gdf_amd_dist_gather(idc.ptr, len(columns), column_array(list(columns)), transport.ptr, outs_arr)
    errs = getattr(transport, "errors", None)
    if errs:
        raise errs.pop(0)
    tdt = {1: torch.int8, 2: torch.int16, 3: torch.int32, 4: torch.int64, 5: torch.float32, 6: torch.float64, 7: torch.int32, 8: torch.int64,
           9: torch.int64}
    res = []
    for o in outs:
        n = int(o.size)
        raw = torch.empty((n + 7) // 8, dtype=torch.uint8, device="cuda")
        if n:
            _hipMemcpyDtoD(raw.data_ptr(), o.valid, raw.numel())
        bits = torch.from_numpy(np.unpackbits(raw.cpu().numpy(), bitorder="little")[:n].astype(bool))
        assert int(o.null_count) == int(n - int(bits.sum()))
        res.append((_take_library_column(o, tdt[int(o.dtype)]), bits))
    return res


def dist_inner_join(probe: Column, build: Column, transport, chunks=4):
    """gdf_amd_dist_inner_join -> None when every rank declined (the shape does not fit the fused path), else
    (probe_pos_of_rows, build_pos_of_rows, probe_indices, build_indices, info): the first two say where each LOCAL row's key went
    in its send buffer (they stay with the sender), the index columns are this rank's pairs as positions in its receive
    buffers (LibraryIndexColumn: the library's buffers, not copied)."""
    import torch
    dev = probe.data.device
    ppos = torch.empty(max(probe.size, 1), dtype=torch.int32, device=dev)[:probe.size]
    bpos = torch.empty(max(build.size, 1), dtype=torch.int32, device=dev)[:build.size]
    li, ri = gdf_column(), gdf_column()
    info = gdf_amd_dist_info()
    declined = C.c_int(1)
    libgdf.gdf_amd_dist_inner_join(probe.ptr, build.ptr, transport.ptr, int(chunks), ppos.data_ptr() if probe.size else None,
                                   bpos.data_ptr() if build.size else None, C.byref(li), C.byref(ri), C.byref(info), C.byref(declined))
    errs = getattr(transport, "errors", None)
    if errs:
        raise errs.pop(0)
    if declined.value:
        return None
    return ppos, bpos, LibraryIndexColumn(li), LibraryIndexColumn(ri), info
