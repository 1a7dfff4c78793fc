"""Python face of the CPU oracle (TEST INFRASTRUCTURE -- see gdf_oracle.c's header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Byte/integer algorithms live in gdf_oracle.c (gcc); element-wise expectations that the reference's own
Python tests state as numpy expressions (np.cumsum for prefix sums, python/tests/test_prefixsum.py:55;
``&`` of mask bytes, test_validity.py:59-74) are numpy here as well.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_DIR, "liboracle.so")

K_OF = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2, np.dtype(np.int64): 3,
        np.dtype(np.float32): 4, np.dtype(np.float64): 5, np.dtype(np.bool_): 0}
NP_OF_KIND = {0: np.int8, 1: np.int16, 2: np.int32, 3: np.int64, 4: np.float32, 5: np.float64}
OPS = {"sum": 0, "min": 1, "max": 2, "avg": 3, "count": 4}
JOINS = {"inner": 0, "left": 1, "full": 2}


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_DIR, "gdf_oracle.c")):
        subprocess.check_call(["make", "-C", _DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_murmur3_32.restype = C.c_uint32
        _lib.orc_murmur3_32.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_hash_combine.restype = C.c_uint32
        _lib.orc_hash_combine.argtypes = [C.c_uint32, C.c_uint32]
        _lib.orc_identity_hash.restype = C.c_uint32
        _lib.orc_identity_hash.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_join.restype = C.c_int64
        _lib.orc_group_by.restype = C.c_int64
        _lib.orc_free.argtypes = [C.c_void_p]
    return _lib


def _ptr_array(arrays):
    return (C.c_void_p * len(arrays))(*[a.ctypes.data if a is not None else None for a in arrays])


def _kinds(arrays):
    return (C.c_int * len(arrays))(*[K_OF[a.dtype] for a in arrays])


def _contig(arrays):
    return [np.ascontiguousarray(a) for a in arrays]


def murmur3_32(value: np.generic | np.ndarray) -> int:
    a = np.ascontiguousarray(value)
    return int(lib().orc_murmur3_32(a.ctypes.data, a.dtype.itemsize))


def hash_combine(l: int, r: int) -> int:
    return int(lib().orc_hash_combine(l, r))


def identity_hash(value) -> int:
    a = np.ascontiguousarray(value)
    return int(lib().orc_identity_hash(a.ctypes.data, K_OF[a.dtype]))


def hash_rows(cols, identity=False) -> np.ndarray:
    """gdf_hash expectation: uint32 row hash (hashing.cu:83-154)."""
    cols = _contig(cols)
    n = len(cols[0])
    out = np.empty(n, dtype=np.uint32)
    lib().orc_hash_rows(len(cols), _ptr_array(cols), _kinds(cols), C.c_int64(n), int(identity), out.ctypes.data_as(C.c_void_p))
    return out


def partition_ids(cols, nparts, identity=False) -> np.ndarray:
    cols = _contig(cols)
    n = len(cols[0])
    out = np.empty(n, dtype=np.uint32)
    lib().orc_partition_ids(len(cols), _ptr_array(cols), _kinds(cols), C.c_int64(n), int(identity), C.c_uint32(nparts),
                            out.ctypes.data_as(C.c_void_p))
    return out


def hash_partition(cols, cols_to_hash, nparts, identity=False):
    """Stable reference partitioning: (permutation, offsets).  The reference leaves the order inside a
    partition unspecified; tests compare partition CONTENTS as multisets plus the offsets."""
    pid = partition_ids([cols[i] for i in cols_to_hash], nparts, identity)
    perm = np.argsort(pid, kind="stable")
    counts = np.bincount(pid, minlength=nparts)
    offsets = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
    return perm, offsets, pid


def _mask_bytes(valid):
    return None if valid is None else np.ascontiguousarray(np.packbits(np.asarray(valid, dtype=bool), bitorder="little"))


def join(left, right, how="inner", left_valid=None, right_valid=None):
    """Sorted (l, r) index pairs of the equi-join on all given columns.  *_valid: per-column bool arrays or None."""
    left, right = _contig(left), _contig(right)
    nl, nr = len(left[0]), len(right[0])
    lv = [_mask_bytes(v) for v in (left_valid or [None] * len(left))]
    rv = [_mask_bytes(v) for v in (right_valid or [None] * len(right))]
    ol, orr = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
    n = lib().orc_join(JOINS[how], len(left), _kinds(left), _ptr_array(left), _ptr_array(lv), C.c_int64(nl),
                       _ptr_array(right), _ptr_array(rv), C.c_int64(nr), C.byref(ol), C.byref(orr))
    l = np.ctypeslib.as_array(ol, shape=(max(n, 1),))[:n].copy()
    r = np.ctypeslib.as_array(orr, shape=(max(n, 1),))[:n].copy()
    lib().orc_free(ol)
    lib().orc_free(orr)
    return l, r


def join_parallel_i64(probe, build, threads=0):
    """orc_join_parallel_i64: the all-cores CPU baseline (OpenMP radix-partitioned hash join, gdf_oracle.c) on one int64 key
    column -> (probe rows, build rows) UNSORTED, and the thread count used.  threads = 0: all the host offers."""
    probe = np.ascontiguousarray(probe, dtype=np.int64)
    build = np.ascontiguousarray(build, dtype=np.int64)
    L = lib()
    L.orc_join_parallel_i64.restype = C.c_int64
    L.orc_join_parallel_i64.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
    L.orc_max_threads.restype = C.c_int
    used = int(threads) if threads else int(L.orc_max_threads())
    cap = max(len(probe), 1)
    while True:
        ol, orr = np.empty(cap, dtype=np.int32), np.empty(cap, dtype=np.int32)
        n = L.orc_join_parallel_i64(probe.ctypes.data, len(probe), build.ctypes.data, len(build), ol.ctypes.data, orr.ctypes.data, cap, used)
        if n <= cap:
            return ol[:n], orr[:n], used
        cap = int(n)


def group_by(op, keys, values, out_dtype=None):
    """(sorted key arrays, aggregate array).  Aggregation in the input dtype; COUNT / AVG in out_dtype."""
    keys = _contig(keys)
    values = np.ascontiguousarray(values)
    n = len(keys[0])
    out_dtype = np.dtype(values.dtype if out_dtype is None else out_dtype)
    agg_dtype = out_dtype if op in ("count", "avg") else values.dtype
    out_keys = [np.empty(n, dtype=k.dtype) for k in keys]
    out_agg = np.empty(n, dtype=agg_dtype)
    g = lib().orc_group_by(OPS[op], len(keys), _kinds(keys), _ptr_array(keys), C.c_int64(n), values.ctypes.data_as(C.c_void_p),
                           K_OF[values.dtype], K_OF[out_dtype], _ptr_array(out_keys), out_agg.ctypes.data_as(C.c_void_p))
    return [k[:g] for k in out_keys], out_agg[:g]


# ---- gpu_hash_columns (src/hashops.cu:25-151) -----------------------------------------------------------
def fnv1a_rows(cols) -> np.ndarray:
    """64-bit FNV-1a over the little-endian bytes of each column's element, columns in order, one hash per row
    (as uint64).  The reference XORs every byte as a signed ``char`` (hashops.cu:46-75), so bytes >= 0x80 are
    sign-extended before the XOR; for 7-bit data this is the published FNV-1a."""
    cols = _contig(cols)
    n = len(cols[0])
    h = np.full(n, 14695981039346656037, dtype=np.uint64)
    prime = np.uint64(1099511628211)
    with np.errstate(over="ignore"):
        for c in cols:
            raw = c.view(np.uint8).reshape(n, c.dtype.itemsize)
            for b in range(c.dtype.itemsize):
                h ^= raw[:, b].astype(np.int8).astype(np.int64).view(np.uint64)
                h *= prime
    return h


# ---- validity-mask aware group-by (beyond the reference; BASELINE config C5) -------------
def group_by_masked(op, keys, values, key_valids=None, value_valid=None, out_dtype=None):
    """(sorted key arrays, aggregate, aggregate-valid bools) with pandas ``dropna=True`` semantics: rows with a
    null in any key column are dropped; null values are skipped; a group without a valid value reports 0 and
    valid=False (COUNT reports 0 and valid=True).  The arithmetic per group is the reference's
    (group_by above): aggregation in the input dtype, AVG = sum / (out dtype)count."""
    keys = _contig(keys)
    values = np.ascontiguousarray(values)
    n = len(values)
    kv = np.ones(n, dtype=bool)
    for v in (key_valids or []):
        if v is not None:
            kv &= np.asarray(v, dtype=bool)
    vv = kv if value_valid is None else (kv & np.asarray(value_valid, dtype=bool))
    all_keys, _ = group_by("count", [k[kv] for k in keys], values[kv], np.int64)
    sub_keys, sub_agg = group_by(op, [k[vv] for k in keys], values[vv], out_dtype)
    g = len(all_keys[0])
    agg = np.zeros(g, dtype=sub_agg.dtype)
    ok = np.zeros(g, dtype=bool)
    where = {tuple(k[i].item() for k in all_keys): i for i in range(g)}
    for j in range(len(sub_agg)):
        i = where[tuple(k[j].item() for k in sub_keys)]
        agg[i] = sub_agg[j]
        ok[i] = True
    if op == "count":
        ok[:] = True
    return all_keys, agg, ok


# ---- SORT method (sqls_ops.cu:1134-1289, 1373-1392; sqls_rtti_comp.hpp:299-320, 397-662) --------------
def order_by(cols) -> np.ndarray:
    """Row permutation that orders the rows lexicographically by (cols[0], cols[1], ...) with typed ``<``
    (LesserRTTI::less, sqls_rtti_comp.hpp:99-126; multi_col_order_by :299-320).  thrust::sort is unstable, so
    the order of equal rows is unspecified in the reference; this restatement (like the library) is stable."""
    return np.lexsort(tuple(np.ascontiguousarray(c) for c in reversed(list(cols)))).astype(np.int64)


def group_by_sort(op, keys, values, out_dtype=None, distinct=False):
    """GDF_SORT group-by: (key arrays in ascending order, aggregate, row index per group).

    sort (multi_col_order_by) -> gather of the aggregation column -> reduce_by_key with LesserRTTI::equal
    (sqls_rtti_comp.hpp:487-522).  SUM/MIN/MAX/AVG compute in the INPUT dtype, AVG = sum / (ValsT)count
    (:651-657, C++ division: truncating for integers); COUNT counts in the OUTPUT dtype (sqls_ops.cu:272-400);
    COUNT_DISTINCT stores the number of groups in element 0 and reports one row (rtti header :440-446).  The
    row index of a group is its LAST row in input order (reference known-answer: sqls_g_tester.cu:250-256)."""
    keys = _contig(keys)
    values = np.ascontiguousarray(values)
    n = len(keys[0])
    perm = order_by(keys)
    if n == 0:
        return [k[:0] for k in keys], values[:0], perm
    head = np.zeros(n, dtype=bool)
    head[0] = True
    for k in keys:
        ks = k[perm]
        head[1:] |= ~(ks[1:] == ks[:-1])             # typed ==: NaN never equal, -0.0 == +0.0
    starts = np.flatnonzero(head)
    ends = np.append(starts[1:], n)
    counts = ends - starts
    idx = perm[ends - 1]
    out_keys = [k[idx] for k in keys]
    vs = values[perm]
    T = values.dtype
    with np.errstate(over="ignore", invalid="ignore"):
        if op == "count":
            out_dtype = np.dtype(T if out_dtype is None else out_dtype)
            if distinct:
                return [k[:1] for k in out_keys], np.array([len(starts)]).astype(out_dtype), idx[:1]
            agg = counts.astype(out_dtype)
        elif op == "sum":
            agg = np.add.reduceat(vs, starts, dtype=T)
        elif op == "min":
            agg = np.minimum.reduceat(vs, starts)
        elif op == "max":
            agg = np.maximum.reduceat(vs, starts)
        elif op == "avg":
            sums = np.add.reduceat(vs, starts, dtype=T)
            c = counts.astype(T)
            if T.kind == "f":
                agg = (sums / c).astype(T)
            else:                                    # C++ integer division truncates toward zero
                si, ci = sums.astype(np.int64), c.astype(np.int64)
                q = np.where(ci != 0, np.abs(si) // np.maximum(np.abs(ci), 1), 0) * np.sign(si) * np.sign(ci)
                agg = q.astype(T)
        else:
            raise ValueError(op)
    return out_keys, agg, idx


# ---- numpy expectations -------------------------------------------------------------------------
def prefixsum(a: np.ndarray, inclusive=True) -> np.ndarray:
    """np.cumsum in the column dtype (wraps), reference python/tests/test_prefixsum.py:55."""
    inc = np.cumsum(a, dtype=a.dtype)
    if inclusive:
        return inc
    out = np.empty_like(inc)
    out[0:1] = 0
    out[1:] = inc[:-1]
    return out


_CMP = [np.equal, np.not_equal, np.less, np.less_equal, np.greater, np.greater_equal]


def comparison(lhs: np.ndarray, rhs, op: int) -> np.ndarray:
    """int8 stencil of op(lhs, rhs) under the usual arithmetic conversions (filterops.cu:17-75, with the
    reference's swapped </<= functors corrected -- SURVEY.md 8a quirk 1)."""
    l = np.asarray(lhs)
    r = np.asarray(rhs)
    common = np.result_type(l.dtype, r.dtype)
    # C promotes (int64, float32) to float32; numpy would pick float64
    if {l.dtype.kind, r.dtype.kind} == {"i", "f"}:
        common = l.dtype if l.dtype.kind == "f" else r.dtype
    return _CMP[op](l.astype(common), r.astype(common)).astype(np.int8)


def apply_stencil(lhs: np.ndarray, stencil: np.ndarray, stencil_valid=None) -> np.ndarray:
    keep = np.asarray(stencil) != 0
    if stencil_valid is not None:
        keep &= np.asarray(stencil_valid, dtype=bool)
    return np.asarray(lhs)[keep]


def filter_rows(cols, values) -> np.ndarray:
    keep = np.ones(len(cols[0]), dtype=bool)
    for c, v in zip(cols, values):
        keep &= np.asarray(c) == np.asarray(v, dtype=np.asarray(c).dtype)
    return np.nonzero(keep)[0].astype(np.uint64)


def count_nonzero_mask(mask_bytes: np.ndarray, nrows: int) -> int:
    return int(np.unpackbits(np.asarray(mask_bytes, dtype=np.uint8), bitorder="little")[:nrows].sum())


# ---- synthetic generators shared by tests and bench (pure functions of (seed, i)) -----------------
def splitmix64(x: np.ndarray) -> np.ndarray:
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))
