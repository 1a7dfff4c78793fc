"""Shared helpers for the parity tests (counterpart of the reference's python/tests/utils.py)."""
import numpy as np

INT_DTYPES = [np.int8, np.int16, np.int32, np.int64]
FLT_DTYPES = [np.float32, np.float64]
ALL_DTYPES = INT_DTYPES + FLT_DTYPES


def gen_rand(dtype, size, low=None, high=None, positive_only=False):
    """reference utils.py:36-52: floats U(-1,1) (U(0,1) if positive_only); ints U[-10000,10000) clipped to the dtype."""
    dtype = np.dtype(dtype)
    if dtype.kind == "f":
        res = np.random.random(size=size).astype(dtype)
        return res if positive_only else (res * 2 - 1).astype(dtype)
    info = np.iinfo(dtype)
    lo = max(info.min, -10000 if low is None else low)
    hi = min(info.max, 10000 if high is None else high)
    return np.random.randint(lo, hi, size=size).astype(dtype)


def sort_pairs(l, r):
    l = np.asarray(l, dtype=np.int64)
    r = np.asarray(r, dtype=np.int64)
    order = np.lexsort((r, l))
    return l[order], r[order]


def sort_groups(keys, agg):
    """Sort group-by output rows lexicographically by key (first key most significant)."""
    keys = [np.asarray(k) for k in keys]
    order = np.lexsort(tuple(reversed(keys)))
    return [k[order] for k in keys], np.asarray(agg)[order]


def random_valid(n):
    """reference tests/join/valid_vectors.h:32-50: first half valid, second half coin flips."""
    v = np.ones(n, dtype=bool)
    v[n // 2:] = np.random.randint(0, 2, size=n - n // 2).astype(bool)
    return v
