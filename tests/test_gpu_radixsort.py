"""-m gpu: gdf_radixsort_* and gdf_segmented_radixsort_* (csrc/sort.hip), replaying the reference's
python/tests/test_sorting.py:14-76 and test_segmented_sorting.py:15-104: keys of five dtypes with an int64 value
column 0..n-1, ascending and descending, expectation = numpy's STABLE argsort."""
import ctypes as C
import random
from itertools import product

import numpy as np
import pytest

from util import gen_rand

pytestmark = pytest.mark.gpu


def _api():
    from libgdf_amd._binding import _gdf_cdll as lib
    lib.gdf_radixsort_plan.restype = C.c_void_p
    lib.gdf_radixsort_plan.argtypes = [C.c_size_t, C.c_int, C.c_uint, C.c_uint]
    lib.gdf_segmented_radixsort_plan.restype = C.c_void_p
    lib.gdf_segmented_radixsort_plan.argtypes = [C.c_size_t, C.c_int, C.c_uint, C.c_uint]
    for n in ("gdf_radixsort_plan_setup", "gdf_segmented_radixsort_plan_setup"):
        getattr(lib, n).argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    for n in ("gdf_radixsort_plan_free", "gdf_segmented_radixsort_plan_free"):
        getattr(lib, n).argtypes = [C.c_void_p]
    lib.gdf_radixsort_generic.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gdf_segmented_radixsort_generic.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
    return lib


def _expected(key, descending):
    """test_sorting.py:23-39: mergesort (stable) on the key, or on ~key / -key for a descending sort."""
    if descending:
        neg = ~key if np.issubdtype(key.dtype, np.integer) else -key
        idx = np.argsort(neg, kind="mergesort")
    else:
        idx = np.argsort(key, kind="mergesort")
    return key[idx], idx


@pytest.mark.parametrize("nelem,descending,dtype", list(product([2, 3, 10, 11, 100, 1000, 5000, 300001], [True, False],
                                                                  [np.int8, np.int32, np.int64, np.float32, np.float64])))
def test_radixsort(gdf, nelem, descending, dtype):
    from libgdf_amd.columns import column_from_numpy
    lib = _api()
    key = gen_rand(dtype, nelem)
    ck, cv = column_from_numpy(key), column_from_numpy(np.arange(nelem, dtype=np.int64))
    plan = lib.gdf_radixsort_plan(nelem, int(descending), 0, key.dtype.itemsize * 8)
    assert lib.gdf_radixsort_plan_setup(plan, key.dtype.itemsize, 8) == 0
    assert lib.gdf_radixsort_generic(plan, C.addressof(ck.c), C.addressof(cv.c)) == 0
    assert lib.gdf_radixsort_plan_free(plan) == 0
    ek, ev = _expected(key, descending)
    np.testing.assert_array_equal(ck.to_numpy(), ek)
    np.testing.assert_array_equal(cv.to_numpy(), ev)


def _segsort_args():
    for nelem, descending, dtype in product([2, 3, 10, 100, 1000, 70000], [True, False], [np.int8, np.int32, np.int64, np.float32, np.float64]):
        for numseg in range(1, 4):
            if nelem // numseg > 0:
                yield nelem, numseg, descending, dtype


@pytest.mark.parametrize("nelem,num_segments,descending,dtype", list(_segsort_args()))
def test_segmented_radixsort(gdf, nelem, num_segments, descending, dtype):
    import torch
    from libgdf_amd.columns import column_from_numpy
    lib = _api()
    random.seed(nelem * 7 + num_segments)
    begins = np.asarray(sorted(random.sample(range(nelem), num_segments)), dtype=np.uint32)      # test_segmented_sorting.py:44-50
    ends = np.asarray(begins.tolist()[1:] + [nelem], dtype=np.uint32)
    key = gen_rand(dtype, nelem)
    ck, cv = column_from_numpy(key), column_from_numpy(np.arange(nelem, dtype=np.int64))
    db, de = torch.from_numpy(begins.view(np.int32)).cuda(), torch.from_numpy(ends.view(np.int32)).cuda()
    plan = lib.gdf_segmented_radixsort_plan(nelem, int(descending), 0, key.dtype.itemsize * 8)
    assert lib.gdf_segmented_radixsort_plan_setup(plan, key.dtype.itemsize, 8) == 0
    assert lib.gdf_segmented_radixsort_generic(plan, C.addressof(ck.c), C.addressof(cv.c), num_segments, db.data_ptr(), de.data_ptr()) == 0
    assert lib.gdf_segmented_radixsort_plan_free(plan) == 0
    got_k, got_v = ck.to_numpy(), cv.to_numpy()
    for s, e in zip(begins, ends):                                   # a segment at a time (:93-104)
        ek, ev = _expected(key[s:e], descending)
        np.testing.assert_array_equal(got_k[s:e], ek)
        np.testing.assert_array_equal(got_v[s:e], ev + s)
    np.testing.assert_array_equal(got_k[:begins[0]], key[:begins[0]])   # rows before the first segment are not touched
    np.testing.assert_array_equal(got_v[:begins[0]], np.arange(begins[0]))


def test_radixsort_argument_checks(gdf):
    from libgdf_amd.columns import column_from_numpy
    lib = _api()
    key, val = column_from_numpy(gen_rand(np.int32, 10)), column_from_numpy(np.arange(10, dtype=np.int64))
    plan = lib.gdf_radixsort_plan(11, 0, 0, 32)
    lib.gdf_radixsort_plan_setup(plan, 4, 8)
    assert lib.gdf_radixsort_generic(plan, C.addressof(key.c), C.addressof(val.c)) == 3      # GDF_COLUMN_SIZE_MISMATCH (sorting.cu:202)
    lib.gdf_radixsort_plan_free(plan)
    plan = lib.gdf_radixsort_plan(10, 0, 0, 32)
    lib.gdf_radixsort_plan_setup(plan, 8, 8)                                                  # wrong key size
    assert lib.gdf_radixsort_generic(plan, C.addressof(key.c), C.addressof(val.c)) == 3
    val32 = column_from_numpy(np.arange(10, dtype=np.int32))
    assert lib.gdf_radixsort_generic(plan, C.addressof(key.c), C.addressof(val32.c)) == 2     # GDF_UNSUPPORTED_DTYPE (:224)
    masked = column_from_numpy(gen_rand(np.int32, 10), np.ones(10, dtype=bool))
    lib.gdf_radixsort_plan_setup(plan, 4, 8)
    assert lib.gdf_radixsort_generic(plan, C.addressof(masked.c), C.addressof(val.c)) == 7    # GDF_VALIDITY_UNSUPPORTED
    lib.gdf_radixsort_plan_free(plan)
