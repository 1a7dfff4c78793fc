"""-m gpu: the join's radix partitioner in isolation (test hook gdf_amd_debug_partition).

Checks the invariants the probe kernels rely on: every joinable row appears exactly once, each tuple
carries its own key, and fine partition f holds exactly the rows whose mix64 bits say f -- repeated many
times because the failure this guards against (a barrier that did not drain LDS atomics, see
csrc/common.h block_sync) corrupted about one run in a hundred."""
import ctypes as C

import numpy as np
import pytest

from util import gen_rand

pytestmark = pytest.mark.gpu


def hash_a(raw):
    """numpy restatement of csrc/join.hip hash_a(): lowbias32(key_fold(raw key))."""
    raw = raw.astype(np.uint64)
    x = (raw & np.uint64(0xFFFFFFFF)).astype(np.uint32) ^ ((raw >> np.uint64(32)).astype(np.uint32) * np.uint32(0x9e3779b1))
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    return x


def _partition(gdf, col, n, fb):
    import torch
    lib = gdf._binding._gdf_cdll
    lib.gdf_amd_debug_partition.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    ok = torch.full((n,), -7, dtype=torch.int64, device="cuda")
    oi = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    off = (C.c_uint32 * ((1 << fb) + 1))()
    nj = C.c_uint32(0)
    info = (C.c_uint64 * 2)()
    assert lib.gdf_amd_debug_partition(C.byref(col.c), fb, ok.data_ptr(), oi.data_ptr(), off, C.byref(nj), info) == 0
    return ok.cpu().numpy(), oi.cpu().numpy(), np.array(list(off), dtype=np.int64), nj.value, (int(info[0]), int(info[1]))


@pytest.mark.parametrize("dtype,fb,n,reps,wide", [(np.int32, 2, 10000, 300, False), (np.int64, 2, 10000, 300, False),
                                                  (np.int64, 11, 3_000_000, 5, False), (np.int32, 9, 1_000_000, 10, False),
                                                  (np.int64, 2, 10000, 100, True), (np.int64, 10, 2_000_000, 5, True)])
def test_partition_invariants(gdf, dtype, fb, n, reps, wide):
    """wide=True spreads the int64 keys over more than 2^32 so the 12-byte (key64, row) tuple format is used;
    otherwise the keys fit 32 bits after subtracting the minimum and the packed 8-byte format is used."""
    from libgdf_amd.columns import column_from_numpy
    keys = gen_rand(dtype, n, low=0, high=2000 if n <= 10000 else 2_000_000)
    if wide:
        keys = keys * np.int64(1 << 33) - np.int64(5)
    col = column_from_numpy(keys)
    width_mask = np.uint64((1 << (8 * np.dtype(dtype).itemsize)) - 1)
    raw = keys.astype(np.int64).view(np.uint64) & width_mask                # zero-extended raw bits
    _, _, _, _, (narrow, kmin) = _partition(gdf, col, n, fb)
    assert narrow == (0 if wide else 1)
    if np.dtype(dtype).itemsize == 8 and narrow:
        assert kmin == int(keys.min()) & 0xFFFFFFFFFFFFFFFF
    key64 = raw - np.uint64(kmin)                                           # what the tuples store
    fine = (hash_a(raw) >> np.uint32(32 - fb)).astype(np.int64)             # partition id: a function of the RAW key
    exp_off = np.concatenate([[0], np.cumsum(np.bincount(fine, minlength=1 << fb))])
    for _ in range(reps):
        k, i, off, nj, _ = _partition(gdf, col, n, fb)
        assert nj == n
        np.testing.assert_array_equal(off, exp_off)
        assert np.array_equal(np.sort(i), np.arange(n)), "every row exactly once"
        assert np.array_equal(k.view(np.uint64), key64[i]), "tuple carries its own key"
        assert np.array_equal(fine[i], np.repeat(np.arange(1 << fb), np.diff(exp_off))), "rows sit in their partition"


def test_partition_skips_null_rows(gdf):
    from libgdf_amd.columns import column_from_numpy
    n = 50000
    keys = gen_rand(np.int64, n)
    valid = np.random.randint(0, 2, size=n).astype(bool)
    col = column_from_numpy(keys, valid)
    k, i, off, nj, _ = _partition(gdf, col, n, 4)
    assert nj == valid.sum()
    assert np.array_equal(np.sort(i[:nj]), np.nonzero(valid)[0])
