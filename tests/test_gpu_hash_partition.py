"""-m gpu: gdf_hash / gdf_hash_partition / gdf_prefixsum_* through the C ABI vs the oracle."""
import ctypes as C

import os

import numpy as np
import pytest

from oracle import oracle
from util import ALL_DTYPES, gen_rand, random_valid

pytestmark = pytest.mark.gpu


def _col(gdf, arr, valid=None):
    from libgdf_amd.columns import column_from_numpy
    return column_from_numpy(arr, valid)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("n", [1, 63, 1000, 100003])
def test_hash_single_column_bit_exact(gdf, dtype, n):
    a = gen_rand(dtype, n)
    got = gdf.api.hash_rows([_col(gdf, a)]).cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(got, oracle.hash_rows([a]))


def test_hash_multi_column_and_identity(gdf):
    n = 50000
    cols = [gen_rand(np.int32, n), gen_rand(np.int64, n), gen_rand(np.float64, n), gen_rand(np.int8, n), gen_rand(np.float32, n)]
    dev = [_col(gdf, c) for c in cols]
    got = gdf.api.hash_rows(dev).cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(got, oracle.hash_rows(cols))
    ints = [cols[0], cols[1], cols[3]]
    got = gdf.api.hash_rows([dev[0], dev[1], dev[3]], hash_func=1).cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(got, oracle.hash_rows(ints, identity=True))


def test_hash_equal_rows_equal_hashes(gdf):
    """reference tests/hashing/hash-test.cu:35-158 / python/tests/test_hashing.py:61-85."""
    cols = [gen_rand(dt, 8) for dt in ALL_DTYPES]
    for c in cols:
        c[4] = c[0]
    h = gdf.api.hash_rows([_col(gdf, c) for c in cols]).cpu().numpy()
    assert h[0] == h[4]


def test_hash_errors(gdf):
    import torch
    from libgdf_amd import Column, libgdf, GDFError
    from libgdf_amd.columns import column_array
    a = _col(gdf, gen_rand(np.int32, 10))
    out64 = Column(torch.empty(10, dtype=torch.int64, device="cuda"))
    with pytest.raises(GDFError, match="GDF_UNSUPPORTED_DTYPE"):
        libgdf.gdf_hash(1, column_array([a]), 0, out64.ptr)
    out32 = Column(torch.empty(10, dtype=torch.int32, device="cuda"))
    with pytest.raises(GDFError, match="GDF_INVALID_HASH_FUNCTION"):
        libgdf.gdf_hash(1, column_array([a]), 7, out32.ptr)


@pytest.mark.parametrize("nparts", [1, 5, 8, 10, 257, 1000, 1024, 1025])          # tests/hashing/hash-partition-test.cu:279-289; 257..1024: the 12288-row tile kernel
@pytest.mark.parametrize("n", [100, 100000, 1000000])
def test_hash_partition_membership_and_offsets(gdf, nparts, n):
    k0 = gen_rand(np.int64, n)
    k1 = gen_rand(np.int32, n)
    payload = gen_rand(np.float64, n)
    cols = [k0, payload, k1]
    outs, offsets = gdf.api.hash_partition([_col(gdf, c) for c in cols], [0, 2], nparts)
    perm, exp_off, pid = oracle.hash_partition(cols, [0, 2], nparts)
    assert offsets == [int(x) for x in exp_off]
    got = [o.to_numpy() for o in outs]
    # every partition holds exactly the reference's rows (as a multiset), and rows stay intact
    bounds = list(exp_off) + [n]
    got_pid = oracle.partition_ids([got[0], got[2]], nparts)
    for p in range(nparts):
        lo, hi = bounds[p], bounds[p + 1]
        assert np.all(got_pid[lo:hi] == p)
    exp_rows = np.stack([c[perm].astype(np.float64) for c in cols], axis=1)
    got_rows = np.stack([g.astype(np.float64) for g in got], axis=1)
    for p in range(nparts):
        lo, hi = bounds[p], bounds[p + 1]
        e = exp_rows[lo:hi]; g = got_rows[lo:hi]
        np.testing.assert_array_equal(e[np.lexsort(e.T[::-1])], g[np.lexsort(g.T[::-1])])


@pytest.mark.parametrize("nparts", [17, 32, 64])
@pytest.mark.parametrize("shape", ["key-value", "value-key", "key-only", "last-chunk-of-one-row", "float64-key"])
def test_hash_partition_pair_kernel(gdf, nparts, shape, force_path):
    """One or two 8-byte columns without masks, 16 < P <= 64, >= 2^16 rows take part_scatter_pairs_kernel (csrc/hashing.hip, round 6):
    both columns staged together, the next tile's words requested before the flush, chunks laid out XCD-major inside a partition.
    Offsets as the oracle's (hashing.cu:434-468 partition rule on the pinned Murmur3 row hash), every partition the reference's rows as a
    multiset with rows intact -- and the same call through the generic tile kernel (GDF_HP_NO_PAIRS).  Sizes: an odd row count, and one
    that leaves a LAST CHUNK OF ONE ROW (the kernel reads row pairs).  Reference: hashing.cu:559-654."""
    n = {"last-chunk-of-one-row": 1024 * 2048 + 1}.get(shape, 1_234_567)
    k = gen_rand(np.float64 if shape == "float64-key" else np.int64, n)
    v = gen_rand(np.int64, n)
    cols, hashed = {"key-value": ([k, v], [0]), "value-key": ([v, k], [1]), "key-only": ([k], [0])}.get(shape, ([k, v], [0]))

    def run():
        outs, offsets = gdf.api.hash_partition([_col(gdf, c) for c in cols], hashed, nparts)
        return [o.to_numpy() for o in outs], offsets
    perm, exp_off, pid = oracle.hash_partition(cols, hashed, nparts)
    bounds = list(exp_off) + [n]
    for kernel in ("pairs", "generic"):
        if kernel == "generic":
            force_path("GDF_HP_NO_PAIRS")
        got, offsets = run()
        assert offsets == [int(x) for x in exp_off]
        got_pid = oracle.partition_ids([got[hashed[0]]], nparts)
        for p in range(nparts):
            lo, hi = bounds[p], bounds[p + 1]
            assert np.all(got_pid[lo:hi] == p)
            e = np.stack([c[perm][lo:hi].view(np.int64) for c in cols], axis=1)
            g = np.stack([c[lo:hi].view(np.int64) for c in got], axis=1)
            np.testing.assert_array_equal(e[np.lexsort(e.T[::-1])], g[np.lexsort(g.T[::-1])])


@pytest.mark.parametrize("nparts", [65, 100, 255, 256])
@pytest.mark.parametrize("shape", ["key-only", "key-value", "three-key-middle", "four-key-last", "float64-key", "last-chunk-of-one-row"])
def test_hash_partition_single_stage_kernel(gdf, nparts, shape, force_path):
    """Up to four 8-byte columns without moved masks, 16 < P <= 256, >= 2^16 rows, where the pair kernel does not apply, take
    part_scatter_cols8_kernel (csrc/hashing.hip, round 6): a 16384-row tile whose destinations are computed once from the key column and
    whose one 8-byte LDS stage is reused column after column.  Offsets as the oracle's (hashing.cu:434-468 partition rule on the pinned
    Murmur3 row hash), every partition the reference's rows as a multiset with rows intact -- and the same call through the generic tile
    kernel (GDF_HP_NO_COLS8).  Sizes: an odd row count, and one that leaves a last chunk of one row.  Reference: hashing.cu:559-654."""
    n = {"last-chunk-of-one-row": 1024 * 2048 + 1}.get(shape, 1_234_567)
    k = gen_rand(np.float64 if shape == "float64-key" else np.int64, n)
    v = [gen_rand(np.int64, n), gen_rand(np.float64, n), gen_rand(np.int64, n)]
    cols, hashed = {"key-only": ([k], [0]), "three-key-middle": ([v[0], k, v[1]], [1]),
                    "four-key-last": ([v[0], v[1], v[2], k], [3])}.get(shape, ([k, v[0]], [0]))

    def run():
        outs, offsets = gdf.api.hash_partition([_col(gdf, c) for c in cols], hashed, nparts)
        return [o.to_numpy() for o in outs], offsets
    perm, exp_off, pid = oracle.hash_partition(cols, hashed, nparts)
    bounds = list(exp_off) + [n]
    for kernel in ("cols8", "generic"):
        if kernel == "generic":
            force_path("GDF_HP_NO_COLS8")
        got, offsets = run()
        assert offsets == [int(x) for x in exp_off]
        got_pid = oracle.partition_ids([got[hashed[0]]], nparts)
        for p in range(nparts):
            lo, hi = bounds[p], bounds[p + 1]
            assert np.all(got_pid[lo:hi] == p)
            e = np.stack([c[perm][lo:hi].view(np.int64) for c in cols], axis=1)
            g = np.stack([c[lo:hi].view(np.int64) for c in got], axis=1)
            np.testing.assert_array_equal(e[np.lexsort(e.T[::-1])], g[np.lexsort(g.T[::-1])])


@pytest.mark.parametrize("nparts", [16, 700])
def test_hash_partition_moves_valid_masks(gdf, nparts):
    n = 200000
    k = gen_rand(np.int32, n)
    v = gen_rand(np.int64, n)
    kv, vv = random_valid(n), random_valid(n)
    outs, offsets = gdf.api.hash_partition([_col(gdf, k, kv), _col(gdf, v, vv)], [0], nparts, with_masks=True)
    gk, gv = outs[0].to_numpy(), outs[1].to_numpy()
    gkv, gvv = outs[0].valid_bits(), outs[1].valid_bits()
    # rows (key, key_valid, value, value_valid) are preserved as a multiset
    exp = np.stack([k, kv, v, vv], axis=1).astype(np.int64)
    got = np.stack([gk, gkv, gv, gvv], axis=1).astype(np.int64)
    np.testing.assert_array_equal(exp[np.lexsort(exp.T[::-1])], got[np.lexsort(got.T[::-1])])
    assert outs[0].c.null_count == n - kv.sum()


@pytest.mark.parametrize("nparts", [1025, 3000, 4096, 12000, 16384])
@pytest.mark.parametrize("keys", ["int64", "int32+int64", "identity-int32"])
def test_hash_partition_two_levels(gdf, nparts, keys, force_path):
    """More than 1024 partitions on >= 2^18 rows: two regroup levels (csrc/hashing.hip hash_partition_two_level: rows grouped
    by super-partition into a temporary table, then every super-partition split) -- one FAST key column, the generic row hash,
    the identity hash; value masks travel; power-of-two and other partition counts.  Offsets and the rows of every partition
    (as a multiset) against the oracle, and against the one-level scatter (GDF_HP_ONE_LEVEL) for the offsets."""
    n = 700_001
    rs = np.random.RandomState(nparts % 1000)
    k0 = rs.randint(-2**40, 2**40, size=n).astype(np.int64) if keys != "identity-int32" else rs.randint(0, 2**31 - 1, size=n).astype(np.int32)
    k1 = rs.randint(0, 1000, size=n).astype(np.int32)
    v = rs.random_sample(n)
    vv = rs.random_sample(n) > 0.3
    hash_cols = [0] if keys != "int32+int64" else [2, 0]
    hf = 1 if keys == "identity-int32" else 0
    cols = [k0, v, k1]
    outs, offsets = gdf.api.hash_partition([_col(gdf, k0), _col(gdf, v, vv), _col(gdf, k1)], hash_cols, nparts, hash_func=hf, with_masks=True)
    perm, exp_off, pid = oracle.hash_partition(cols, hash_cols, nparts, hf) if hf else oracle.hash_partition(cols, hash_cols, nparts)
    assert offsets == [int(x) for x in exp_off]
    got = [o.to_numpy() for o in outs]
    gvv = outs[1].valid_bits()
    got_pid = oracle.partition_ids([got[c] for c in hash_cols], nparts, hf) if hf else oracle.partition_ids([got[c] for c in hash_cols], nparts)
    bounds = np.array(list(exp_off) + [n])
    assert np.array_equal(got_pid, np.repeat(np.arange(nparts), np.diff(bounds)))         # every row sits in its partition's range
    # the rows (with the value's validity) are preserved as a multiset, partition by partition: sort by (partition, row content)
    exp_rows = np.stack([pid[perm].astype(np.float64), k0[perm].astype(np.float64), np.where(vv[perm], v[perm], -1.0), k1[perm].astype(np.float64)], axis=1)
    got_rows = np.stack([got_pid.astype(np.float64), got[0].astype(np.float64), np.where(gvv, got[1], -1.0), got[2].astype(np.float64)], axis=1)
    np.testing.assert_array_equal(exp_rows[np.lexsort(exp_rows.T[::-1])], got_rows[np.lexsort(got_rows.T[::-1])])
    force_path("GDF_HP_ONE_LEVEL")
    _, offsets1 = gdf.api.hash_partition([_col(gdf, k0), _col(gdf, v, vv), _col(gdf, k1)], hash_cols, nparts, hash_func=hf, with_masks=True)
    assert offsets1 == offsets


@pytest.mark.parametrize("nparts", [1025, 12000, 16384])
@pytest.mark.parametrize("piece_rows", [None, 3000, 150])
def test_hash_partition_two_levels_pieces_and_skew(gdf, nparts, piece_rows, force_path):
    """Level B walks a super-partition in PIECES -- the rows it got from a group of consecutive level-A chunks (csrc/hashing.hip
    PartLevel::pieces; ~49152 rows at full size, GDF_HP_PIECE brings the target down so that 2e6 rows make dozens of pieces per
    super-partition, down to one level-A chunk per piece).  A third of the rows sit on ONE key: one super-partition of ~700 000
    rows next to short ones.  Offsets and per-partition row multisets against the oracle."""
    n = 2_000_003
    if piece_rows is not None:
        force_path("GDF_HP_PIECE", str(piece_rows))
    rs = np.random.RandomState(nparts % 977)
    k0 = rs.randint(-2**40, 2**40, size=n).astype(np.int64)
    k0[rs.rand(n) < 0.33] = 123456789
    v = rs.randint(0, 2**31, size=n).astype(np.int32)
    outs, offsets = gdf.api.hash_partition([_col(gdf, k0), _col(gdf, v)], [0], nparts)
    perm, exp_off, pid = oracle.hash_partition([k0, v], [0], nparts)
    assert offsets == [int(x) for x in exp_off]
    g0, g1 = outs[0].to_numpy(), outs[1].to_numpy()
    got_pid = oracle.partition_ids([g0], nparts)
    assert np.array_equal(got_pid, np.repeat(np.arange(nparts), np.diff(np.array(list(exp_off) + [n]))))
    exp_rows = np.stack([pid[perm].astype(np.int64), k0[perm], v[perm].astype(np.int64)], axis=1)
    got_rows = np.stack([got_pid.astype(np.int64), g0, g1.astype(np.int64)], axis=1)
    np.testing.assert_array_equal(exp_rows[np.lexsort(exp_rows.T[::-1])], got_rows[np.lexsort(got_rows.T[::-1])])


def test_hash_partition_errors(gdf):
    from libgdf_amd import GDFError
    a = _col(gdf, gen_rand(np.int32, 100))
    b = _col(gdf, gen_rand(np.int32, 50))
    with pytest.raises(GDFError, match="GDF_COLUMN_SIZE_MISMATCH"):
        gdf.api.hash_partition([a, b], [0], 4)
    with pytest.raises(GDFError, match="GDF_INVALID_HASH_FUNCTION"):
        gdf.api.hash_partition([a], [0], 4, hash_func=9)


@pytest.mark.parametrize("dtype", [np.int8, np.int32, np.int64])
@pytest.mark.parametrize("n", [1, 2, 13, 64, 100, 1000, 2048, 2049, 4095, 4096, 4097, 8192, 1000003])      # python/tests/test_prefixsum.py:16-62 + tile edges
@pytest.mark.parametrize("inclusive", [True, False])
def test_prefixsum(gdf, dtype, n, inclusive):
    a = np.random.randint(-100, 100, size=n).astype(dtype)
    got = gdf.api.prefixsum(_col(gdf, a), inclusive).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.prefixsum(a, inclusive))


@pytest.mark.parametrize("dtype", [np.int8, np.int32, np.int64])
def test_prefixsum_many_tiles_unaligned_and_in_place(gdf, dtype):
    """The single-pass (decoupled look-back) scan over thousands of tiles -- every tile waits for its predecessors, in
    whatever order the hardware runs them -- repeated, so that a lost or torn publication would show; a column that starts
    1 element into a buffer (no 16-byte alignment: the three-pass kernels); and in == out."""
    import torch
    from libgdf_amd import Column, libgdf
    n = 30_000_001
    a = np.random.randint(-100, 100, size=n).astype(dtype)
    exp = np.cumsum(a, dtype=dtype)
    dev = torch.from_numpy(a).cuda()
    for _ in range(5):
        got = gdf.api.prefixsum(Column(dev), True)
        assert torch.equal(got.cpu(), torch.from_numpy(exp))
    got = gdf.api.prefixsum(Column(dev[1:]), False).cpu().numpy()                # unaligned slice, exclusive
    np.testing.assert_array_equal(got, (np.cumsum(a[1:], dtype=dtype) - a[1:]).astype(dtype))
    col = Column(dev)
    libgdf.gdf_prefixsum_generic(col.ptr, col.ptr, 1)                            # in place
    assert torch.equal(dev.cpu(), torch.from_numpy(exp))


@pytest.mark.parametrize("switch,value", [("GDF_SCAN_LOOKBACK", "1"), ("GDF_SCAN_LOOKBACK", "2"), ("GDF_SCAN_LOOKBACK", "3"), ("GDF_SCAN_BLOCKED", "1")])
def test_prefixsum_alternative_kernels(gdf, force_path, switch, value):
    """The decoupled look-back kernel (GDF_SCAN_LOOKBACK = 1, 2 = spine mode, 3 = lockstep rounds; not the default: profiles/r2_c_scan_ablation.md)
    and the element-wise kernels (GDF_SCAN_BLOCKED) against numpy, selected through the test hook gdf_amd_debug_force."""
    import torch
    from libgdf_amd import Column
    force_path(switch, value)
    rs = np.random.RandomState(3)
    for dt in (np.int8, np.int32, np.int64):
        for n in (1, 4095, 4096, 4097, 10_000_019):
            a = rs.randint(-100, 100, size=n).astype(dt)
            for inc in (True, False):
                got = gdf.api.prefixsum(Column(torch.from_numpy(a).cuda()), inc).cpu().numpy()
                exp = np.cumsum(a, dtype=dt)
                exp = exp if inc else (exp - a).astype(dt)
                assert np.array_equal(got, exp), (dt, n, inc)


@pytest.mark.parametrize("case", ["default", "bail-out", "in-place"])
def test_prefixsum_rounds_default_bail_out_and_in_place(gdf, force_path, case):
    """From 2^22 elements on gdf_prefixsum_* is ONE pass in lockstep rounds (csrc/scan.hip scan_lookback<.., ROUNDS>; reference contract
    src/scan.cu:11-76).  Its workgroups wait for one another; when they cannot all be resident the kernel bails out and the three
    launches start over from the input (GDF_SCAN_FORCE_BAIL sets the flag before the launch) -- and an in-place scan, whose input a
    bail-out would have destroyed, never takes the rounds; nor do 1-byte elements (their 4 KB tiles lose to the three launches).  int8 /
    int32 / int64, inclusive and exclusive, sizes around whole tiles; the launches are checked."""
    import torch
    from libgdf_amd import Column, libgdf
    from bench import read_profile
    lib = gdf._binding._gdf_cdll
    if case == "bail-out":
        force_path("GDF_SCAN_FORCE_BAIL")
    rs = np.random.RandomState(11)
    for dt in (np.int8, np.int32, np.int64):
        lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
        for n in (1 << 22, (1 << 22) + 4097, 13_000_001):
            a = rs.randint(-100, 100, size=n).astype(dt)
            for inc in (True, False):
                exp = np.cumsum(a, dtype=dt)
                exp = exp if inc else (exp - a).astype(dt)
                if case == "in-place":
                    col = Column(torch.from_numpy(a).cuda())
                    libgdf.gdf_prefixsum_generic(col.ptr, col.ptr, 1 if inc else 0)
                    got = col.data.cpu().numpy()
                else:
                    got = gdf.api.prefixsum(Column(torch.from_numpy(a).cuda()), inc).cpu().numpy()
                assert np.array_equal(got, exp), (dt, n, inc)
        lib.gdf_amd_profile_enable(0)
        names = {k.split("@")[0] for k in read_profile(gdf)}
        rounds_expected = case != "in-place" and dt != np.int8
        assert ("scan_rounds" in names) == rounds_expected, (dt, names)
        assert ("scan_apply" in names) == (case != "default" or dt == np.int8), (dt, names)


def test_prefixsum_large_wraps_like_numpy(gdf):
    n = 20_000_000
    a = np.random.randint(-2**31, 2**31 - 1, size=n, dtype=np.int64).astype(np.int32)
    got = gdf.api.prefixsum(_col(gdf, a), True).cpu().numpy()
    np.testing.assert_array_equal(got, np.cumsum(a, dtype=np.int32))


def test_narrow_keys_extension(gdf):
    """gdf_amd_narrow_keys (include/gdf/gdf_amd_ext.h): what the multi-GPU layer ships instead of 8-byte keys."""
    import torch
    from libgdf_amd import multigpu
    k = torch.randint(-1000, 5000, (300_001,), dtype=torch.int64, device="cuda") + (1 << 45)
    lo, hi = (1 << 45), (1 << 45) + 3999
    out = multigpu._device_narrow(k, lo, hi)
    exp = torch.where((k >= lo) & (k <= hi), k - lo, torch.full_like(k, -1)).to(torch.int32)
    assert out.dtype == torch.int32 and torch.equal(out, exp)
    from libgdf_amd import GDFError
    with pytest.raises(GDFError, match="GDF_INVALID_API_CALL"):
        multigpu._device_narrow(k, 0, 1 << 40)                       # range too wide for 31 bits


@pytest.mark.parametrize("nparts", [1, 2, 7, 8, 64])
@pytest.mark.parametrize("mode", ["narrow", "int64", "int32"])
@pytest.mark.parametrize("n", [0, 1, 5000, 700_003])
def test_shuffle_partition_extension(gdf, nparts, mode, n):
    """gdf_amd_shuffle_partition == narrow + row numbers + gdf_hash_partition(MURMUR3) of the oracle: same offsets, same
    (key, row) multiset in every partition."""
    from libgdf_amd import Column
    import torch
    base = 1 << 41 if mode != "int32" else 0
    k = (np.random.randint(-300, 4000, size=n) + base).astype(np.int32 if mode == "int32" else np.int64)
    narrow = (base, base + 3500) if mode == "narrow" else None
    row_base = 123456
    pk, pr, off = gdf.api.shuffle_partition(Column(torch.from_numpy(k).cuda()), nparts, row_base=row_base, narrow=narrow)
    kk = np.where((k >= narrow[0]) & (k <= narrow[1]), k - narrow[0], -1).astype(np.int32) if narrow else k
    perm, eoff, _ = oracle.hash_partition([kk], [0], nparts)
    assert off == [int(o) for o in eoff]
    pk, pr = pk.cpu().numpy(), pr.cpu().numpy()
    assert pk.dtype == kk.dtype and pr.dtype == np.int32
    bounds = list(off) + [n]
    for p in range(nparts):
        a, b = bounds[p], bounds[p + 1]
        rows = np.sort(pr[a:b])
        np.testing.assert_array_equal(rows, np.sort(perm[a:b]) + row_base)
    if n:
        np.testing.assert_array_equal(kk[pr - row_base], pk)          # every key travels with ITS row number


@pytest.mark.parametrize("nparts", [1, 2, 3, 8, 64])
@pytest.mark.parametrize("mode", ["narrow", "int64", "int32"])
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 5000, 700_003])
def test_shuffle_partition_stable_extension(gdf, nparts, mode, n):
    """gdf_amd_shuffle_partition_stable: same partition assignment as gdf_hash_partition (oracle), keys of a partition in
    INPUT order, and bitmap p marks exactly the input rows of partition p.  Narrowed: rows outside the range stay home."""
    from libgdf_amd import Column
    import torch
    base = 1 << 41 if mode != "int32" else 0
    k = (np.random.randint(-300, 4000, size=n) + base).astype(np.int32 if mode == "int32" else np.int64)
    narrow = (base, base + 3500) if mode == "narrow" else None
    pk, bm, off = gdf.api.shuffle_partition_stable(Column(torch.from_numpy(k).cuda()), nparts, narrow=narrow)
    kk = np.where((k >= narrow[0]) & (k <= narrow[1]), k - narrow[0], -1).astype(np.int32) if narrow else k
    part = (oracle.hash_rows([kk]).astype(np.uint64) % np.uint64(nparts)).astype(np.int64) if n else np.zeros(0, np.int64)
    if narrow:
        part[kk == -1] = -1                                              # dropped: in no partition, in no bitmap
    kept = int((part >= 0).sum())
    pk = pk.cpu().numpy()
    assert len(pk) == kept
    bits = np.unpackbits(bm.cpu().numpy().view(np.uint8).reshape(nparts, -1), axis=1, bitorder="little")[:, :n].astype(bool)
    bounds = list(off) + [kept]
    for p in range(nparts):
        rows = np.flatnonzero(part == p)
        assert bounds[p + 1] - bounds[p] == len(rows)
        np.testing.assert_array_equal(pk[bounds[p]:bounds[p + 1]], kk[rows])          # input order
        np.testing.assert_array_equal(np.flatnonzero(bits[p]), rows)
    assert bits.sum() == kept


def test_shuffle_partition_errors(gdf):
    import torch
    from libgdf_amd import Column, GDFError
    k = Column(torch.zeros(10, dtype=torch.int64, device="cuda"))
    with pytest.raises(GDFError, match="GDF_INVALID_API_CALL"):
        gdf.api.shuffle_partition(k, 4, narrow=(0, 1 << 40))           # range too wide for 31 bits
    with pytest.raises(GDFError, match="GDF_COLUMN_SIZE_TOO_BIG"):
        gdf.api.shuffle_partition(k, 4, row_base=2**31 - 5)            # row numbers must stay int32
    with pytest.raises(GDFError, match="GDF_UNSUPPORTED_DTYPE"):
        gdf.api.shuffle_partition(Column(torch.zeros(10, dtype=torch.float64, device="cuda")), 4)


@pytest.mark.parametrize("dtypes", [[np.int8], [np.int64], [np.int32, np.float64, np.int16], [np.float32, np.int8, np.int64, np.int64]],
                         ids=lambda d: "-".join(np.dtype(x).name for x in d))
def test_gpu_hash_columns_fnv1a(gdf, dtypes):
    """gpu_hash_columns (src/hashops.cu): FNV-1a 64 per row, bit-exact vs the oracle; output mask = AND of the input masks."""
    import ctypes as C
    import torch
    from libgdf_amd import Column, libgdf
    from libgdf_amd._binding import _gdf_cdll
    from libgdf_amd.columns import column_array, column_from_numpy
    from util import gen_rand, random_valid
    n = 10007
    arrs = [gen_rand(dt, n) for dt in dtypes]
    valids = [random_valid(n) if i % 2 == 0 else None for i in range(len(arrs))]
    cols = [column_from_numpy(a, v) for a, v in zip(arrs, valids)]
    out = Column(torch.empty(n, dtype=torch.int64, device="cuda"), torch.zeros((n + 7) // 8 + 64, dtype=torch.uint8, device="cuda"), 4)
    fn = _gdf_cdll.gpu_hash_columns
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    assert fn(column_array(cols), len(cols), C.addressof(out.c), None) == 0
    np.testing.assert_array_equal(out.to_numpy().view(np.uint64), oracle.fnv1a_rows(arrs))
    exp_valid = np.ones(n, dtype=bool)
    for v in valids:
        if v is not None and not v.all():
            exp_valid &= v
    np.testing.assert_array_equal(out.valid_bits(), exp_valid)
    assert out.c.null_count == n - exp_valid.sum()
