"""-m gpu: gdf_order_by and the GDF_SORT group-by (sort.hip) through the C ABI vs the oracle.

Cases follow the reference's tests/sqls/sqls_g_tester.cu:114-865 and tests/cpp/sqls_tester.cu:834-895 (their
known-answer vectors are replayed bit for bit, including the row index per group) plus randomized inputs
checked against oracle.group_by_sort / oracle.order_by.  Integer results are bit-exact and so is the output
ORDER (ascending lexicographic); float sums within 1e-6 of the group's sum of magnitudes."""
import json
import os

import numpy as np
import pytest

from oracle import oracle
from util import gen_rand

pytestmark = pytest.mark.gpu
RTOL = 1e-6
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cols(arrs):
    from libgdf_amd.columns import column_from_numpy
    return [column_from_numpy(a) for a in arrs]


def _run(gdf, op, keys, vals, out_dtype=None, **kw):
    from libgdf_amd.columns import GDF_SORT, get_dtype
    od = None if out_dtype is None else get_dtype(out_dtype)
    k, a, i = gdf.api.group_by(op, _cols(keys), _cols([vals])[0], out_dtype=od, method=GDF_SORT, with_indices=True, **kw)
    return [x.cpu().numpy() for x in k], a.cpu().numpy(), i.cpu().numpy()


def _check(gdf, op, keys, vals, out_dtype=None):
    gk, ga, gi = _run(gdf, op, keys, vals, out_dtype)
    ek, ea, ei = oracle.group_by_sort(op, keys, vals, out_dtype)
    assert len(ga) == len(ea)
    for g, e in zip(gk, ek):
        np.testing.assert_array_equal(g, e)                    # same rows in the same (sorted) order
    np.testing.assert_array_equal(gi, ei)
    assert ga.dtype == ea.dtype
    if op in ("sum", "avg") and np.asarray(vals).dtype.kind == "f":
        _, mag, _ = oracle.group_by_sort("sum", keys, np.abs(np.asarray(vals, dtype=np.float64)))
        if op == "avg":
            _, cnt, _ = oracle.group_by_sort("count", keys, vals, np.int64)
            mag = mag / cnt
        err = np.abs(ga.astype(np.float64) - ea.astype(np.float64))
        assert np.all(err <= RTOL * mag + 1e-300), np.max(err / (mag + 1e-300))
    else:
        np.testing.assert_array_equal(ga, ea)


@pytest.fixture(scope="module")
def sqls():
    with open(os.path.join(GOLD, "sqls_known_answers.json")) as f:
        return json.load(f)


def test_reference_known_answers(gdf, sqls):
    g = sqls["group_by"]
    keys = [np.array(g["keys"][c]["values"], dtype=g["keys"][c]["dtype"]) for c in ("c0", "c1", "c2")]
    for case in g["cases"]:
        vals = np.array(case["agg"]["values"], dtype=case["agg"]["dtype"])
        gk, ga, gi = _run(gdf, case["op"], keys, vals, case["out_dtype"])
        for c, name in zip(gk, ("c0", "c1", "c2")):
            assert list(c) == g["expected_keys"][name], case["ref"]
        assert list(ga) == case["expected"], case["ref"]
        assert list(gi) == g["expected_first_rows"], case["ref"]


def test_order_by_known_answer(gdf, sqls):
    ob = sqls["order_by"]
    cols = [np.array(ob["cols"]["c0"], dtype=np.int32), np.array(ob["cols"]["c1"], dtype=np.int32),
            np.array(ob["cols"]["c2"], dtype=np.float64)]
    assert gdf.api.order_by(_cols(cols)).cpu().tolist() == ob["expected_permutation"]


OPS = ["sum", "min", "max", "count", "avg"]
AGG_DTYPES = [np.int8, np.int16, np.int32, np.int64, np.float32, np.float64]


@pytest.mark.parametrize("op", OPS)
@pytest.mark.parametrize("agg_dtype", AGG_DTYPES, ids=lambda d: np.dtype(d).name)
def test_single_int32_key(gdf, op, agg_dtype):
    n = 30000
    keys = [gen_rand(np.int32, n, 0, 200)]
    vals = gen_rand(agg_dtype, n, -100, 100)
    _check(gdf, op, keys, vals, np.int64 if op == "count" else None)


@pytest.mark.parametrize("op", OPS)
@pytest.mark.parametrize("key_dtypes", [[np.int64], [np.int32, np.int32], [np.int64, np.int32], [np.int8, np.int16, np.int32],
                                        [np.float64], [np.int32, np.float32], [np.int64, np.int64, np.int64],
                                        [np.float32, np.int8, np.float64, np.int16]],
                         ids=lambda d: "-".join(np.dtype(x).name for x in d))
def test_key_shapes(gdf, op, key_dtypes):
    n = 20000
    keys = [gen_rand(dt, n, -6, 6) if np.dtype(dt).kind == "i" else np.round(gen_rand(dt, n) * 6).astype(dt) for dt in key_dtypes]
    vals = gen_rand(np.float64 if op != "count" else np.int32, n)
    _check(gdf, op, keys, vals, np.int32 if op == "count" else None)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 100000])
def test_sizes_and_tile_edges(gdf, n):
    keys = [gen_rand(np.int64, n, -50, 50)]
    _check(gdf, "sum", keys, gen_rand(np.int64, n))
    _check(gdf, "min", keys, gen_rand(np.int32, n))


@pytest.mark.parametrize("shape", ["all_same", "all_different", "wave_same", "block_same", "two_big_groups"])
def test_group_shapes(gdf, shape):
    """tests/groupby/groupby-test.cu:369-445 (AllKeysSame / AllKeysDifferent / WarpKeysSame / BlockKeysSame)."""
    n = 50000
    i = np.arange(n)
    k = {"all_same": np.zeros(n), "all_different": np.random.permutation(n), "wave_same": i // 64, "block_same": i // 256,
         "two_big_groups": (np.random.random(n) < 0.5)}[shape].astype(np.int32)
    np.random.shuffle(k)
    for op in OPS:
        _check(gdf, op, [k], gen_rand(np.float64, n), np.int32 if op == "count" else None)
        _check(gdf, op, [k], gen_rand(np.int32, n), np.int32 if op == "count" else None)


def test_full_range_keys_all_eight_digits(gdf):
    n = 200000
    k = np.random.randint(np.iinfo(np.int64).min, np.iinfo(np.int64).max, size=n, dtype=np.int64)
    k[::7] = k[3]                                              # some duplicates
    _check(gdf, "sum", [k], gen_rand(np.int64, n))
    perm = gdf.api.order_by(_cols([k])).cpu().numpy()
    np.testing.assert_array_equal(perm, oracle.order_by([k]))


@pytest.mark.parametrize("dtypes", [[np.int8], [np.int16], [np.int32], [np.int64], [np.float32], [np.float64],
                                    [np.int32, np.float64], [np.int8, np.int8, np.int64, np.float32]],
                         ids=lambda d: "-".join(np.dtype(x).name for x in d))
def test_order_by_random(gdf, dtypes):
    n = 70001
    cols = [gen_rand(dt, n, -40, 40) if np.dtype(dt).kind == "i" else np.round(gen_rand(dt, n) * 20).astype(dt) / 4 for dt in dtypes]
    perm = gdf.api.order_by(_cols(cols)).cpu().numpy()
    np.testing.assert_array_equal(perm, oracle.order_by(cols))   # stable on both sides: identical, not just equivalent


def test_order_by_float_specials(gdf):
    c = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.5, -1.5, -0.0, 0.0, np.nan, 3e300, -3e300, 5e-324, -5e-324])
    for dt in (np.float64, np.float32):
        with np.errstate(over="ignore"):
            col = c.astype(dt)
        perm = gdf.api.order_by(_cols([col])).cpu().numpy()
        np.testing.assert_array_equal(perm, oracle.order_by([col]))   # -0.0 == +0.0 keep input order; NaN last


def test_nan_keys_are_singleton_groups(gdf):
    k = np.array([1.0, np.nan, 1.0, np.nan, -0.0, 0.0], dtype=np.float64)
    v = np.arange(6, dtype=np.int32)
    gk, ga, gi = _run(gdf, "sum", [k], v)
    assert len(ga) == 4 and list(ga) == [4 + 5, 0 + 2, 1, 3] and list(gi) == [5, 2, 1, 3]
    _check(gdf, "count", [k], v, np.int32)


def test_count_distinct(gdf):
    n = 10000
    keys = [gen_rand(np.int32, n, 0, 37), gen_rand(np.int16, n, 0, 3)]
    vals = gen_rand(np.int32, n)
    gk, ga, gi = _run(gdf, "count", keys, vals, np.int64, distinct=True)
    ek, ea, ei = oracle.group_by_sort("count", keys, vals, np.int64, distinct=True)
    assert len(ga) == 1 and ga[0] == ea[0] == len(set(zip(*[k.tolist() for k in keys])))
    assert len(gi) == 1 and gi[0] == ei[0]
    for g, e in zip(gk, ek):
        np.testing.assert_array_equal(g, e)


def test_count_distinct_is_sort_only(gdf):
    from libgdf_amd import GDFError
    keys, vals = _cols([gen_rand(np.int32, 100)]), _cols([gen_rand(np.int32, 100)])[0]
    with pytest.raises(GDFError, match="GDF_UNSUPPORTED_METHOD"):       # sqls_ops.cu:1347-1349
        gdf.api.group_by("count", keys, vals, distinct=True)


def test_presorted_input_skips_the_sort(gdf):
    n = 5000
    k = np.sort(gen_rand(np.int32, n, 0, 50))
    v = gen_rand(np.int64, n)
    gk, ga, gi = _run(gdf, "sum", [k], v, presorted=True)
    ek, ea, ei = oracle.group_by_sort("sum", [k], v)
    np.testing.assert_array_equal(gk[0], ek[0])
    np.testing.assert_array_equal(ga, ea)
    np.testing.assert_array_equal(gi, ei)


def test_int8_wraps_and_integer_average_truncates(gdf):
    keys = [np.zeros(5, dtype=np.int32)]
    vals = np.array([100, 100, 100, 27, -3], dtype=np.int8)             # 324 -> 68 (mod 256); 68 / 5 = 13
    _, s, _ = _run(gdf, "sum", keys, vals)
    _, a, _ = _run(gdf, "avg", keys, vals)
    assert s.dtype == np.int8 and s[0] == 68 and a.dtype == np.int8 and a[0] == 13
    neg = np.array([-7, -8], dtype=np.int32)                            # -15 / 2 = -7 in C++ (truncation), not -8
    _, a, _ = _run(gdf, "avg", [np.zeros(2, dtype=np.int32)], neg)
    assert a[0] == -7
    _check(gdf, "avg", [gen_rand(np.int32, 3000, 0, 9)], gen_rand(np.int32, 3000, -50, 50))


def test_errors_and_empty(gdf):
    from libgdf_amd import GDFError
    from libgdf_amd.columns import GDF_SORT, column_from_numpy
    k = gen_rand(np.int32, 10)
    masked = column_from_numpy(k, np.ones(10, dtype=bool))
    with pytest.raises(GDFError, match="GDF_VALIDITY_UNSUPPORTED"):
        gdf.api.group_by("sum", [masked], _cols([k])[0], method=GDF_SORT)
    with pytest.raises(GDFError, match="GDF_VALIDITY_UNSUPPORTED"):
        gdf.api.order_by([masked])
    gk, ga, gi = _run(gdf, "sum", [np.zeros(0, dtype=np.int32)], np.zeros(0, dtype=np.int32))
    assert len(ga) == 0 and len(gi) == 0 and len(gk[0]) == 0
    assert gdf.api.order_by(_cols([np.zeros(0, dtype=np.int64)])).numel() == 0


@pytest.mark.parametrize("method", ["sort", "hash"])
def test_mismatched_key_column_sizes_are_an_error_not_a_read_out_of_bounds(gdf, method):
    """the SORT method's direct-path shortcut entered the hash method's code in front of group_by_sort's argument
    checks -- a second key column SHORTER than the first was read out of bounds.  Both methods now answer GDF_COLUMN_SIZE_MISMATCH
    (sort.hip group_by_sort / the reference's gdf_table asserts equal sizes, gdf_table.cuh:249-322), for key columns and for an
    aggregation column of another size; and the SORT method's shortcut leaves the caller's output validity masks untouched."""
    import ctypes as C
    import torch
    from libgdf_amd import GDFError
    from libgdf_amd.columns import GDF_HASH, GDF_SORT, Column, column_array, column_from_numpy, new_context
    m = GDF_SORT if method == "sort" else GDF_HASH
    rs = np.random.RandomState(3)
    n = 50_000
    k0, k1 = rs.randint(0, 50, size=n).astype(np.int64), rs.randint(0, 20, size=n // 2).astype(np.int64)
    v = rs.randint(-9, 9, size=n).astype(np.int64)
    with pytest.raises(GDFError, match="GDF_COLUMN_SIZE_MISMATCH"):
        gdf.api.group_by("sum", [column_from_numpy(k0), column_from_numpy(k1)], column_from_numpy(v), method=m, capacity=n)
    with pytest.raises(GDFError, match="GDF_COLUMN_SIZE_MISMATCH"):
        gdf.api.group_by("sum", [column_from_numpy(k0)], column_from_numpy(v[: n // 3]), method=m, capacity=n)
    if method == "sort":
        # the shortcut must not write validity masks the sort itself never touches: poison them and look afterwards
        kc, vc = column_from_numpy(k0), column_from_numpy(v)
        poison = lambda: torch.full((n // 8 + 64,), 0x5A, dtype=torch.uint8, device="cuda")
        ok = Column(torch.empty(n, dtype=torch.int64, device="cuda"), poison(), 4, size=n)
        oa = Column(torch.empty(n, dtype=torch.int64, device="cuda"), poison(), 4, size=n)
        ctx = new_context(method=GDF_SORT)
        gdf.libgdf.gdf_group_by_sum(1, column_array([kc]), vc.ptr, None, column_array([ok]), oa.ptr, C.byref(ctx))
        assert int(oa.size) == 50
        assert bool((ok.valid == 0x5A).all()) and bool((oa.valid == 0x5A).all())
        ek, ea = oracle.group_by("sum", [k0], v)
        np.testing.assert_array_equal(ok.data[:50].cpu().numpy(), ek[0])
        np.testing.assert_array_equal(oa.data[:50].cpu().numpy(), ea)


def test_large_sort_properties(gdf):
    """1e7 rows: sortedness, permutation and checksum properties (size-independent), plus SORT == HASH aggregates."""
    import torch
    n = 10_000_000
    k = torch.randint(0, 1 << 40, (n,), dtype=torch.int64, device="cuda")
    from libgdf_amd.columns import GDF_SORT, Column
    perm = gdf.api.order_by([Column(k)])
    s = k[perm]
    assert bool((s[1:] >= s[:-1]).all())
    assert int(perm.sum()) == n * (n - 1) // 2 and int(torch.unique(perm).numel()) == n
    kk = k % 100_003
    v = torch.randint(-1000, 1000, (n,), dtype=torch.int64, device="cuda")
    sk, sa = gdf.api.group_by("sum", [Column(kk)], Column(v), method=GDF_SORT)
    hk, ha = gdf.api.group_by("sum", [Column(kk)], Column(v), sort_result=True)
    assert torch.equal(sk[0], hk[0]) and torch.equal(sa, ha) and int(sa.sum()) == int(v.sum())


@pytest.mark.parametrize("op", OPS)
@pytest.mark.parametrize("shape", ["one-int64-column", "two-columns", "negative-keys", "int8-values"])
def test_sort_method_small_key_ranges_take_the_direct_path(gdf, op, shape, force_path):
    """A SORT-method group-by over integer keys with a small value range needs no sort to come out sorted: group_by_single serves
    it from the hash method's direct path (LDS accumulators indexed by the lexicographic group number) plus a last-row pass for
    out_col_indices -- VERDICT r3 item 7.  The contract is the SORT method's (sqls_ops.cu:1134-1289: ascending groups, aggregation in
    the input dtype, COUNT in the output column's, out_col_indices = every group's last row, sqls_g_tester.cu:250-256): against
    oracle.group_by_sort, kernel names through the profile hook, and the same call through the sort (GDF_SORT_NO_DIRECT)."""
    from bench import read_profile
    rs = np.random.RandomState(len(shape) + len(op))
    n = 300_000
    if shape == "two-columns":
        keys = [rs.randint(0, 40, size=n).astype(np.int32), rs.randint(-3, 60, size=n).astype(np.int64)]
    elif shape == "negative-keys":
        keys = [rs.randint(-5000, 3000, size=n).astype(np.int64)]
    else:
        keys = [rs.randint(0, 10_000, size=n).astype(np.int64)]
    vals = rs.randint(-100, 100, size=n).astype(np.int8) if shape == "int8-values" else rs.randint(-1000, 1000, size=n).astype(np.int64)
    lib = gdf._binding._gdf_cdll

    def names_of(call):
        lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
        try:
            call()
        finally:
            lib.gdf_amd_profile_enable(0)
        return set(read_profile(gdf))

    out = np.int64 if op == "count" else None
    names = names_of(lambda: _check(gdf, op, keys, vals, out))
    assert "gb_direct_aggregate" in names and "gb_direct_last_rows" in names and "rs_scatter" not in names, names
    force_path("GDF_SORT_NO_DIRECT")
    names = names_of(lambda: _check(gdf, op, keys, vals, out))
    assert "rs_scatter" in names and "gb_direct_aggregate" not in names, names


def _profiled(gdf, call):
    from bench import read_profile
    lib = gdf._binding._gdf_cdll
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    try:
        out = call()
    finally:
        lib.gdf_amd_profile_enable(0)
    return out, read_profile(gdf)


@pytest.mark.parametrize("shape", ["uniform-62-bit", "duplicates", "int32-then-float64", "two-wide-columns", "three-columns", "buckets-of-2000",
                                   "clustered-top-bits", "negative-and-positive"])
def test_hybrid_sort_equals_the_oracle(gdf, shape, force_path):
    """gdf_order_by over wide keys: the top bits by stable array passes, the rest inside LDS per bucket (sort.hip, hs_local) --
    VERDICT r3 item 7.  Same contract as the all-LSD sort (sqls_ops.cu:1373-1392; stable, so the permutation is IDENTICAL to the
    oracle's, duplicates included), checked on shapes that take the wave-per-bucket route, the workgroup-per-bucket route
    (buckets of ~2000 pairs), a second column group (the row numbers riding along are no longer ascending) and on a shape
    the sample turns away (half of the rows under one value of the top bits: plain LSD, no hs_local launch)."""
    rs = np.random.RandomState(len(shape))
    n = 300_000
    force_path("GDF_HS_MIN_ROWS", 4096)
    expect_hybrid = True
    if shape == "uniform-62-bit":
        cols = [rs.randint(0, 1 << 62, size=n, dtype=np.int64)]
    elif shape == "duplicates":
        pool = rs.randint(0, 1 << 44, size=50_000, dtype=np.int64)
        cols = [pool[rs.randint(0, len(pool), size=n)]]
    elif shape == "int32-then-float64":
        cols = [rs.randint(-1000, 1000, size=n).astype(np.int32), np.round(rs.standard_normal(n) * 1e6) / 64]
        expect_hybrid = None                                     # float images cluster under few exponents: the sample decides
    elif shape == "two-wide-columns":                            # two column groups, the first column with duplicates
        pool = rs.randint(0, 1 << 44, size=50_000, dtype=np.int64)
        cols = [pool[rs.randint(0, len(pool), size=n)], rs.randint(0, 1 << 62, size=n, dtype=np.int64)]
    elif shape == "three-columns":
        cols = [rs.randint(0, 1 << 20, size=n).astype(np.int32), rs.randint(-(1 << 30), 1 << 30, size=n).astype(np.int64),
                rs.randint(-128, 128, size=n).astype(np.int8)]
    elif shape == "buckets-of-2000":
        b = rs.randint(0, 150, size=n).astype(np.int64)
        b[0] = 511                                               # nine bits of bucket number, 150 of the 512 used
        cols = [(b << 53) | rs.randint(0, 1 << 53, size=n, dtype=np.int64)]
    elif shape == "clustered-top-bits":
        k = rs.randint(0, 1 << 30, size=n, dtype=np.int64)
        k[::2] += rs.randint(0, 1 << 61, size=len(k[::2]), dtype=np.int64)
        cols = [k]
        expect_hybrid = False
    else:
        cols = [rs.randint(-(1 << 62), 1 << 62, size=n, dtype=np.int64)]
    perm, prof = _profiled(gdf, lambda: gdf.api.order_by(_cols(cols)).cpu().numpy())
    np.testing.assert_array_equal(perm, oracle.order_by(cols))
    if expect_hybrid is not None:
        assert ("hs_local" in prof) == expect_hybrid, sorted(prof)
    if shape == "two-wide-columns":
        assert prof["hs_local"][1] == 2, prof
    force_path("GDF_SORT_NO_HYBRID")
    perm2, prof2 = _profiled(gdf, lambda: gdf.api.order_by(_cols(cols)).cpu().numpy())
    np.testing.assert_array_equal(perm2, perm)
    assert "hs_local" not in prof2


def test_hybrid_sort_group_by_rides_on_it(gdf, force_path):
    """The SORT-method group-by over keys too wide for the direct route sorts through order_rows as well: same answers as the
    oracle (group order, aggregates, last-row indices) with the hybrid sort underneath."""
    rs = np.random.RandomState(77)
    n = 200_000
    force_path("GDF_HS_MIN_ROWS", 4096)
    pool = rs.randint(-(1 << 60), 1 << 60, size=30_000, dtype=np.int64)
    keys = [pool[rs.randint(0, len(pool), size=n)]]
    vals = rs.randint(-1000, 1000, size=n).astype(np.int64)
    for op in ("sum", "min", "count"):
        _, prof = _profiled(gdf, lambda: _check(gdf, op, keys, vals, np.int64 if op == "count" else None))
        assert "hs_local" in prof, sorted(prof)


def test_hybrid_sort_bucket_the_sample_missed(gdf):
    """A bucket larger than a workgroup's 4096 slots that the strided sample cannot see (its rows sit between the sampled
    windows): hs_local raises its flag and the LSD sort finishes the job from the top-sorted pairs -- same permutation."""
    rs = np.random.RandomState(5)
    n = 1 << 23
    k = rs.randint(1 << 48, 1 << 62, size=n, dtype=np.int64)
    nwin, window = 2048, 1024
    sampled = np.zeros(n, dtype=bool)
    for b in range(nwin):
        begin = b * (n - window) // (nwin - 1)
        sampled[begin:begin + window] = True
    free = np.flatnonzero(~sampled)
    assert len(free) > 100_000
    hide = free[rs.choice(len(free), size=6000, replace=False)]
    k[hide] = rs.randint(0, 1 << 40, size=len(hide), dtype=np.int64)          # all under one value of the top 14 bits
    perm, prof = _profiled(gdf, lambda: gdf.api.order_by(_cols([k])).cpu().numpy())
    assert "hs_local" in prof and prof["rs_scatter"][1] == 2 + 7, prof           # two top passes, then all seven
    np.testing.assert_array_equal(perm, oracle.order_by([k]))
