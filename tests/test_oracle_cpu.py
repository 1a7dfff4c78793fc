"""The oracle against independent CPU formulations (pandas / numpy), including BASELINE config C1:
1M-row int32 group-by-sum through the pandas reference path (plumbing, no GPU)."""
import numpy as np
import pandas as pd
import pytest

from oracle import oracle
from util import gen_rand, random_valid, sort_pairs


@pytest.mark.parametrize("how", ["inner", "left", "full"])
@pytest.mark.parametrize("dtype", [np.int32, np.int64, np.float64])
def test_join_matches_pandas(how, dtype):
    l = gen_rand(dtype, 2000, low=0, high=300) if np.dtype(dtype).kind == "i" else np.round(gen_rand(dtype, 2000) * 50)
    r = gen_rand(dtype, 700, low=0, high=300) if np.dtype(dtype).kind == "i" else np.round(gen_rand(dtype, 700) * 50)
    li, ri = oracle.join([l], [r], how)
    ldf = pd.DataFrame({"k": l, "l": np.arange(len(l))})
    rdf = pd.DataFrame({"k": r, "r": np.arange(len(r))})
    m = ldf.merge(rdf, on="k", how={"inner": "inner", "left": "left", "full": "outer"}[how])
    el = m["l"].fillna(-1).astype(np.int64).to_numpy()
    er = m["r"].fillna(-1).astype(np.int64).to_numpy()
    a, b = sort_pairs(li, ri)
    c, d = sort_pairs(el, er)
    np.testing.assert_array_equal(a, c)
    np.testing.assert_array_equal(b, d)


def test_join_nulls_never_match():
    l = np.array([1, 2, 3, 4], dtype=np.int32)
    r = np.array([2, 3, 4, 4], dtype=np.int32)
    lv = [np.array([1, 1, 0, 1], dtype=bool)]
    rv = [np.array([1, 1, 1, 0], dtype=bool)]
    li, ri = oracle.join([l], [r], "inner", lv, rv)
    assert list(zip(li, ri)) == [(1, 0), (3, 2)]
    li, ri = oracle.join([l], [r], "left", lv, rv)
    assert list(zip(li, ri)) == [(0, -1), (1, 0), (2, -1), (3, 2)]
    li, ri = oracle.join([l], [r], "full", lv, rv)
    assert list(zip(li, ri)) == [(-1, 1), (-1, 3), (0, -1), (1, 0), (2, -1), (3, 2)]


def test_join_multi_column():
    a0 = gen_rand(np.int32, 500, 0, 8); a1 = gen_rand(np.int64, 500, 0, 8)
    b0 = gen_rand(np.int32, 300, 0, 8); b1 = gen_rand(np.int64, 300, 0, 8)
    li, ri = oracle.join([a0, a1], [b0, b1], "inner")
    m = pd.DataFrame({"x": a0, "y": a1, "l": np.arange(500)}).merge(pd.DataFrame({"x": b0, "y": b1, "r": np.arange(300)}), on=["x", "y"])
    a, b = sort_pairs(li, ri)
    c, d = sort_pairs(m["l"].to_numpy(), m["r"].to_numpy())
    np.testing.assert_array_equal(a, c)
    np.testing.assert_array_equal(b, d)


def test_c1_one_million_row_int32_groupby_sum_vs_pandas():
    """BASELINE.json configs[0]."""
    n = 1_000_000
    k = np.random.randint(0, 1000, size=n).astype(np.int32)
    v = np.random.randint(-10000, 10000, size=n).astype(np.int32)
    keys, agg = oracle.group_by("sum", [k], v)
    exp = pd.DataFrame({"k": k, "v": v.astype(np.int64)}).groupby("k")["v"].sum()
    np.testing.assert_array_equal(keys[0], exp.index.to_numpy().astype(np.int32))
    np.testing.assert_array_equal(agg, exp.to_numpy().astype(np.int32))     # int32 wrap-around semantics


@pytest.mark.parametrize("op", ["sum", "min", "max", "count", "avg"])
def test_groupby_ops_vs_pandas(op):
    n = 20000
    k0 = gen_rand(np.int32, n, 0, 40); k1 = gen_rand(np.int64, n, 0, 5)
    v = gen_rand(np.float64, n)
    out_dtype = np.int64 if op == "count" else np.float64
    keys, agg = oracle.group_by(op, [k0, k1], v, out_dtype=out_dtype)
    g = pd.DataFrame({"a": k0, "b": k1, "v": v}).groupby(["a", "b"])["v"]
    exp = {"sum": g.sum, "min": g.min, "max": g.max, "count": g.count, "avg": g.mean}[op]()
    np.testing.assert_array_equal(keys[0], exp.index.get_level_values(0).to_numpy())
    np.testing.assert_array_equal(keys[1], exp.index.get_level_values(1).to_numpy())
    np.testing.assert_allclose(agg, exp.to_numpy(), rtol=1e-9)


def test_partition_ids_rule():
    k = gen_rand(np.int64, 5000)
    h = oracle.hash_rows([k])
    for p in (1, 5, 8, 10, 257):
        pid = oracle.partition_ids([k], p)
        exp = h & np.uint32(p - 1) if p & (p - 1) == 0 else h % np.uint32(p)
        np.testing.assert_array_equal(pid, exp)


def test_comparison_uses_c_promotions():
    l = np.array([2**24 + 1, 5], dtype=np.int64)
    r = np.array([2**24, 5], dtype=np.float32)
    # int64 vs float32 compares in float32: 2^24+1 rounds to 2^24 -> "equal"
    assert list(oracle.comparison(l, r, 0)) == [1, 1]
    assert list(oracle.comparison(np.array([1, 2, 3], dtype=np.int8), np.int32(2), 2)) == [1, 0, 0]     # correct '<'


@pytest.mark.parametrize("op", ["sum", "min", "max", "avg", "count"])
def test_masked_group_by_semantics_match_pandas(op):
    """BASELINE config C5: rows with a null key are dropped, null values are skipped, an
    all-null group is reported as null -- pandas' ``groupby(dropna=True)`` is the stated oracle."""
    import pandas as pd
    n = 4000
    k0 = np.random.randint(0, 40, n).astype(np.int64)
    k1 = np.random.randint(0, 4, n).astype(np.int32)
    v = np.random.random(n)
    k0_ok = np.random.random(n) > 0.05
    v_ok = np.random.random(n) > 0.5
    v_ok[k0 == 7] = False                                      # some groups without a single valid value
    keys, agg, ok = oracle.group_by_masked(op, [k0, k1], v, [k0_ok, None], v_ok, np.float64 if op == "avg" else (np.int64 if op == "count" else None))
    df = pd.DataFrame({"k0": pd.array(np.where(k0_ok, k0, 0), dtype="Int64"), "k1": k1, "v": np.where(v_ok, v, np.nan)})
    df.loc[~k0_ok, "k0"] = pd.NA
    gb = df.groupby(["k0", "k1"], dropna=True)["v"]
    exp = {"sum": gb.sum(min_count=1), "min": gb.min(), "max": gb.max(), "avg": gb.mean(), "count": gb.count()}[op]
    exp = exp.sort_index()
    np.testing.assert_array_equal(keys[0], exp.index.get_level_values(0).to_numpy(dtype=np.int64))
    np.testing.assert_array_equal(keys[1], exp.index.get_level_values(1).to_numpy(dtype=np.int32))
    if op == "count":
        np.testing.assert_array_equal(agg, exp.to_numpy())
        assert ok.all()
    else:
        np.testing.assert_array_equal(ok, ~np.isnan(exp.to_numpy()))
        np.testing.assert_allclose(agg[ok], exp.to_numpy()[ok], rtol=1e-12)
        assert (agg[~ok] == 0).all() and (~ok).sum() >= 4


@pytest.mark.parametrize("shape", ["fk-pk", "duplicates-and-misses", "empty-build", "empty-probe", "one-key", "negative-keys"])
def test_all_cores_cpu_baseline_join_equals_the_oracle(shape):
    """bench.py's cpu_baseline_all_cores leg (optional second baseline) times orc_join_parallel_i64, an OpenMP
    radix-partitioned hash join in oracle/gdf_oracle.c: it must produce exactly the pairs of orc_join (the restatement of the
    reference's in-test CPU solution, join-tests.cu:260-356) -- multimap semantics, any thread count."""
    rs = np.random.RandomState(len(shape))
    if shape == "fk-pk":
        build, probe = rs.permutation(40_000).astype(np.int64), rs.randint(0, 40_000, size=300_000).astype(np.int64)
    elif shape == "duplicates-and-misses":
        build, probe = rs.randint(0, 5_000, size=20_000).astype(np.int64), rs.randint(0, 7_000, size=50_000).astype(np.int64)
    elif shape == "empty-build":
        build, probe = np.zeros(0, dtype=np.int64), rs.randint(0, 10, size=100).astype(np.int64)
    elif shape == "empty-probe":
        build, probe = rs.randint(0, 10, size=100).astype(np.int64), np.zeros(0, dtype=np.int64)
    elif shape == "one-key":
        build, probe = np.full(300, 7, dtype=np.int64), np.full(500, 7, dtype=np.int64)
    else:
        build, probe = (rs.permutation(30_000) - 15_000).astype(np.int64) * (1 << 33), (rs.randint(-20_000, 20_000, size=90_000)).astype(np.int64) * (1 << 33)
    el, er = oracle.join([probe], [build], "inner")
    for threads in (1, 3, 0):
        l, r, used = oracle.join_parallel_i64(probe, build, threads)
        assert used >= 1
        o = np.lexsort((r, l))
        np.testing.assert_array_equal(l[o], el)
        np.testing.assert_array_equal(r[o], er)
