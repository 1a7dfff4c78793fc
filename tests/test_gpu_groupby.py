"""-m gpu: gdf_group_by_{sum,min,max,avg,count} (HASH) through the C ABI vs the oracle.

Cases follow tests/groupby/groupby-test.cu:369-445 and test_parameters.cuh:126-153 (1-3 key columns,
all aggregation dtypes; AllKeysSame, AllKeysDifferent, WarpKeysSame, BlockKeysSame, EmptyInput) plus the
reference's known-answer vectors.  Integer aggregates are bit-exact; float sums/averages within 1e-6
relative (the reference's own test allows 1 %, groupby-test.cu:346-364)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import oracle
from util import gen_rand, sort_groups

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def _cols(arrs):
    from libgdf_amd.columns import column_from_numpy
    return [column_from_numpy(a) for a in arrs]


def _run(gdf, op, keys, vals, out_dtype=None, sort_result=False):
    from libgdf_amd.columns import get_dtype
    od = None if out_dtype is None else get_dtype(out_dtype)
    k, a = gdf.api.group_by(op, _cols(keys), _cols([vals])[0], out_dtype=od, sort_result=sort_result)
    return [x.cpu().numpy() for x in k], a.cpu().numpy()


def _check(gdf, op, keys, vals, out_dtype=None):
    gk, ga = _run(gdf, op, keys, vals, out_dtype)
    ek, ea = oracle.group_by(op, keys, vals, out_dtype)
    gk, ga = sort_groups(gk, ga)
    assert len(ga) == len(ea)
    for g, e in zip(gk, ek):
        np.testing.assert_array_equal(g, e)
    if op in ("sum", "avg") and (ea.dtype.kind == "f" or np.asarray(vals).dtype.kind == "f") and np.all(np.asarray(vals) >= 0):
        # nothing cancels: the PLAIN relative tolerance BASELINE.json's north_star states (1e-6), no sum-of-magnitudes scale.
        # float32 values: the oracle (like the reference) adds in float32, one value after the other, and is itself ~2e-6 off
        # the exact sum after a thousand additions -- the yardstick is the sum taken in float64
        if np.asarray(vals).dtype == np.float32:
            _, ea = oracle.group_by(op, keys, np.asarray(vals, dtype=np.float64), np.float64 if op == "avg" else None)
        np.testing.assert_allclose(ga.astype(np.float64), ea.astype(np.float64), rtol=RTOL, atol=0.0)
    elif op in ("sum", "avg") and (ea.dtype.kind == "f" or np.asarray(vals).dtype.kind == "f"):
        # values of both signs (the reference's own generator: U(-1, 1), utils.py:36-52): a group's sum can cancel to ~0, where
        # no summation order -- not the reference's atomics either -- is within 1e-6 of the exact value RELATIVE TO THE RESULT.
        # This case, and only this one, is held to 1e-6 of the group's sum of magnitudes (the standard summation error scale).
        _, mag = oracle.group_by("sum", keys, np.abs(np.asarray(vals, dtype=np.float64)))
        if op == "avg":
            _, cnt = oracle.group_by("count", keys, vals, np.int64)
            mag = mag / cnt
        assert np.all(np.abs(ga.astype(np.float64) - ea.astype(np.float64)) <= RTOL * mag + 1e-300), \
            np.max(np.abs(ga.astype(np.float64) - ea.astype(np.float64)) / (mag + 1e-300))
    elif ea.dtype.kind == "f":
        np.testing.assert_array_equal(ga, ea)          # min / max / count pick or count values: exact
    else:
        np.testing.assert_array_equal(ga, ea)


OPS = ["sum", "min", "max", "count", "avg"]
AGG_DTYPES = [np.int8, np.int16, np.int32, np.int64, np.float32, np.float64]


@pytest.fixture(params=["direct", "dense"])
def small_range_path(request, force_path):
    """Integer keys with a small value range take the direct-index path; GDF_GB_NO_DIRECT=1 keeps the dictionary
    (dense) path covered on the same inputs."""
    if request.param == "dense":
        force_path("GDF_GB_NO_DIRECT")
    return request.param


@pytest.mark.parametrize("op", OPS)
@pytest.mark.parametrize("agg_dtype", AGG_DTYPES, ids=lambda d: np.dtype(d).name)
def test_single_int32_key(gdf, op, agg_dtype, small_range_path):
    n = 30000
    keys = [gen_rand(np.int32, n, 0, 200)]
    vals = gen_rand(agg_dtype, n, -100, 100)
    out = np.int64 if op == "count" else (np.float64 if op == "avg" else None)
    _check(gdf, op, keys, vals, out)


@pytest.mark.parametrize("op", OPS)
@pytest.mark.parametrize("key_dtypes", [[np.int64], [np.int32, np.int32], [np.int64, np.int32], [np.int8, np.int16, np.int32],
                                        [np.float64], [np.int32, np.float32], [np.int64, np.int64, np.int64]],
                         ids=lambda d: "-".join(np.dtype(x).name for x in d))
def test_key_shapes(gdf, op, key_dtypes, small_range_path):
    n = 20000
    keys = [gen_rand(dt, n, 0, 12) if np.dtype(dt).kind == "i" else np.round(gen_rand(dt, n) * 6).astype(dt) for dt in key_dtypes]
    vals = gen_rand(np.float64 if op != "count" else np.int32, n)
    out = np.int32 if op == "count" else None
    _check(gdf, op, keys, vals, out)


@pytest.mark.parametrize("op", OPS)
@pytest.mark.parametrize("groups,per", [(1, 16384), (16384, 1), (1024, 32), (1024, 256)],
                         ids=["AllKeysSame", "AllKeysDifferent", "WarpKeysSame", "BlockKeysSame"])
def test_reference_key_patterns(gdf, op, groups, per):
    keys = [np.repeat(np.arange(groups, dtype=np.int64), per)]
    vals = gen_rand(np.int64, groups * per, -1000, 1000)
    _check(gdf, op, keys, vals, np.int64 if op in ("count", "avg") else None)


@pytest.mark.parametrize("op", ["sum", "avg"])
@pytest.mark.parametrize("val_dtype", [np.float32, np.float64], ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("shape", ["direct", "dictionary", "sorted-partitioned", "table"])
def test_float_sums_plain_relative_tolerance(gdf, op, val_dtype, shape):
    """Values in [0, 1): nothing cancels, so every group-by path is held to the PLAIN 1e-6 relative tolerance of
    BASELINE.json's north_star (the sum-of-magnitudes scale is a relaxation that must be stated per case)."""
    n = 400_000
    if shape == "direct":
        keys = [gen_rand(np.int64, n, 0, 300)]
    elif shape == "dictionary":
        lut = np.random.randint(-2**62, 2**62, size=5000, dtype=np.int64)          # sparse 64-bit keys, LDS-accumulator sized group count
        keys = [lut[np.random.randint(0, 5000, size=n)]]
    elif shape == "sorted-partitioned":
        keys = [gen_rand(np.int64, n, 0, 100_000), gen_rand(np.int32, n, 0, 4)]    # more groups than LDS accumulators hold
    else:
        k = np.round(gen_rand(np.float64, n) * 3000)
        k[::97] = np.nan                                                            # NaN keys: the row-comparing hash table
        keys = [k]
    vals = gen_rand(val_dtype, n, positive_only=True)
    if shape == "table":
        # NaN != NaN: every NaN row is its own group in the reference; compare the non-NaN groups only
        from libgdf_amd.columns import get_dtype
        gk, ga = _run(gdf, op, keys, vals, np.float64 if op == "avg" else None)
        m = ~np.isnan(gk[0])
        ek, ea = oracle.group_by(op, [keys[0][~np.isnan(keys[0])]], vals[~np.isnan(keys[0])], np.float64 if op == "avg" else None)
        o = np.argsort(gk[0][m])
        np.testing.assert_array_equal(gk[0][m][o], ek[0])
        np.testing.assert_allclose(ga[m][o].astype(np.float64), ea.astype(np.float64), rtol=RTOL, atol=0.0)
        assert int((~m).sum()) == int(np.isnan(keys[0]).sum())
        return
    _check(gdf, op, keys, vals, np.float64 if op == "avg" else None)


def test_empty_input_sets_sizes_to_zero(gdf):
    import torch
    from libgdf_amd import Column, libgdf, new_context
    from libgdf_amd.columns import column_array
    k = Column(torch.empty(1, dtype=torch.int32, device="cuda"), None, 3, size=0)
    v = Column(torch.empty(1, dtype=torch.int32, device="cuda"), None, 3, size=0)
    ok = Column(torch.empty(4, dtype=torch.int32, device="cuda"), None, 3, size=4)
    oa = Column(torch.empty(4, dtype=torch.int32, device="cuda"), None, 3, size=4)
    ctx = new_context()
    libgdf.gdf_group_by_sum(1, column_array([k]), v.ptr, None, column_array([ok]), oa.ptr, C.byref(ctx))
    assert ok.size == 0 and oa.size == 0


def test_integer_wraparound_and_avg_typing(gdf):
    keys = [np.zeros(4, dtype=np.int32)]
    vals = np.array([100, 100, 100, 27], dtype=np.int8)
    _check(gdf, "sum", keys, vals)
    _check(gdf, "avg", keys, vals, np.float64)
    _check(gdf, "avg", keys, vals, np.int32)
    big = np.array([2**62, 2**62, 2**62, 5], dtype=np.int64)
    _check(gdf, "sum", keys, big)
    _check(gdf, "count", keys, vals, np.int8)
    _check(gdf, "count", keys, vals, np.float32)


def test_reserved_key_pattern_is_a_normal_key(gdf):
    keys = [np.array([-2**63, 5, -2**63, 5, 0], dtype=np.int64)]
    vals = np.array([1, 2, 3, 4, 5], dtype=np.int64)
    for op in OPS:
        _check(gdf, op, keys, vals, np.int64 if op in ("count", "avg") else None)


def test_sorted_output_and_avg_are_ordered(gdf):
    n = 50000
    keys = [gen_rand(np.int32, n, -50, 50), gen_rand(np.int64, n, -3, 3)]
    vals = gen_rand(np.float64, n)
    gk, ga = _run(gdf, "sum", keys, vals, sort_result=True)
    ek, ea = oracle.group_by("sum", keys, vals)
    for g, e in zip(gk, ek):
        np.testing.assert_array_equal(g, e)               # already in lexicographic order
    np.testing.assert_allclose(ga, ea, rtol=RTOL)
    gk, ga = _run(gdf, "avg", keys, vals, np.float64)    # AVG output is always sorted (groupby.cuh:345-386)
    ek, ea = oracle.group_by("avg", keys, vals, np.float64)
    for g, e in zip(gk, ek):
        np.testing.assert_array_equal(g, e)
    np.testing.assert_allclose(ga, ea, rtol=RTOL)


def test_many_groups_grow_the_table(gdf):
    n = 3_000_000
    keys = [np.random.permutation(n).astype(np.int64)]        # every row its own group: forces table growth
    vals = gen_rand(np.int32, n)
    _check(gdf, "sum", keys, vals)


def test_config_c2_shape_reduced(gdf):
    """BASELINE configs[1] at 10M rows: int64 keys = splitmix64 % 10000, int64 values; integers bit-exact."""
    n = 10_000_000
    i = np.arange(n, dtype=np.uint64)
    keys = [(oracle.splitmix64(i + np.uint64(0x5EED0003)) % np.uint64(10000)).astype(np.int64)]
    vals = (oracle.splitmix64(i + np.uint64(0x5EED0004)) % np.uint64(1000)).astype(np.int64)
    _check(gdf, "sum", keys, vals)


def test_config_c2_full_size_bit_exact(gdf):
    """BASELINE configs[1] at its FULL size: gdf_group_by_sum over 100M int64 keys = splitmix64(seed + i) mod
    10000, int64 values = splitmix64(seed2 + i) mod 1000, HASH method -- every group's key and sum bit-exact against the oracle
    (the CPU restatement of groupby-test.cu:227-259; ~7 s of host time for 1e8 rows).  VERDICT r3 item 8b."""
    n = 100_000_000
    i = np.arange(n, dtype=np.uint64)
    keys = [(oracle.splitmix64(i + np.uint64(0x5EED0003)) % np.uint64(10000)).astype(np.int64)]
    vals = (oracle.splitmix64(i + np.uint64(0x5EED0004)) % np.uint64(1000)).astype(np.int64)
    del i
    _check(gdf, "sum", keys, vals)


def test_config_c1_shape_through_the_hip_path(gdf):
    """BASELINE configs[0] (the plumbing config): 1M int32 keys U[0, 1000), int32 values U[-10000, 10000),
    gdf_group_by_sum -- the exact shape, through the HIP path, against pandas.DataFrame.groupby('k')['v'].sum() with the int32
    wrap-around of the reference (aggregation in the input dtype, aggregation_operations.cuh:30-86) and against the oracle."""
    import pandas as pd
    rs = np.random.RandomState(0xabcdef)
    k = rs.randint(0, 1000, size=1_000_000).astype(np.int32)
    v = rs.randint(-10000, 10000, size=1_000_000).astype(np.int32)
    gk, ga = _run(gdf, "sum", [k], v)
    gk, ga = sort_groups(gk, ga)
    want = pd.DataFrame({"k": k, "v": v.astype(np.int64)}).groupby("k")["v"].sum()
    np.testing.assert_array_equal(gk[0], want.index.to_numpy().astype(np.int32))
    np.testing.assert_array_equal(ga, want.to_numpy().astype(np.int32))          # pandas adds in int64: cast back = wrap-around
    assert ga.dtype == np.int32
    _check(gdf, "sum", [k], v)


def test_known_answer_vectors(gdf):
    with open(os.path.join(os.path.dirname(__file__), "golden", "sqls_known_answers.json")) as f:
        g = json.load(f)["group_by"]
    keys = [np.array(g["keys"][c]["values"], dtype=g["keys"][c]["dtype"]) for c in ("c0", "c1", "c2")]
    for case in g["cases"]:
        vals = np.array(case["agg"]["values"], dtype=case["agg"]["dtype"])
        gk, ga = _run(gdf, case["op"], keys, vals, np.dtype(case["out_dtype"]), sort_result=True)
        for c, name in zip(gk, ("c0", "c1", "c2")):
            assert list(c) == g["expected_keys"][name], case["ref"]
        assert list(ga) == case["expected"], case["ref"]


def test_all_valid_masks_change_nothing(gdf):
    """The reference rejects masks (sqls_ops.cu:1103-1106); here an all-ones mask gives the unmasked answer."""
    from libgdf_amd.columns import column_from_numpy
    k, v = gen_rand(np.int32, 1000, 0, 30), gen_rand(np.int32, 1000)
    gk, ga = gdf.api.group_by("sum", [column_from_numpy(k, np.ones(1000, dtype=bool))], column_from_numpy(v, np.ones(1000, dtype=bool)))
    gk, ga = sort_groups([x.cpu().numpy() for x in gk], ga.cpu().numpy())
    ek, ea = oracle.group_by("sum", [k], v)
    np.testing.assert_array_equal(gk[0], ek[0])
    np.testing.assert_array_equal(ga, ea)


# ---- validity masks (BASELINE config C5; beyond the reference, which rejects every mask) ------------------------
def _check_masked(gdf, op, keys, vals, key_valids, val_valid, out_dtype=None, sort_result=False):
    from libgdf_amd.columns import column_from_numpy, get_dtype
    kc = [column_from_numpy(k, v) for k, v in zip(keys, key_valids)]
    vc = column_from_numpy(vals, val_valid)
    od = None if out_dtype is None else get_dtype(out_dtype)
    gk, ga, gok = gdf.api.group_by(op, kc, vc, out_dtype=od, sort_result=sort_result, with_masks=True)
    gk, ga, gok = [x.cpu().numpy() for x in gk], ga.cpu().numpy(), gok.numpy()
    ek, ea, eok = oracle.group_by_masked(op, keys, vals, key_valids, val_valid, out_dtype)
    if not (sort_result or op == "avg"):
        order = np.lexsort(tuple(reversed(gk)))
        gk, ga, gok = [k[order] for k in gk], ga[order], gok[order]
    assert len(ga) == len(ea)
    for g, e in zip(gk, ek):
        np.testing.assert_array_equal(g, e)
    np.testing.assert_array_equal(gok, eok)
    assert (ga[~gok] == 0).all()
    if op in ("sum", "avg") and np.asarray(vals).dtype.kind == "f":
        np.testing.assert_allclose(ga[gok].astype(np.float64), ea[eok].astype(np.float64), rtol=1e-6, atol=1e-9)
    else:
        np.testing.assert_array_equal(ga[gok], ea[eok])


@pytest.mark.parametrize("op", OPS)
@pytest.mark.parametrize("val_dtype", [np.int32, np.int64, np.float32, np.float64], ids=lambda d: np.dtype(d).name)
def test_masked_values_and_keys(gdf, op, val_dtype):
    n = 50000
    k0 = gen_rand(np.int64, n, 0, 300)
    k1 = gen_rand(np.int32, n, 0, 5)
    vals = gen_rand(val_dtype, n, 0, 100) if np.dtype(val_dtype).kind == "i" else gen_rand(val_dtype, n, positive_only=True)
    k0_ok = np.random.random(n) > 0.01
    v_ok = np.random.random(n) > 0.5
    v_ok[k0 < 6] = False                                         # all-null groups
    out = np.int64 if op == "count" else (np.float64 if op == "avg" else None)
    _check_masked(gdf, op, [k0, k1], vals, [k0_ok, None], v_ok, out)
    _check_masked(gdf, op, [k0, k1], vals, [None, None], v_ok, out, sort_result=True)
    _check_masked(gdf, op, [k0], vals, [k0_ok], None, out)


@pytest.mark.parametrize("op", ["sum", "avg", "min"])
def test_masked_many_groups_general_path(gdf, op):
    """More groups than the LDS-accumulator path holds (16384): the global-table path with masks."""
    n = 400000
    k0 = gen_rand(np.int64, n, 0, 50000)
    k1 = gen_rand(np.int32, n, 0, 3)
    vals = gen_rand(np.float64, n, positive_only=True)
    v_ok = np.random.random(n) > 0.5
    k1_ok = np.random.random(n) > 0.01
    _check_masked(gdf, op, [k0, k1], vals, [None, k1_ok], v_ok, np.float64 if op == "avg" else None)


def test_masked_float_keys_first_row_path(gdf):
    n = 30000
    k = np.round(gen_rand(np.float64, n) * 40)
    vals = gen_rand(np.int64, n)
    _check_masked(gdf, "sum", [k], vals, [np.random.random(n) > 0.1], np.random.random(n) > 0.3)


def test_all_rows_null(gdf):
    n = 1000
    k = gen_rand(np.int32, n, 0, 10)
    v = gen_rand(np.int32, n)
    _check_masked(gdf, "sum", [k], v, [np.zeros(n, dtype=bool)], None)        # every key null: no groups
    _check_masked(gdf, "max", [k], v, [None], np.zeros(n, dtype=bool))          # every value null: groups, all null


@pytest.mark.parametrize("op", OPS)
def test_wide_integer_keys_pack_by_range(gdf, op):
    """(int64, int32, int16) is 14 bytes of key: packed as (value - min) bit fields after a min/max pass."""
    n = 60000
    keys = [gen_rand(np.int64, n, -20, 20) + (1 << 40), gen_rand(np.int32, n, -3, 3), gen_rand(np.int16, n, 100, 104)]
    vals = gen_rand(np.int64, n)
    _check(gdf, op, keys, vals, np.int64 if op == "count" else (np.float64 if op == "avg" else None))
    spread = [np.random.randint(-2**62, 2**62, n, dtype=np.int64), gen_rand(np.int32, n, -3, 3)]   # does not fit 63 bits
    spread[0][::3] = spread[0][0]
    _check(gdf, op, spread, vals, np.int64 if op == "count" else (np.float64 if op == "avg" else None))


def test_sort_method_still_rejects_masks(gdf):
    from libgdf_amd import GDFError
    from libgdf_amd.columns import GDF_SORT, column_from_numpy
    k = gen_rand(np.int32, 10)
    with pytest.raises(GDFError, match="GDF_VALIDITY_UNSUPPORTED"):            # sqls_ops.cu:1103-1106
        gdf.api.group_by("sum", [column_from_numpy(k, np.ones(10, dtype=bool))], column_from_numpy(k), method=GDF_SORT)


@pytest.mark.parametrize("op", OPS)
@pytest.mark.parametrize("path", ["partitioned", "sorted", "hash_table"])
def test_many_groups_both_large_paths(gdf, op, path, force_path):
    """Beyond the LDS-resident paths the packed-key group-by radix sorts (key, value) pairs: only the high key bits
    when the low 13 can index LDS accumulators directly ("partitioned"), the whole key otherwise ("sorted",
    forced here with GDF_GB_NO_PART=1); GDF_GB_NO_SORTED=1 keeps the global hash table.  All three must give the
    oracle's answer, masked or not."""
    if path == "hash_table":
        force_path("GDF_GB_NO_SORTED")
    if path == "sorted":
        force_path("GDF_GB_NO_PART")
    n = 300000
    keys = [gen_rand(np.int64, n, -40000, 40000), gen_rand(np.int16, n, 0, 2)]
    vals = gen_rand(np.float64, n)
    out = np.int64 if op == "count" else (np.float64 if op == "avg" else None)
    _check(gdf, op, keys, vals, out)
    _check(gdf, op, [np.random.randint(-2**62, 2**62, n, dtype=np.int64) // 1000 * 1000], gen_rand(np.int32, n), out)   # 64-bit natural layout
    _check_masked(gdf, op, keys, vals, [np.random.random(n) > 0.02, None], np.random.random(n) > 0.5, out)
    _check_masked(gdf, op, keys, gen_rand(np.int64, n), [None, np.random.random(n) > 0.02], None, out)


@pytest.mark.parametrize("op", OPS)
def test_direct_path_ranges(gdf, op):
    """Direct-index path edge cases: negative minima, several columns in mixed radix, a range just inside and just
    outside the 12288-id limit, ids that never occur, constant columns."""
    n = 100000
    out = np.int64 if op == "count" else (np.float64 if op == "avg" else None)
    _check(gdf, op, [gen_rand(np.int64, n, -6000, 6000)], gen_rand(np.int64, n), out)                  # span 12000: direct
    _check(gdf, op, [gen_rand(np.int64, n, -6200, 6200)], gen_rand(np.int64, n), out)                  # span 12400: not direct
    _check(gdf, op, [gen_rand(np.int8, n, -128, 127), gen_rand(np.int16, n, -20, 20)], gen_rand(np.float64, n), out)
    k = (gen_rand(np.int32, n, 0, 50) * 97).astype(np.int32)                                           # sparse ids
    _check(gdf, op, [k, np.full(n, 7, dtype=np.int64)], gen_rand(np.int32, n), out)
    _check(gdf, op, [np.full(n, -(2 ** 62), dtype=np.int64) + gen_rand(np.int64, n, 0, 9)], gen_rand(np.float32, n), out)


@pytest.mark.parametrize("case", ["inside_widened_window", "exact_range_after_bad_guess", "too_wide_after_bad_guess"])
def test_direct_path_guessed_window(gdf, case):
    """Above 2^20 rows a single key column's id window is guessed from the first 65536 rows and widened to the id
    space; rows outside it make the library repeat with the exact range (or leave the direct path)."""
    n = 1_300_000
    k = gen_rand(np.int64, n, 0, 10)
    if case == "inside_widened_window":
        k[100_000:] = gen_rand(np.int64, n - 100_000, -3000, 3000)
    elif case == "exact_range_after_bad_guess":
        k[100_000:] = gen_rand(np.int64, n - 100_000, 0, 12000)
    else:
        k[100_000:] = gen_rand(np.int64, n - 100_000, 0, 50000)
    for op in ("sum", "avg", "min"):
        _check(gdf, op, [k], gen_rand(np.int64, n), np.float64 if op == "avg" else None)


@pytest.mark.parametrize("op", ["sum", "count", "min", "avg"])
@pytest.mark.parametrize("kdt", [np.float64, np.float32], ids=lambda d: np.dtype(d).name)
def test_float_keys_take_the_packed_paths(gdf, op, kdt, force_path):
    """Float key columns enter the packed-key paths through an order-preserving integer image (csrc/groupby.hip,
    f64_image): -0.0 and +0.0 are one group, +-inf and denormals are ordinary keys; few groups (dictionary path) and
    many groups (sorted path); the row-comparing path (GDF_GB_NO_FLOAT_IMAGE) gives the same groups."""
    rs = np.random.RandomState(3)
    n = 60000
    special = np.array([0.0, -0.0, np.inf, -np.inf, 5e-324 if kdt == np.float64 else 1e-45, -1.5, 1.5, 1e30, -1e30], dtype=kdt)
    few = np.concatenate([special[rs.randint(0, len(special), size=n // 2)], (rs.randint(-50, 50, size=n // 2) * 0.25).astype(kdt)])
    many = (rs.randint(-20000, 20000, size=n) * 0.125).astype(kdt)
    vals = gen_rand(np.int64, n, -1000, 1000)
    out = np.int64 if op == "count" else (np.float64 if op == "avg" else None)
    for keys in (few, many):
        gk, ga = _run(gdf, op, [keys], vals, out)
        _check(gdf, op, [keys], vals, out)
        force_path("GDF_GB_NO_FLOAT_IMAGE")
        rk, ra = _run(gdf, op, [keys], vals, out)
        force_path("GDF_GB_NO_FLOAT_IMAGE", None)
        (gk, ga), (rk, ra) = sort_groups(gk, ga), sort_groups(rk, ra)
        np.testing.assert_array_equal(gk[0], rk[0])
        np.testing.assert_array_equal(ga, ra)
    # (float, int) two-column keys, with a mask on the float column
    k1 = gen_rand(np.int32, n, 0, 6)
    _check(gdf, op, [many[: n // 4] * 0 + few[: n // 4], k1[: n // 4]], vals[: n // 4], out)


def test_float_keys_with_nan_keep_one_group_per_nan_row(gdf):
    """NaN != NaN: every NaN key row is a group of its own (the reference's typed ==).  Such inputs keep the row-comparing
    path; the number of groups is what the oracle says."""
    rs = np.random.RandomState(4)
    n = 20000
    keys = (rs.randint(0, 50, size=n) * 0.5).astype(np.float64)
    keys[rs.choice(n, size=37, replace=False)] = np.nan
    vals = gen_rand(np.int64, n, 0, 100)
    gk, ga = _run(gdf, "sum", [keys], vals)
    ek, ea = oracle.group_by("sum", [keys], vals)
    assert len(ga) == len(ea) == 50 + 37
    assert int(np.isnan(gk[0]).sum()) == 37
    order = np.argsort(gk[0][~np.isnan(gk[0])])
    np.testing.assert_array_equal(gk[0][~np.isnan(gk[0])][order], ek[0][~np.isnan(ek[0])])
    np.testing.assert_array_equal(ga[~np.isnan(gk[0])][order], ea[~np.isnan(ek[0])])
    assert sorted(ga[np.isnan(gk[0])].tolist()) == sorted(ea[np.isnan(ek[0])].tolist())


# ---- LDS dictionary path (few groups under sparse keys, >= 2^22 rows: gb_ld_encode / gb_ld_aggregate) ------------------------
def _sparse_lut(ngroups, seed):
    rng = np.random.default_rng(seed)
    lut = rng.integers(-2**62, 2**62, size=ngroups, dtype=np.int64)
    lut[0] = -2**63                                    # the library's reserved key pattern is a normal key
    return np.unique(lut)


@pytest.mark.parametrize("op", OPS)
@pytest.mark.parametrize("val_dtype", [np.int64, np.float64, np.int32, np.float32, np.int8], ids=lambda d: np.dtype(d).name)
def test_lds_dictionary_sparse_keys(gdf, op, val_dtype):
    """C2's sparse twin, reduced: 10 k int64 keys scattered over 2^63, 5M rows; integers bit-exact, float sums plain 1e-6."""
    n = 5_000_000
    lut = _sparse_lut(10_000, 1)
    rng = np.random.default_rng(2)
    keys = [lut[rng.integers(0, len(lut), size=n)]]
    vals = gen_rand(val_dtype, n, positive_only=True) if np.dtype(val_dtype).kind == "f" else gen_rand(val_dtype, n)
    _check(gdf, op, keys, vals, np.float64 if op == "avg" else (np.int64 if op == "count" else None))


@pytest.mark.parametrize("shape", ["sorted", "late-keys", "rare-keys", "limit", "above-limit", "two-columns"])
def test_lds_dictionary_sampling_and_limits(gdf, shape):
    """The dictionary comes from a strided sample: keys it never saw (a sorted column's short runs, keys that only occur
    late or only a few times) must reach the result through the second numbering round; group counts at and above the
    path's limit (10922) and packed multi-column keys take it or leave it without changing the answer."""
    n = 4_500_000
    rng = np.random.default_rng(3)
    if shape == "sorted":
        k = [np.sort(_sparse_lut(9_000, 4)[rng.integers(0, 8_990, size=n)])]
    elif shape == "late-keys":
        lut = _sparse_lut(6_000, 5)
        g = rng.integers(0, 3_000, size=n)
        g[-1000:] = rng.integers(3_000, len(lut), size=1000)          # half of the keys appear in the last 1000 rows only
        k = [lut[g]]
    elif shape == "rare-keys":
        lut = _sparse_lut(8_000, 6)
        g = rng.integers(0, 4_000, size=n)
        pos = rng.choice(n, size=4_000, replace=False)                # every key of the upper half occurs about once
        g[pos] = rng.integers(4_000, len(lut), size=4_000)
        k = [lut[g]]
    elif shape == "limit":
        lut = _sparse_lut(10_922, 7)[:10_922]
        k = [lut[rng.integers(0, len(lut), size=n)]]
    elif shape == "above-limit":
        lut = _sparse_lut(11_500, 8)
        k = [lut[rng.integers(0, len(lut), size=n)]]
    else:
        a = rng.integers(-2**31, 2**31 - 1, size=120, dtype=np.int64).astype(np.int32)
        b = rng.integers(-2**15, 2**15 - 1, size=70, dtype=np.int64).astype(np.int16)
        k = [a[rng.integers(0, 120, size=n)], b[rng.integers(0, 70, size=n)]]
    vals = gen_rand(np.int64, n)
    for op in ("sum", "count", "max"):
        _check(gdf, op, k, vals, np.int64 if op == "count" else None)
    _check(gdf, "avg", k, gen_rand(np.float64, n, positive_only=True), np.float64)


def test_lds_dictionary_switch_matches_dense_path(gdf, force_path):
    """GDF_GB_NO_LDS_DICT (forced through gdf_amd_debug_force) falls back to the L2 dictionary: identical integer results."""
    n = 4_200_000
    lut = _sparse_lut(3_000, 9)
    rng = np.random.default_rng(10)
    keys = [lut[rng.integers(0, len(lut), size=n)]]
    vals = gen_rand(np.int64, n)
    a = sort_groups(*_run(gdf, "sum", keys, vals))
    force_path("GDF_GB_NO_LDS_DICT")
    b = sort_groups(*_run(gdf, "sum", keys, vals))
    np.testing.assert_array_equal(a[0][0], b[0][0])
    np.testing.assert_array_equal(a[1], b[1])


# ---- fused partition pass: the statically typed kernels (csrc/groupby.hip gbp_count<K0, K1> / gbp_scatter<..., K0, K1, VMASK>) ----
@pytest.mark.parametrize("key_dtypes", [(np.int32,), (np.int64,), (np.int32, np.int32), (np.int32, np.int64), (np.int64, np.int32),
                                        (np.int64, np.int64)], ids=lambda ks: "+".join(np.dtype(k).name for k in ks))
@pytest.mark.parametrize("op,val_dtype", [("sum", np.int64), ("avg", np.float64), ("min", np.float64), ("count", np.int64)],
                         ids=lambda x: x if isinstance(x, str) else np.dtype(x).name)
def test_fused_partition_pass_static_signatures(gdf, key_dtypes, val_dtype, op, force_path):
    """>= 2^20 rows and more groups than one set of LDS accumulators: the fused partition pass, whose count / scatter kernels
    are instantiated per (key kinds, value mask) for one or two 4- / 8-byte integer key columns with an 8-byte value column
    (COUNT and every other shape keep the kernels with the type switches; GDF_GBP_DYNAMIC=1 forces those: same answers).
    Without masks, with a value mask (SUM / MIN: no validity bit in the key; AVG: with it), with null key elements."""
    n = (1 << 20) + 4321
    rs = np.random.RandomState(11)
    first = rs.randint(-150_000, 150_000, size=n).astype(key_dtypes[0]) if len(key_dtypes) == 1 else \
        rs.randint(-30_000, 30_000, size=n).astype(key_dtypes[0])
    keys = [first] + [rs.randint(-3, 4, size=n).astype(dt) for dt in key_dtypes[1:]]
    vals = rs.randint(-1000, 1000, size=n).astype(val_dtype) if np.dtype(val_dtype).kind == "i" else rs.random_sample(n)
    out = np.int64 if op == "count" else (np.float64 if op == "avg" else None)
    v_ok = rs.random_sample(n) > 0.5
    k_ok = rs.random_sample(n) > 0.02
    nokeys = [None] * len(keys)
    _check(gdf, op, keys, vals, out)
    _check_masked(gdf, op, keys, vals, nokeys, v_ok, out)
    _check_masked(gdf, op, keys, vals, [k_ok] + nokeys[1:], v_ok, out)
    force_path("GDF_GBP_DYNAMIC")
    _check_masked(gdf, op, keys, vals, nokeys, v_ok, out)


# ---- hot key window: pre-aggregated in the fused scatter kernel's LDS (csrc/groupby.hip GbHot, gbp_scatter_static<..., HOT>) ----
def _kernels_of(gdf, call):
    """names of the kernels one library call launched (the exported profile hooks of include/gdf/gdf_amd_ext.h)"""
    from bench import read_profile
    lib = gdf._binding._gdf_cdll
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    try:
        call()
    finally:
        lib.gdf_amd_profile_enable(0)
    return set(read_profile(gdf))


def _zipf(rs, n, values):
    u = rs.random_sample(n)
    return np.clip(np.exp(u * np.log(values + 1.0)).astype(np.int64) - 1, 0, values - 1)      # p(rank) ~ 1 / rank, rank = value


@pytest.mark.parametrize("op,val_dtype,masked", [("avg", np.float64, True), ("sum", np.int64, False), ("sum", np.float64, True),
                                                 ("min", np.float64, True), ("max", np.int64, False)],
                         ids=["avg-f64-masked", "sum-i64", "sum-f64-masked", "min-f64-masked", "max-i64"])
@pytest.mark.parametrize("shape", ["zipf", "zipf-sorted-input", "zipf-one-int32-column", "uniform", "one-key", "hot-groups-all-null",
                                   "forced-cold-window", "forced-last-window"])
def test_hot_window_is_aggregated_in_the_scatter_kernel(gdf, shape, op, val_dtype, masked, force_path):
    """Skewed keys (BASELINE config C5: Zipf(1) x 16): the densest aligned window of 4096 group ids of a strided sample is
    aggregated by the fused scatter kernel in LDS -- counted by nobody, staged nowhere, merged into the cells with atomics --
    and only the cold rows travel as records.  Reference semantics of the aggregation itself:
    groupby_kernels.cuh:42-108, groupby.cuh:308-419; masks as in DESIGN.md section 4.  Against the oracle, with the window
    (kernel name checked through the profile hook) and with GDF_GBP_NO_HOT; uniform keys must NOT take a window; all rows on one
    key leave no record at all; input sorted by key (the sample's strided windows still see the skew); a window forced onto cold
    keys / onto the last window overflows the 5120-record stage of every tile and exercises the multi-round flush; hot groups
    whose every value is null come out as null groups."""
    if shape == "hot-groups-all-null" and not masked:
        pytest.skip("needs a value mask")
    if shape in ("zipf-sorted-input", "uniform", "forced-last-window") and (op, masked) in (("sum", True), ("max", False)):
        pytest.skip("the same kernels as this shape's other operators (suite time, VERDICT r4 item 8)")
    rs = np.random.RandomState(len(shape) + len(op))
    n = (1 << 22) + 12345
    if shape == "uniform" or shape == "forced-cold-window":
        k0 = rs.randint(0, 40_000, size=n).astype(np.int64)
    elif shape == "one-key":
        k0 = np.full(n, 77, dtype=np.int64)
    else:
        k0 = _zipf(rs, n, 100_000)
    if shape == "zipf-sorted-input":
        k0 = np.sort(k0)
    if shape == "zipf-one-int32-column":
        keys = [(k0 * 16 + rs.randint(0, 16, size=n)).astype(np.int32)]
    elif shape == "one-key":
        keys = [k0, np.full(n, 3, dtype=np.int32)]
        keys[0][:60_000] = rs.randint(0, 100_000, size=60_000)        # (groups enough for the partitioned path: the others hold <= 16384)
        keys[1][:60_000] = rs.randint(0, 16, size=60_000)
    else:
        keys = [k0, rs.randint(0, 16, size=n).astype(np.int32)]
    vals = rs.randint(-1000, 1000, size=n).astype(val_dtype) if np.dtype(val_dtype).kind == "i" else rs.random_sample(n)
    v_ok = (rs.random_sample(n) > 0.5) if masked else None
    if shape == "hot-groups-all-null":
        v_ok[k0 < 8] = False
    if shape == "forced-cold-window":
        force_path("GDF_GBP_HOT_WINDOW", "5")
    if shape == "forced-last-window":
        force_path("GDF_GBP_HOT_WINDOW", "1000000")
    out = np.float64 if op == "avg" else None
    nokeys = [None] * len(keys)

    def run():
        if masked:
            _check_masked(gdf, op, keys, vals, nokeys, v_ok, out)
        else:
            _check(gdf, op, keys, vals, out)

    names = _kernels_of(gdf, run)
    assert ("gbp_scatter_hot" in names) == (shape != "uniform"), names
    force_path("GDF_GBP_NO_HOT")
    names = _kernels_of(gdf, run)
    assert "gbp_scatter_hot" not in names and "gbp_scatter" in names, names


@pytest.mark.parametrize("op,val_dtype,masked", [("avg", np.float64, True), ("sum", np.int64, False), ("max", np.float64, True)],
                         ids=["avg-f64-masked", "sum-i64", "max-f64-masked"])
@pytest.mark.parametrize("shape", ["zipf", "uniform", "zipf-sorted-input", "second-half-on-other-keys", "one-key", "null-keys"])
def test_speculative_record_layout_needs_no_count_pass(gdf, shape, op, val_dtype, masked, force_path):
    """The fused partition pass on the SPECULATIVE layout (csrc/groupby.hip GbSpec): every scatter workgroup appends to its own
    segment of every partition, sized from a strided sample -- no count pass, no atomics -- and the aggregation masks the slack.
    Against the oracle, kernel names through the profile hook: shapes that fit (Zipf and uniform keys, a single dominant key, null
    keys) run without gbp_count; clustered input (rows sorted by key; the second half of the table on other keys than the first)
    overflows a segment, every workgroup stops at its next tile and the call repeats on the exact layout -- with the same answer.
    Reference semantics of the aggregation: groupby_kernels.cuh:42-108, groupby.cuh:308-419."""
    rs = np.random.RandomState(len(shape) * 7 + len(op))
    n = (1 << 22) + 777
    if shape == "uniform":
        k0 = rs.randint(0, 40_000, size=n).astype(np.int64)
    elif shape == "one-key":
        k0 = np.full(n, 77, dtype=np.int64)
        k0[rs.permutation(n)[:60_000]] = rs.randint(0, 100_000, size=60_000)        # (spread over the table: a block of them is clustered input)
    elif shape == "second-half-on-other-keys":
        k0 = np.concatenate([rs.randint(0, 30_000, size=n // 2), rs.randint(30_000, 90_000, size=n - n // 2)]).astype(np.int64)
    else:
        k0 = _zipf(rs, n, 100_000)
    if shape == "zipf-sorted-input":
        k0 = np.sort(k0)
    keys = [k0, rs.randint(0, 16, size=n).astype(np.int32)]
    vals = rs.randint(-1000, 1000, size=n).astype(val_dtype) if np.dtype(val_dtype).kind == "i" else rs.random_sample(n)
    v_ok = (rs.random_sample(n) > 0.5) if masked else None
    k_ok = [(rs.random_sample(n) > 0.03) if shape == "null-keys" else None, None]
    out = np.float64 if op == "avg" else None
    force_path("GDF_GBP_SPEC_MIN_ROWS", "1")

    def run():
        if masked or shape == "null-keys":
            _check_masked(gdf, op, keys, vals, k_ok, v_ok, out)
        else:
            _check(gdf, op, keys, vals, out)

    names = _kernels_of(gdf, run)
    clustered = shape in ("zipf-sorted-input", "second-half-on-other-keys")
    assert "gbp_sample_hist" in names and ("gbp_count" in names) == clustered, names
    # the ranks inside a (tile, partition) group: plain returning LDS atomics when the sample finds no busy partition among the rows
    # that are ranked, the leader ballots otherwise (gbp_rank_plain / gbp_rank) -- the sample decides by default, both are forced here
    # (one-key: every lane of a wave on ONE counter under the plain atomics)
    # (forced for the AVG variant of every shape; the other two operators share the ranking code and keep the sample's choice)
    for plain in (("1", "0") if op == "avg" else ()):
        force_path("GDF_GBP_PLAIN_RANK", plain)
        run()
    force_path("GDF_GBP_PLAIN_RANK", None)
    # (default: ONE segment per partition and XCD, claimed with L2-local atomics; GDF_GBP_NO_XCD: one segment per partition and workgroup, no atomics)
    force_path("GDF_GBP_NO_XCD")
    names = _kernels_of(gdf, run)
    assert "gbp_sample_hist" in names and ("gbp_count" in names) == clustered, names
    force_path("GDF_GBP_NO_SPEC")
    names = _kernels_of(gdf, run)
    assert "gbp_count" in names, names
