"""-m gpu: BASELINE config C5 -- multi-key gdf_group_by_avg, fp64 values with a validity mask, 50 % nulls, Zipf-skewed keys --
and the reference's MaxJoinTest (tests/join/join-tests.cu:732-748).

C5 has no counterpart in the reference (it rejects every mask, sqls_ops.cu:1103-1106); the mask semantics are those of
oracle.group_by_masked, which tests/test_oracle_cpu.py pins against pandas.groupby(dropna=True).
  (i)  up to 4e6 rows against the oracle, PLAIN relative tolerance 1e-6 (BASELINE.json north_star) -- the values are in
       [0, 1), nothing cancels, so no tolerance relative to a sum of magnitudes is needed;
  (ii) the full 1e9-row relation through size-independent properties (tools/bench_c5.py: c5_property_checks)."""
import math
import os
import sys

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu
RTOL = 1e-6          # north_star: "within 1e-6 relative for fp32/fp64 sum/avg"


def _c5_numpy(n, zipf_values, seed):
    rs = np.random.RandomState(seed)
    u = rs.random_sample(n)
    k0 = np.clip(np.exp(u * math.log(zipf_values + 1.0)).astype(np.int64) - 1, 0, zipf_values - 1)   # p(r) ~ 1 / r
    k1 = rs.randint(0, 16, size=n).astype(np.int32)
    v = rs.random_sample(n)
    ok = rs.random_sample(n) < 0.5
    return k0, k1, v, ok


@pytest.mark.parametrize("n,zipf_values", [(200_000, 1_000), (4_000_000, 1_000_000), (3_000_000, 50)],
                         ids=["dense-path", "partitioned-sorted-path", "direct-path-heavy-skew"])
@pytest.mark.parametrize("op", ["avg", "sum", "count"])
def test_c5_shape_against_the_oracle(gdf, n, zipf_values, op):
    from libgdf_amd.columns import column_from_numpy, get_dtype
    k0, k1, v, ok = _c5_numpy(n, zipf_values, 17)
    top = np.bincount(k0).max() / n
    assert top > 0.05                                            # Zipf(s=1): the hottest key holds > 5 % of the rows
    kc = [column_from_numpy(k0), column_from_numpy(k1)]
    vc = column_from_numpy(v, ok)
    out = np.int64 if op == "count" else np.float64
    gk, ga, gok = gdf.api.group_by(op, kc, vc, out_dtype=get_dtype(out), with_masks=True)
    gk, ga, gok = [x.cpu().numpy() for x in gk], ga.cpu().numpy(), gok.numpy()
    ek, ea, eok = oracle.group_by_masked(op, [k0, k1], v, [None, None], ok, out)
    if op != "avg":                                              # AVG comes out sorted (groupby.cuh:345-386); the others need not
        order = np.lexsort((gk[1], gk[0]))
        gk, ga, gok = [k[order] for k in gk], ga[order], gok[order]
    assert len(ga) == len(ea)
    np.testing.assert_array_equal(gk[0], ek[0])
    np.testing.assert_array_equal(gk[1], ek[1])
    np.testing.assert_array_equal(gok, eok)                       # a group is null iff it has no valid value
    assert (ga[~gok] == 0).all()
    if op == "count":
        np.testing.assert_array_equal(ga, ea)
    else:
        np.testing.assert_allclose(ga[gok], ea[eok], rtol=RTOL, atol=0.0)        # plain relative


@pytest.mark.parametrize("order", ["sorted", "shuffled", "second-column-outliers"])
def test_guessed_key_ranges_are_verified(gdf, order):
    """From 2^24 rows on, keys that only pack by range get their ranges GUESSED from a 65536-row prefix and the partitioned
    path's count kernel checks every key against the guess (csrc/groupby.hip gb_plan_range_sampled / gbp_count).  Sorted
    keys make the prefix see a sliver of the range: the call must notice and redo the plan exactly.  Shuffled keys keep
    the guess; outliers in the SECOND column only are for the scatter kernel to notice.  All against the oracle."""
    from libgdf_amd.columns import column_from_numpy
    n = (1 << 24) + 12345
    rs = np.random.RandomState(5)
    k0 = (np.arange(n, dtype=np.int64) // 40) - 1000            # 420 k values, ascending: the prefix spans ~1600 of them
    k1 = rs.randint(-2, 3, size=n).astype(np.int32)
    if order != "sorted":
        perm = rs.permutation(n)
        k0, k1 = k0[perm], k1[perm]
    if order == "second-column-outliers":
        # the count kernel skips a last key column that cannot change the partition id; its values are checked against the
        # guess by the scatter kernel instead (gbp_scatter, flags[1]): values the prefix never showed, far behind it
        k1[n - 5000::7] = 100
        k1[3_000_000] = -77
    v = rs.randint(-1000, 1000, size=n).astype(np.int64)
    gk, ga = gdf.api.group_by("sum", [column_from_numpy(k0), column_from_numpy(k1)], column_from_numpy(v))
    gk, ga = [x.cpu().numpy() for x in gk], ga.cpu().numpy()
    ek, ea = oracle.group_by("sum", [k0, k1], v)
    o = np.lexsort((gk[1], gk[0]))
    np.testing.assert_array_equal(gk[0][o], ek[0])
    np.testing.assert_array_equal(gk[1][o], ek[1])
    np.testing.assert_array_equal(ga[o], ea)


@pytest.mark.parametrize("null_keys", [0.0, 0.01], ids=["no-null-keys", "1pct-null-keys"])
def test_c5_full_size_properties(gdf, null_keys):
    """1e9 rows, 1e6 Zipf values x 16: ~1.6e7 groups, the hottest key pair holds ~0.4 % of the rows.  Second variant (SURVEY
    8d): 1 % NULL KEYS -- a validity mask on key column 0; those rows belong to no group."""
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from bench_c5 import c5_property_checks, make_c5, null_key_mask
    dev = torch.device("cuda", 0)
    n = 1_000_000_000
    k0, k1, v, ok, mask = make_c5(n, dev)
    kok, kmask = null_key_mask(n, dev, null_keys) if null_keys > 0 else (None, None)
    checks, good = c5_property_checks(gdf, k0, k1, v, ok, mask, 20_000_000, kok, kmask)
    assert good, checks
    assert checks["groups"] > 15_000_000
    del k0, k1, v, ok, mask, kok, kmask
    torch.cuda.empty_cache()


@pytest.mark.parametrize("op", ["avg", "count"])
def test_c5_shape_with_null_keys_against_the_oracle(gdf, op):
    """The 1 %-null-keys variant of C5 at a size the oracle finishes: null keys in either key column drop the row."""
    from libgdf_amd.columns import column_from_numpy, get_dtype
    n = 4_000_000
    k0, k1, v, ok = _c5_numpy(n, 1_000_000, 23)
    rs = np.random.RandomState(24)
    kok0, kok1 = rs.random_sample(n) >= 0.01, rs.random_sample(n) >= 0.01
    kc = [column_from_numpy(k0, kok0), column_from_numpy(k1, kok1)]
    out = np.int64 if op == "count" else np.float64
    gk, ga, gok = gdf.api.group_by(op, kc, column_from_numpy(v, ok), out_dtype=get_dtype(out), with_masks=True)
    gk, ga, gok = [x.cpu().numpy() for x in gk], ga.cpu().numpy(), gok.numpy()
    ek, ea, eok = oracle.group_by_masked(op, [k0, k1], v, [kok0, kok1], ok, out)
    order = np.lexsort((gk[1], gk[0]))
    gk, ga, gok = [k[order] for k in gk], ga[order], gok[order]
    np.testing.assert_array_equal(gk[0], ek[0])
    np.testing.assert_array_equal(gk[1], ek[1])
    np.testing.assert_array_equal(gok, eok)
    if op == "count":
        np.testing.assert_array_equal(ga, ea)
    else:
        np.testing.assert_allclose(ga[gok], ea[eok], rtol=RTOL, atol=0.0)


@pytest.mark.parametrize("how", ["inner", "left"])
def test_max_join_size_2_to_the_29(gdf, how):
    """MaxJoinTest.HugeJoinSize (join-tests.cu:732-748): a 100-row left table against 2^29 int32 rows on the right must
    succeed.  LEFT keeps the 2^29-row table on the build side; INNER may flip (joining.h:58-66).  Checked against a
    brute-force count of every left key in the right column."""
    import torch
    from libgdf_amd.columns import Column
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(29)
    nr = 1 << 29
    right = torch.randint(0, 2**31 - 1, (nr,), generator=g, device=dev, dtype=torch.int32)        # rand() % RAND_MAX
    left = torch.randint(0, 2**31 - 1, (100,), generator=g, device=dev, dtype=torch.int32)
    left[:40] = right[torch.randint(0, nr, (40,), generator=g, device=dev)]                       # make sure something matches
    li, ri = gdf.api.join([Column(left)], [Column(right)], how=how)
    li, ri = li.long(), ri.long()
    counts = torch.stack([(right == k).sum() for k in left])
    matched = ri >= 0
    assert bool((left[li[matched]] == right[ri[matched]]).all())
    assert int(matched.sum()) == int(counts.sum())
    assert int(torch.unique(li[matched] * nr + ri[matched]).numel()) == int(matched.sum())        # no pair twice
    got = torch.bincount(li[matched], minlength=100)
    assert torch.equal(got, counts)
    if how == "left":
        assert int((~matched).sum()) == int((counts == 0).sum())                                  # (l, -1) once per unmatched row
        assert torch.equal(torch.sort(li[~matched]).values, torch.nonzero(counts == 0).flatten())
    else:
        assert bool(matched.all())
    del right, li, ri
    torch.cuda.empty_cache()


def test_inner_join_with_a_2_to_the_29_row_build_side(gdf):
    """The same size with the big table forced onto the BUILD side of an inner join (the probe side is larger): 2^29 unique
    build keys, 6e8 probe rows that all hit -- the third partitioning level of csrc/join.hip."""
    import torch
    from bench import make_probe_keys
    from libgdf_amd.columns import Column
    dev = torch.device("cuda", 0)
    nb, npr = 1 << 29, 600_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(31)
    build = torch.randperm(nb, generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    probe = make_probe_keys(npr, nb, 0x5EED0009, dev).to(torch.int32)
    li, ri = gdf.api.join([Column(probe)], [Column(build)])
    assert li.numel() == npr
    seen = torch.zeros(npr, dtype=torch.bool, device=dev)
    step = 1 << 27
    for s in range(0, npr, step):
        l, r = li[s:s + step].long(), ri[s:s + step].long()
        assert bool((probe[l] == build[r]).all())
        seen[l] = True
    assert bool(seen.all())
    del build, probe, li, ri, seen
    torch.cuda.empty_cache()
