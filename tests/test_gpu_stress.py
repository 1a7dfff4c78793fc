"""-m gpu: randomised stress of the two hot operators, time-boxed (GDF_STRESS_SECONDS per operator; the in-suite default is a short
6-second sample per operator so that the whole GPU suite stays inside the driver's limit -- VERDICT r4 item 8 -- and the long runs are
tools/stress_join.py / tools/stress_groupby.py, minutes at a time, recorded under profiles/).
Joins are held to oracle-free properties (tools/stress_join.py); group-bys of random key shapes / group counts / value
dtypes are compared with the oracle: integer aggregates bit-exact, float sums within 1e-6 of the group's sum of magnitudes."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from oracle import oracle
from util import sort_groups

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SECONDS = float(os.environ.get("GDF_STRESS_SECONDS", "6"))


def test_join_properties_over_random_shapes():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_join.py"), "--seconds", str(SECONDS), "--seed", "5"],
                       capture_output=True, text=True, timeout=SECONDS * 10 + 600)
    assert r.returncode == 0 and "all properties hold" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_group_by_random_shapes_against_the_oracle(gdf):
    from libgdf_amd.columns import column_from_numpy
    rng = np.random.default_rng(7)
    key_dtypes = [np.int8, np.int16, np.int32, np.int64, np.float32, np.float64]
    val_dtypes = [np.int8, np.int32, np.int64, np.float32, np.float64]
    ops = ["sum", "min", "max", "count", "avg"]
    t0 = time.time()
    it = 0
    while time.time() - t0 < SECONDS:
        n = int(rng.integers(1, 3_000_000)) if it % 4 else int(rng.integers(4_200_000, 6_000_000))      # (>= 2^22 rows: the LDS dictionary)
        ncols = int(rng.integers(1, 4))
        groups_per_col = [int(rng.integers(1, [40, 3_000, 200_000][int(rng.integers(0, 3))])) for _ in range(ncols)]
        keys = []
        for c in range(ncols):
            dt = key_dtypes[int(rng.integers(0, len(key_dtypes)))]
            if np.dtype(dt).kind == "f":
                pool = np.round(rng.normal(0, 1e6, size=groups_per_col[c])).astype(dt)
            else:
                info = np.iinfo(dt)
                sparse = rng.integers(0, 2) == 0
                lo, hi = (int(info.min), int(info.max)) if sparse else (-50, max(-49, min(int(info.max), groups_per_col[c])))
                pool = rng.integers(lo, hi, size=groups_per_col[c], endpoint=True).astype(dt)
            keys.append(pool[rng.integers(0, len(pool), size=n)])
        vdt = val_dtypes[int(rng.integers(0, len(val_dtypes)))]
        if np.dtype(vdt).kind == "f":
            vals = rng.random(n).astype(vdt)
        else:
            info = np.iinfo(vdt)
            vals = rng.integers(max(info.min, -1000), min(info.max, 1000), size=n, endpoint=True).astype(vdt)
        op = ops[int(rng.integers(0, len(ops)))]
        out_dtype = np.float64 if op == "avg" else (np.int64 if op == "count" else None)
        from libgdf_amd.columns import get_dtype
        gk, ga = gdf.api.group_by(op, [column_from_numpy(k) for k in keys], column_from_numpy(vals),
                                  out_dtype=None if out_dtype is None else get_dtype(out_dtype))
        gk, ga = sort_groups([k.cpu().numpy() for k in gk], ga.cpu().numpy())
        ek, ea = oracle.group_by(op, keys, vals, out_dtype)
        tag = (it, n, [k.dtype.name for k in keys], groups_per_col, np.dtype(vdt).name, op)
        assert len(ga) == len(ea), (tag, len(ga), len(ea))
        for a, b in zip(gk, ek):
            np.testing.assert_array_equal(a, b, err_msg=str(tag))
        if op == "avg" and np.dtype(vdt).kind != "f":
            # integer values: the sum WRAPS in the value dtype before the division (the reference's typing) -- the oracle itself
            np.testing.assert_allclose(ga.astype(np.float64), ea.astype(np.float64), rtol=1e-12, atol=0.0, err_msg=str(tag))
        elif op in ("sum", "avg") and np.dtype(vdt).kind == "f":
            # float32 sums: the oracle adds in float32 in row order and is itself off the exact sum; the yardstick is float64
            _, ex = oracle.group_by(op, keys, vals.astype(np.float64), np.float64 if op == "avg" else None)
            np.testing.assert_allclose(ga.astype(np.float64), ex.astype(np.float64), rtol=2e-6 if vdt == np.float32 else 1e-6, atol=0.0,
                                       err_msg=str(tag))
        else:
            np.testing.assert_array_equal(ga, ea, err_msg=str(tag))
        it += 1
    assert it >= 3


def test_multi_column_masked_joins_against_the_oracle(gdf):
    """1-3 key columns of random dtypes (integers packed by range or hashed, floats with NaN / -0.0), optional validity masks,
    inner / left / full, sizes that reach the partitioned paths: the pair SET equals the oracle's."""
    from libgdf_amd.columns import column_from_numpy
    from util import sort_pairs
    rng = np.random.default_rng(11)
    dtypes = [np.int8, np.int16, np.int32, np.int64, np.float32, np.float64]
    t0 = time.time()
    it = 0
    while time.time() - t0 < SECONDS:
        nl = int(rng.integers(1, 3_000_000 if it % 3 == 0 else 200_000))
        nr = int(rng.integers(1, 400_000 if it % 3 == 0 else 50_000))
        ncols = int(rng.integers(1, 4))
        how = ["inner", "left", "full"][int(rng.integers(0, 3))]
        left, right = [], []
        for c in range(ncols):
            dt = dtypes[int(rng.integers(0, len(dtypes)))]
            card = int(rng.integers(1, [8, 300, max(2, nr)][int(rng.integers(0, 3))] + 1))
            if c == 0:
                # the first column anchors the size of the result: at least nr / 8 distinct values, so a probe row meets
                # ~8 build rows at most and the oracle's pair list stays within a few hundred MB of host memory
                dt = [np.int32, np.int64, np.float64][int(rng.integers(0, 3))]
                card = max(card, nr // 8 + 1)
            if np.dtype(dt).kind == "f":
                pool = np.round(rng.normal(0, 1e3 if c else 1e9, size=card)).astype(dt)
                if card > 3 and rng.integers(0, 3) == 0:
                    pool[0], pool[1] = np.nan, -0.0
            else:
                info = np.iinfo(dt)
                wide = rng.integers(0, 3) == 0
                lo, hi = (int(info.min), int(info.max)) if wide else (max(int(info.min), -100), min(int(info.max), card))
                pool = rng.integers(lo, hi, size=card, endpoint=True).astype(dt)
            left.append(pool[rng.integers(0, card, size=nl)])
            right.append(pool[rng.integers(0, card, size=nr)])
        masked = rng.integers(0, 3) == 0
        lv = [rng.random(nl) > 0.1 if masked and rng.integers(0, 2) else None for _ in range(ncols)] if masked else None
        rv = [rng.random(nr) > 0.1 if masked and rng.integers(0, 2) else None for _ in range(ncols)] if masked else None
        el, er = oracle.join(left, right, how, lv, rv)
        tag = (it, nl, nr, [a.dtype.name for a in left], how, masked, len(el))
        lc = [column_from_numpy(a, None if lv is None else lv[i]) for i, a in enumerate(left)]
        rc = [column_from_numpy(a, None if rv is None else rv[i]) for i, a in enumerate(right)]
        li, ri = gdf.api.join(lc, rc, how=how)
        assert li.numel() == len(el), (tag, li.numel())
        a, b = sort_pairs(li.cpu().numpy(), ri.cpu().numpy())
        c, d = sort_pairs(el, er)
        np.testing.assert_array_equal(a, c, err_msg=str(tag))
        np.testing.assert_array_equal(b, d, err_msg=str(tag))
        it += 1
    assert it >= 3


def test_fused_join_self_feed_random_worlds(gdf):
    """gdf_amd_fj_* as 1..8 ranks would see them, fed back into one GPU: random sizes, repeated keys, probe keys outside the
    build range, skewed probe sides (the plan's fixed-size regions overflow: *overflowed is reported and nothing is joined)."""
    import torch
    from libgdf_amd import api
    from libgdf_amd.columns import Column
    g = torch.Generator(device="cuda").manual_seed(99)
    r = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=g, device="cuda"))
    t0 = time.time()
    it = done = 0
    while time.time() - t0 < SECONDS:
        it += 1
        world = r(1, 9)
        nb = r(1_000, 2_000_000)
        npr = r(1_000, 20_000_000)
        space = max(2, int(nb * [0.5, 1.0, 2.0][r(0, 3)]))
        base = [0, 1 << 40, -5000][r(0, 3)]
        build = torch.randint(0, space, (nb,), generator=g, device="cuda") + base
        probe = torch.randint(-space // 10, space + space // 10, (npr,), generator=g, device="cuda") + base
        skew = r(0, 5) == 0
        if skew:
            probe[torch.randint(0, npr, (npr // 5,), generator=g, device="cuda")] = int(build[0])
        lo, hi = int(build.min()), int(build.max())
        slices = r(1, 5)
        step = (npr + slices - 1) // slices
        lay_b = api.fj_plan(world, nb * world, nb, max(1.0, nb / space))
        lay_p = api.fj_plan(world, nb * world, step, max(1.0, npr / max(1, min(nb, space))))
        if lay_b is None or lay_p is None:
            continue
        bk, brows, bfill, over = api.fj_send(Column(build), lo, hi, lay_b, 0)
        if over:
            continue
        b = api.FjBuild(bk, bfill, lo, lay_b, nb)
        acc = b.accumulate(npr)
        prows, overflowed = [], False
        per_buf = world * lay_p.block
        for i in range(slices):
            a0, a1 = min(npr, i * step), min(npr, (i + 1) * step)
            pk, prow, pfill, over = api.fj_send(Column(probe[a0:a1]), lo, hi, lay_p, a0)
            overflowed = overflowed or over
            if not overflowed:
                acc.add_recv(pk, pfill, lay_p, i * per_buf)
            prows.append(prow)
        if overflowed:
            assert skew, (it, world, nb, npr, space)            # only the skewed shapes may outgrow 6 sigma of room
            del acc
            b.close()
            continue
        try:
            li, ri = acc.finish()
        except gdf.GDFError as e:                                 # a fine partition outgrew the accumulator's room: allowed for skew only
            assert skew and e.errcode == 12, (it, e)
            b.close()
            continue
        li, ri = li.long(), ri.long()
        rows_b = brows.materialize()[ri].long()
        rows_p = torch.empty_like(li)
        which = li // per_buf
        for i in range(slices):
            sel = which == i
            if bool(sel.any()):
                rows_p[sel] = prows[i].materialize()[li[sel] - i * per_buf].long()
        mult = torch.bincount(build - base, minlength=space)
        inside = (probe >= lo) & (probe <= hi)
        expected = int(mult[(probe[inside] - base)].sum())
        tag = (it, world, nb, npr, space, base, skew, slices)
        assert li.numel() == expected, (tag, li.numel(), expected)
        assert bool((probe[rows_p] == build[rows_b]).all()), tag
        assert int(torch.unique(rows_p * nb + rows_b).numel()) == expected, tag
        b.close()
        done += 1
    assert done >= 2


def test_partition_scan_filter_random_sizes(gdf):
    """gdf_hash_partition (random fan-outs up to 3000, 1-3 columns), gdf_prefixsum_* and comparison + stencil compaction at
    random sizes against numpy."""
    import torch
    from libgdf_amd import Column
    from libgdf_amd.columns import column_from_numpy
    rng = np.random.default_rng(5)
    t0 = time.time()
    it = 0
    while time.time() - t0 < SECONDS:
        n = int(rng.integers(1, 5_000_000))
        # prefix sum
        dt = [np.int8, np.int32, np.int64][int(rng.integers(0, 3))]
        a = rng.integers(-100, 100, size=n).astype(dt)
        inc = bool(rng.integers(0, 2))
        got = gdf.api.prefixsum(column_from_numpy(a), inc).cpu().numpy()
        exp = np.cumsum(a, dtype=dt)
        np.testing.assert_array_equal(got, exp if inc else (exp - a).astype(dt), err_msg=str((it, n, dt, inc)))
        # hash partition: every row lands in the partition its hash names, partitions are contiguous and complete
        P = int(rng.integers(1, 3000))
        k = rng.integers(-2**40, 2**40, size=n).astype(np.int64)
        v = np.arange(n, dtype=np.int32)
        outs, offs = gdf.api.hash_partition([column_from_numpy(k), column_from_numpy(v)], [0], P)
        ok, ov = outs[0].data.cpu().numpy(), outs[1].data.cpu().numpy()
        offs = np.asarray(offs, dtype=np.int64)
        assert len(offs) == P and offs[0] == 0 and np.all(np.diff(offs) >= 0) and offs[-1] <= n, (it, n, P)
        np.testing.assert_array_equal(np.sort(ov), v)
        np.testing.assert_array_equal(k[ov], ok)
        part = oracle.partition_ids([k], P)
        bounds = np.append(offs, n)
        got_part = np.repeat(np.arange(P), np.diff(bounds))
        np.testing.assert_array_equal(part[ov], got_part, err_msg=str((it, n, P)))
        it += 1
    assert it >= 2
