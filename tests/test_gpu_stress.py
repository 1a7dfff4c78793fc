"""-m gpu: randomised stress of the two hot operators, time-boxed (GDF_STRESS_SECONDS per operator, default 20).
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
SECONDS = float(os.environ.get("GDF_STRESS_SECONDS", "20"))


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
