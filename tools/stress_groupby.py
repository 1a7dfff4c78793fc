#!/usr/bin/env python
"""Randomised parity check of the HASH group-by's PARTITIONED path (hot key window, speculative / exact record layout, XCD-shared or
per-workgroup segments, plain or ballot ranks, pipelined aggregation) against the oracle: random sizes from 2^22 rows up, key distributions
(Zipf, uniform, one dominant key, sorted input, two halves on different keys), 1 - 2 key columns, ops, value dtypes, value / key masks, and
random path switches.  Test infrastructure (it imports the oracle through the tests' own checkers).  Usage: python tools/stress_groupby.py [--seconds S] [--seed N]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-rows", type=int, default=12_000_000)
    a = ap.parse_args()
    import numpy as np
    os.environ.setdefault("LIBGDF_AMD_TESTHOOK", "1")      # path switches (--force ...) go through libgdf_testhook.so: loaded in front of libgdf.so
    import libgdf_amd as gdf
    from test_gpu_groupby import _check, _check_masked, _zipf
    switches = ["GDF_GBP_NO_XCD", "GDF_GBP_NO_SPEC", "GDF_GBP_NO_HOT", "GDF_GBP_PLAIN_RANK"]
    t0, it = time.time(), 0
    while time.time() - t0 < a.seconds:
        rs = np.random.RandomState(a.seed * 1_000_003 + it)
        n = int(rs.randint(1 << 22, a.max_rows))
        shape = ["zipf", "uniform", "one-key", "sorted", "halves"][rs.randint(5)]
        space = int(rs.choice([40_000, 100_000, 600_000]))
        if shape == "uniform":
            k0 = rs.randint(0, space, size=n).astype(np.int64)
        elif shape == "one-key":
            k0 = np.full(n, 77, dtype=np.int64)
            sel = rs.permutation(n)[:n // 50]
            k0[sel] = rs.randint(0, space, size=len(sel))
        elif shape == "halves":
            k0 = np.concatenate([rs.randint(0, space // 3, size=n // 2), rs.randint(space // 3, space, size=n - n // 2)]).astype(np.int64)
        else:
            k0 = _zipf(rs, n, space)
            if shape == "sorted":
                k0 = np.sort(k0)
        k0 += int(rs.choice([0, 10**12, -5000]))
        keys = [k0] + ([rs.randint(0, 16, size=n).astype(np.int32)] if rs.randint(2) else [])
        op = ["sum", "min", "max", "count", "avg"][rs.randint(5)]
        fdt = bool(rs.randint(2))
        vals = rs.random_sample(n) if fdt else rs.randint(-1000, 1000, size=n).astype(np.int64)
        masked = bool(rs.randint(2))
        v_ok = (rs.random_sample(n) > rs.choice([0.5, 0.05])) if masked else None
        k_ok = [(rs.random_sample(n) > 0.02) if (masked and rs.randint(3) == 0) else None] + [None] * (len(keys) - 1)
        out = np.float64 if op == "avg" else None
        forced = {}
        for sw in switches:
            if rs.randint(4) == 0:
                forced[sw] = str(rs.randint(2)) if sw == "GDF_GBP_PLAIN_RANK" else "1"
        gdf.libgdf.gdf_amd_debug_force(b"GDF_GBP_SPEC_MIN_ROWS", b"1")
        for k, v in forced.items():
            gdf.libgdf.gdf_amd_debug_force(k.encode(), v.encode())
        try:
            if masked:
                _check_masked(gdf, op, keys, vals, k_ok, v_ok, out)
            else:
                _check(gdf, op, keys, vals, out)
        except Exception:
            print(f"stress_groupby: FAILED at seed {a.seed} case {it}: n {n} shape {shape} space {space} keys {len(keys)} op {op} float {fdt} masked {masked} forced {forced}", flush=True)
            raise
        finally:
            for k in forced:
                gdf.libgdf.gdf_amd_debug_force(k.encode(), None)
        it += 1
    print(f"stress_groupby: {it} group-bys in {time.time() - t0:.0f} s, all equal to the oracle (seed {a.seed})")


if __name__ == "__main__":
    main()
