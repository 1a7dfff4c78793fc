#!/usr/bin/env python
"""Micro-metrics of SURVEY.md 8d for the non-join operators, through the C ABI, inputs resident in HBM.
Prints one JSON line per operator: rows/s, algorithmic GB/s (bytes of 8d), per-kernel ms (HIP events in libgdf.so).
Usage: python tools/bench_ops.py [--rows N] [--ops groupby,partition,scan,filter,hash]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--groups", type=int, default=10_000)
    ap.add_argument("--ops", default="groupby,partition,scan,filter,hash,sort")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch
    import libgdf_amd as gdf
    from bench import make_probe_keys, read_profile
    from libgdf_amd._binding import rmmOptions_t
    from libgdf_amd.columns import Column
    gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
    lib = gdf._binding._gdf_cdll
    dev = torch.device("cuda", 0)
    n = a.rows

    def timed(name, fn, alg_bytes, extra=None):
        fn()
        lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.reps
        lib.gdf_amd_profile_enable(0)
        prof = read_profile(gdf)
        out = {"op": name, "rows": n, "ms": dt * 1e3, "rows_per_s": n / dt, "algorithmic_GBps": alg_bytes / dt / 1e9,
               "frac_of_8TBps": alg_bytes / dt / 8e12, "kernels_ms": {k: round(v[0] / a.reps, 3) for k, v in prof.items()}}
        if extra:
            out.update(extra)
        print(json.dumps(out), flush=True)

    ops = a.ops.split(",")
    keys = make_probe_keys(n, a.groups, 0x5EED0003, dev)
    vals = make_probe_keys(n, 1000, 0x5EED0004, dev)
    if "groupby" in ops:
        kc, vc = Column(keys), Column(vals)
        for op in ("sum", "avg"):
            timed(f"gdf_group_by_{op} int64 keys, {a.groups} groups, int64 values", lambda: gdf.api.group_by(op, [kc], vc, capacity=1 << 20), 16.0 * n)
    if "sort" in ops:
        from libgdf_amd.columns import GDF_SORT
        kc, vc = Column(keys), Column(vals)
        wide = Column(make_probe_keys(n, 1 << 62, 0x5EED0005, dev))
        timed(f"gdf_order_by int64, {a.groups} distinct values", lambda: gdf.api.order_by([kc]), 16.0 * n,
              {"note": "bytes = keys read + size_t permutation written"})
        timed("gdf_order_by int64, 62-bit keys", lambda: gdf.api.order_by([wide]), 16.0 * n)
        timed(f"gdf_group_by_sum GDF_SORT int64 keys, {a.groups} groups, int64 values",
              lambda: gdf.api.group_by("sum", [kc], vc, capacity=1 << 20, method=GDF_SORT), 16.0 * n)
        spread = Column((keys * 461168601842739) & ((1 << 62) - 1))
        timed(f"gdf_group_by_sum GDF_SORT int64 keys, {a.groups} groups spread over 2^62, int64 values",
              lambda: gdf.api.group_by("sum", [spread], vc, capacity=1 << 20, method=GDF_SORT), 16.0 * n)
    if "hash" in ops:
        kc = Column(keys)
        timed("gdf_hash int64 -> int32", lambda: gdf.api.hash_rows([kc]), 12.0 * n)
    if "partition" in ops:
        kc, vc = Column(keys), Column(vals)
        for p in (8, 64, 256, 1000, 12000):
            timed(f"gdf_hash_partition 2 x int64 columns, P={p}", lambda: gdf.api.hash_partition([kc, vc], [0], p), (16.0 + 16.0 + 8.0) * n,
                  {"note": "bytes = 2 cols read + 2 cols written + key re-read for the histogram"})
    if "scan" in ops:
        vc = Column(vals)
        timed("gdf_prefixsum_i64 inclusive", lambda: gdf.api.prefixsum(vc, True), 16.0 * n)
    if "filter" in ops:
        vc = Column(vals)
        for thr, sel in ((899, "10%"), (499, "50%")):
            def f():
                st = gdf.api.comparison(vc, __import__("numpy").int64(thr), 4)
                gdf.api.apply_stencil(vc, st)
            kept = int((vals > thr).sum().item())
            timed(f"gpu_comparison_static_i64 + gpu_apply_stencil, selectivity {sel}", f, 8.0 * n + 2.0 * n + 8.0 * n + 8.0 * kept,
                  {"kept": kept, "note": "bytes = column read twice (predicate, compaction) + stencil written and read + kept rows written"})


if __name__ == "__main__":
    main()
