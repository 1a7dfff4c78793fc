#!/usr/bin/env python
"""Randomised property check of gdf_hash_partition at sizes and shapes that reach every scatter kernel (the generic tile kernel, the
pair kernel for <= 2 8-byte columns and <= 64 partitions, the single-stage kernel for <= 4 8-byte columns and <= 256 partitions, the
small-input path, two regroup levels beyond 1024 partitions).  No oracle: the library's own gdf_hash (pinned to the reference's Murmur3
by tests/test_gpu_hash_partition.py) names the partition of every OUTPUT row -- it must be non-decreasing and change exactly at the
returned offsets -- and the output rows are the input rows as a multiset (rows hashed over ALL columns, sorted and compared).
Usage: python tools/stress_hash_partition.py [--seconds S] [--seed N]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=180.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--case", type=int, default=-1)
    ap.add_argument("--max-rows", type=int, default=30_000_000)
    ap.add_argument("--selftest", action="store_true", help="corrupt one output value of every call: the checks must notice (exit code 0 when they do)")
    a = ap.parse_args()
    import torch
    os.environ.setdefault("LIBGDF_AMD_TESTHOOK", "1")
    import libgdf_amd as gdf
    from libgdf_amd.columns import Column
    g = torch.Generator(device="cuda")
    r = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=g, device="cuda"))
    t0 = time.time()
    it = 0 if a.case < 0 else a.case
    while time.time() - t0 < a.seconds:
        g.manual_seed(a.seed * 1_000_003 + it)
        n = r(1, a.max_rows if it % 4 == 0 else 400_000)
        P = [r(1, 17), r(17, 65), r(65, 257), r(257, 1025), r(1025, 13000)][r(0, 5)]
        ncols = r(1, 6)
        all8 = r(0, 2) == 0                      # the 8-byte-only shapes are the pair / single-stage kernels'
        dts = [torch.int64 if (all8 or r(0, 2)) else torch.int32 for _ in range(ncols)]
        if not all8 and r(0, 3) == 0:
            dts[r(0, ncols)] = torch.float64
        cols = []
        for dt in dts:
            if dt == torch.float64:
                cols.append(torch.randint(-1000, 1000, (n,), generator=g, device="cuda").double() * 0.25)
            else:
                span = [50, 100_000, 2**31 - 1][r(0, 3)]
                cols.append(torch.randint(0, span, (n,), generator=g, device="cuda").to(dt))
        nhash = r(1, min(ncols, 3) + 1)
        hashed = torch.randperm(ncols, generator=g, device="cuda")[:nhash].cpu().tolist()
        tag = (it, n, P, [str(d).replace("torch.", "") for d in dts], hashed)
        if os.environ.get("GDF_STRESS_VERBOSE"):
            print("case", tag, flush=True)
        outs, offsets = gdf.api.hash_partition([Column(c) for c in cols], hashed, P)
        torch.cuda.synchronize()
        got = [o.data[:n] for o in outs]
        if a.selftest:
            got[ncols - 1][n // 2] += 1
            try:
                check(gdf, Column, torch, cols, got, hashed, P, offsets, n, tag)
            except AssertionError as e:
                print("selftest: the corruption was noticed:", str(e)[:120])
                return
            raise SystemExit("selftest FAILED: a corrupted output passed the checks")
        check(gdf, Column, torch, cols, got, hashed, P, offsets, n, tag)
        it += 1
        del outs, got, cols
        if a.case >= 0:
            break
    print(f"stress_hash_partition: {it} calls in {time.time() - t0:.0f} s, all properties hold (seed {a.seed})")


def check(gdf, Column, torch, cols, got, hashed, P, offsets, n, tag):
    # the partition of every output row, by the library's own row hash (a uint32 in an int32 column)
    h = gdf.api.hash_rows([Column(got[c].contiguous()) for c in hashed]).long() & 0xFFFFFFFF
    pid = h % P
    assert bool((pid[1:] >= pid[:-1]).all()), (tag, "partition ids of the output rows are not sorted")
    counts = torch.bincount(pid, minlength=P)
    exp_off = (torch.cumsum(counts, 0) - counts).cpu().tolist()
    assert offsets == exp_off, (tag, "offsets")
    # the same rows: a hash over ALL columns (row contents), as sorted multisets
    hin = gdf.api.hash_rows([Column(c) for c in cols]).long() & 0xFFFFFFFF
    hout = gdf.api.hash_rows([Column(c.contiguous()) for c in got]).long() & 0xFFFFFFFF
    # (a second, independent mix so that a pair of swapped values inside two rows does not cancel)
    mix_in = sum((c.double() if c.dtype == torch.float64 else c).to(torch.float64) * (k + 1.5) for k, c in enumerate(cols))
    mix_out = sum((c.double() if c.dtype == torch.float64 else c).to(torch.float64) * (k + 1.5) for k, c in enumerate(got))
    oi, oo = torch.argsort(hin), torch.argsort(hout)
    assert bool((hin[oi] == hout[oo]).all()), (tag, "rows changed")
    assert abs(float(mix_in.sum()) - float(mix_out.sum())) <= 1e-6 * max(1.0, abs(float(mix_in.sum()))), (tag, "row contents changed")


if __name__ == "__main__":
    main()
