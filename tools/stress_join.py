#!/usr/bin/env python
"""Randomised property check of gdf_inner_join / gdf_left_join at sizes that reach every partition layout and probe path
(host-built and device-built units, speculative and exact layouts, lean / general / chained kernels, optimistic, sparse
optimistic + compaction, count + write).  No oracle: the number of pairs equals the sum over probe rows of the key's
multiplicity in the build relation, every pair joins equal keys, no pair occurs twice, and a LEFT join adds exactly the
probe rows without a partner; a third of the cases also materialise 1 - 3 non-key columns per side (result_cols) and check them.  Usage: python tools/stress_join.py [--seconds S] [--seed N]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def join_with_payloads(gdf, torch, how, pk, bk, npay, tag):
    """gdf_{inner,left}_join with result_cols: `npay` non-key columns per side (int64 / float64 / int32 images of the row number) in front of
    the probe key and behind the build key; checks every materialised column against its source column at the returned indices and
    returns the index tensors"""
    import ctypes as C
    from libgdf_amd import gdf_column, libgdf
    from libgdf_amd.columns import Column, column_array, new_context
    make = [lambda n: torch.arange(n, device="cuda", dtype=torch.int64) * 3 + 1,
            lambda n: torch.arange(n, device="cuda", dtype=torch.float64) * 0.5 - 7.0,
            lambda n: (torch.arange(n, device="cuda", dtype=torch.int64) % 1_000_003).to(torch.int32)]
    pp = [make[i](pk.numel()) for i in range(npay)]
    bp = [make[(i + 1) % 3](bk.numel()) for i in range(npay)]
    left, right = [Column(t) for t in pp] + [Column(pk)], [Column(bk)] + [Column(t) for t in bp]
    nres = len(left) + len(right) - 1
    res = [gdf_column() for _ in range(nres)]
    res_arr = (C.POINTER(gdf_column) * nres)(*[C.pointer(x) for x in res])
    li, ri = gdf_column(), gdf_column()
    ctx = new_context()
    fn = {"inner": libgdf.gdf_inner_join, "left": libgdf.gdf_left_join}[how]
    fn(column_array(left), len(left), (C.c_int * 1)(npay), column_array(right), len(right), (C.c_int * 1)(0), 1, nres, res_arr,
       C.byref(li), C.byref(ri), C.byref(ctx))
    n = int(li.size)
    a = gdf.api._take_library_column(li, torch.int32) if n else torch.zeros(0, dtype=torch.int32, device="cuda")
    b = gdf.api._take_library_column(ri, torch.int32) if n else torch.zeros(0, dtype=torch.int32, device="cuda")
    al, bl = a.long(), b.long()
    sources = [(t, al) for t in pp] + [(pk, al)] + [(t, bl) for t in bp]
    for col, (src, idx) in zip(res, sources):
        assert int(col.size) == n, (tag, "result column size")
        if n == 0:
            continue
        data = torch.empty(n, dtype=src.dtype, device="cuda")
        valid = torch.empty((n + 7) // 8, dtype=torch.uint8, device="cuda")
        gdf.api._hipMemcpyDtoD(data.data_ptr(), col.data, n * data.element_size())
        gdf.api._hipMemcpyDtoD(valid.data_ptr(), col.valid, valid.numel())
        libgdf.gdf_column_free(C.byref(col))
        have = idx >= 0
        bits = ((valid[torch.arange(n, device="cuda") >> 3] >> (torch.arange(n, device="cuda") & 7).to(torch.uint8)) & 1).bool()
        assert bool((bits == have).all()), (tag, "valid bits of a result column")
        assert bool((data[have] == src[idx[have]]).all()), (tag, "a result column differs from its source rows")
    return a, b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--case", type=int, default=-1, help="run only this case number of the seed (every case reseeds the generator)")
    ap.add_argument("--max-build", type=int, default=4_000_000, help="largest build side of the 'big' cases (every third case)")
    ap.add_argument("--max-probe", type=int, default=60_000_000)
    ap.add_argument("--force", action="append", default=[], metavar="NAME[=VALUE]",
                    help="path switches set through gdf_amd_debug_force for the whole run, e.g. --force GDF_JK_FORCE_FB=15 --force GDF_JK_FORCE_L6")
    a = ap.parse_args()
    import torch
    os.environ.setdefault("LIBGDF_AMD_TESTHOOK", "1")      # path switches (--force ...) go through libgdf_testhook.so: loaded in front of libgdf.so
    import libgdf_amd as gdf
    from libgdf_amd.columns import Column
    for f in a.force:
        name, _, value = f.partition("=")
        gdf.libgdf.gdf_amd_debug_force(name.encode(), (value or "1").encode())
    g = torch.Generator(device="cuda")
    g.manual_seed(a.seed)
    r = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=g, device="cuda"))
    t0 = time.time()
    it = 0 if a.case < 0 else a.case
    while time.time() - t0 < a.seconds:
        g.manual_seed(a.seed * 1_000_003 + it)          # a failing case can be re-run alone: --seed S --case N
        big = it % 3 == 0
        nb = r(1_000, a.max_build if big else 300_000)
        npr = r(10_000, a.max_probe if big else 2_000_000)
        spread = [0.25, 0.5, 1.0, 1.5, 3.0, 12.0, 1000.0][r(0, 7)]        # < 1: repeated build keys; > 1: probes that miss
        space = min(max(2, int(nb * spread)), 600_000_000)      # (bincount below allocates `space` counters)
        dtype = torch.int64 if r(0, 3) else torch.int32
        base = [0, 0, 1 << 40, -1_000_000, (1 << 62) - space - 5][r(0, 5)]
        if dtype == torch.int32 and abs(base) > (1 << 30):
            base = -1_000_000
        skew = r(0, 4) == 0
        build = torch.randint(0, space, (nb,), generator=g, device="cuda")
        probe = torch.randint(0, space, (npr,), generator=g, device="cuda")
        if skew:                                                          # a tenth of the probe rows share one key
            probe[torch.randint(0, npr, (npr // 10,), generator=g, device="cuda")] = int(build[0])
        mult = torch.bincount(build, minlength=space)
        per_row = mult[probe]
        expected = int(per_row.sum())
        if expected >= 2**31 - 1 or expected > 900_000_000:
            it += 1
            if a.case >= 0:
                break
            continue
        bk, pk = (build + base).to(dtype), (probe + base).to(dtype)
        # WIDE keys (round 6: the ten-byte tuples of csrc/join.hip p10_key): a bijection of the 62-bit word spreads the keys over 2^62,
        # multiplicities as before
        wide = dtype == torch.int64 and r(0, 3) == 0
        if wide:
            M = (1 << 62) - 1
            bk, pk = ((bk & M) * 0x1E3779B97F4A7C15) & M, ((pk & M) * 0x1E3779B97F4A7C15) & M
            if base < 0:                                                    # (& M folded negative keys: recompute the multiplicities on the images)
                uk, inv = torch.unique(torch.cat([bk, pk]), return_inverse=True)
                mult2 = torch.bincount(inv[:nb], minlength=uk.numel())
                per_row = mult2[inv[nb:]]
                expected = int(per_row.sum())
        how = "left" if r(0, 4) == 0 else "inner"
        tag = (it, nb, npr, space, str(dtype), base, skew, how, "wide" if wide else "narrow")
        if os.environ.get("GDF_STRESS_VERBOSE"):
            print("case", tag, "expected", expected, flush=True)
        # result_cols (round 6: one or TWO carried 8-byte payload words per side, csrc/join.hip PayCarry modes 1 / 4; the rest gathered): in a
        # third of the cases the join also materialises 1 - 3 non-key columns per side; every result row must be its source row's values
        npay = r(1, 4) if (r(0, 3) == 0 and expected + npr < 400_000_000) else 0
        if npay:
            li, ri = join_with_payloads(gdf, torch, how, pk, bk, npay, tag)
        else:
            li, ri = gdf.api.join([Column(pk)], [Column(bk)], how=how)
        torch.cuda.synchronize()
        lonely = int((per_row == 0).sum()) if how == "left" else 0
        assert li.numel() == expected + lonely, (tag, li.numel(), expected, lonely)
        l, rr = li.long(), ri.long()
        hit = rr >= 0
        assert int((~hit).sum()) == lonely, tag
        assert bool((pk[l[hit]] == bk[rr[hit]]).all()), tag
        pair = l[hit] * nb + rr[hit]
        assert int(torch.unique(pair).numel()) == expected, tag
        if how == "left":
            assert bool((per_row[l[~hit]] == 0).all()), tag
            assert int(torch.unique(l[~hit]).numel()) == lonely, tag
        it += 1
        del li, ri, l, rr, hit, pair, build, probe, bk, pk, mult, per_row
        if a.case >= 0:
            break
    print(f"stress_join: {it} joins in {time.time() - t0:.0f} s, all properties hold (seed {a.seed})")


if __name__ == "__main__":
    main()
