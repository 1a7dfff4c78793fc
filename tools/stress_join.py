#!/usr/bin/env python
"""Randomised property check of gdf_inner_join / gdf_left_join at sizes that reach every partition layout and probe path
(host-built and device-built units, speculative and exact layouts, lean / general / chained kernels, optimistic, sparse
optimistic + compaction, count + write).  No oracle: the number of pairs equals the sum over probe rows of the key's
multiplicity in the build relation, every pair joins equal keys, no pair occurs twice, and a LEFT join adds exactly the
probe rows without a partner.  Usage: python tools/stress_join.py [--seconds S] [--seed N]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


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
