#!/usr/bin/env python
"""The shapes bench.py's headline does NOT take: C3-sized joins whose keys, hit rate or multiplicity
leave the fast paths (NARROW tuples / histogram-free probe layout / optimistic single write pass), and C2 with sparse keys.
One JSON line per shape: ms, probe rows/s, end-to-end algorithmic GB/s (8 B x (probe + build + output pairs); group-by:
16 B per row) and its fraction of the 8 TB/s HBM peak, plus the per-kernel HIP-event times inside libgdf.so.
Usage: python tools/bench_shapes.py [--probe-rows N] [--reps R] [--only name,name]"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--probe-rows", type=int, default=1_000_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    import torch
    import libgdf_amd as gdf
    from bench import settle_placement, make_build_keys, make_probe_keys, read_profile, splitmix64_torch
    from libgdf_amd._binding import rmmOptions_t
    from libgdf_amd.columns import Column
    gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
    lib = gdf._binding._gdf_cdll
    dev = torch.device("cuda", 0)
    npr = a.probe_rows
    nb = npr // 10
    only = set(x for x in a.only.split(",") if x)

    def timed(name, fn, alg_bytes_fn, rows, note):
        if only and name not in only:
            return
        # warm calls until the pool has reached its steady state AND the placement searches of this shape's buffers have settled
        # (round 6: a call spends a bounded time on candidate blocks, the searches go on over the first two to four calls)
        settle_placement(gdf, fn)
        out = fn()
        lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.reps):
            out = fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.reps
        lib.gdf_amd_profile_enable(0)
        prof = read_profile(gdf)
        alg = alg_bytes_fn(out)
        print(json.dumps({"shape": name, "note": note, "rows": rows, "out_rows": out, "ms": dt * 1e3, "rows_per_s": rows / dt,
                          "algorithmic_GBps": alg / dt / 1e9, "frac_of_8TBps": alg / dt / 8e12,
                          "kernels_ms": {k: round(v[0] / a.reps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}), flush=True)

    def join(probe, build, how="inner"):
        li, ri = gdf.api.join([Column(probe)], [Column(build)], how=how, copy=False)
        n = li.numel() if hasattr(li, "numel") else int(li.size)
        del li, ri
        return n

    def gather_keys(table, n, seed):
        """table[idx] for n pseudo-random idx, in slices"""
        out = torch.empty(n, dtype=table.dtype, device=dev)
        step = 1 << 27
        for s in range(0, n, step):
            e = min(n, s + step)
            out[s:e] = table[make_probe_keys(e - s, table.numel(), seed + s, dev)]
        return out

    jb = lambda nprobe, nbuild: (lambda out: 8.0 * nprobe + 8.0 * nbuild + 8.0 * out)

    # 0. the headline itself, for reference on the same box
    build = make_build_keys(nb, 0x5EED0001, dev)
    probe = make_probe_keys(npr, nb, 0x5EED0002, dev)
    timed("c3_headline", lambda: join(probe, build), jb(npr, nb), npr, "int64 keys in [0, 1e8): NARROW tuples, histogram-free probe layout, optimistic write pass")
    # 0a. C3 variant B: validity masks on BOTH key columns -- all ones (same pairs as the headline; + 2 x ceil(N / 8)
    # bytes of mask reads) and Bernoulli(0.99) on both sides (98 % of the probe rows find a valid partner)
    def packed_mask(n, seed, p_null):
        ok = torch.ones(n, dtype=torch.bool, device=dev)
        if p_null > 0:
            step = 1 << 27
            for s in range(0, n, step):
                e = min(n, s + step)
                u = (splitmix64_torch(torch.arange(s, e, dtype=torch.int64, device=dev) + seed) >> 11) & ((1 << 53) - 1)
                ok[s:e] = u >= int((1 << 53) * p_null)
        pad = (-n) % 8
        bits = torch.cat([ok, torch.zeros(pad, dtype=torch.bool, device=dev)]) if pad else ok
        weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=dev)
        packed = torch.empty((n + 7) // 8, dtype=torch.uint8, device=dev)
        step = 1 << 27
        for s in range(0, packed.numel(), step):
            e = min(packed.numel(), s + step)
            packed[s:e] = (bits[8 * s:8 * e].view(-1, 8).to(torch.uint8) * weights).sum(dim=1, dtype=torch.uint8)
        return packed, int(n - ok.sum())

    def join_masked(pm, pn, bm, bn):
        li, ri = gdf.api.join([Column(probe, pm, null_count=pn)], [Column(build, bm, null_count=bn)], how="inner", copy=False)
        n = li.numel() if hasattr(li, "numel") else int(li.size)
        del li, ri
        return n
    jbm = lambda out: 8.0 * npr + 8.0 * nb + 8.0 * out + (npr + 7) // 8 + (nb + 7) // 8
    for name, p_null, note in (("c3_masked", 0.0, "C3 variant B: ALL-ONES validity masks on both key columns (paired data + mask reads; bytes = headline + 2 x ceil(N / 8))"),
                               ("c3_masked_99pct_valid", 0.01, "the same with Bernoulli(0.99) masks on both sides: 98 % of the probe rows join")):
        if only and name not in only:
            continue
        pm, pn = packed_mask(npr, 0x5EED0072, p_null)
        bm, bn = packed_mask(nb, 0x5EED0071, p_null)
        timed(name, lambda: join_masked(pm, pn, bm, bn), jbm, npr, note)
        del pm, bm
    # 0b. the same join with result_cols: [probe payload, key, build payload] materialised (joining.cu:375-479)
    def join_materialise():
        from libgdf_amd import gdf_column, libgdf, new_context
        from libgdf_amd.columns import column_array
        res = [gdf_column(), gdf_column(), gdf_column()]
        res_arr = (C.POINTER(gdf_column) * 3)(*[C.pointer(r) for r in res])
        li, ri = gdf_column(), gdf_column()
        ctx = new_context()
        libgdf.gdf_inner_join(column_array([Column(ppay), Column(probe)]), 2, (C.c_int * 1)(1), column_array([Column(build), Column(bpay)]), 2,
                              (C.c_int * 1)(0), 1, 3, res_arr, C.byref(li), C.byref(ri), C.byref(ctx))
        n = int(li.size)
        for c in res + [li, ri]:
            libgdf.gdf_column_free(C.byref(c))
        return n
    if not only or "c3_materialise_2_payload_cols" in only or "c3_materialise_4_payload_cols" in only:
        ppay = torch.arange(npr, dtype=torch.int64, device=dev)
        bpay = torch.arange(nb, dtype=torch.int64, device=dev)
        timed("c3_materialise_2_payload_cols", join_materialise, lambda out: 8.0 * npr + 8.0 * nb + 8.0 * out + 8.0 * npr + 8.0 * nb + 3 * 8.0 * out, npr,
              "C3 + result_cols = [int64 probe payload, key, int64 build payload]: bytes = join + both payload columns read once + three "
              "8-byte result columns written; the probe payload and the key are CARRIED through the partition passes (Tuples::pay), "
              "the build payload sits next to the build tuple in the LDS image and is written per pair (jk_probe_bp)")
        # 0c. two int64 payload columns per side: one word per side is carried, the second column of each side is gathered
        def join_materialise4():
            from libgdf_amd import gdf_column, libgdf, new_context
            from libgdf_amd.columns import column_array
            res = [gdf_column() for _ in range(5)]
            res_arr = (C.POINTER(gdf_column) * 5)(*[C.pointer(r) for r in res])
            li, ri = gdf_column(), gdf_column()
            ctx = new_context()
            libgdf.gdf_inner_join(column_array([Column(ppay), Column(ppay2), Column(probe)]), 3, (C.c_int * 1)(2),
                                  column_array([Column(build), Column(bpay), Column(bpay2)]), 3, (C.c_int * 1)(0), 1, 5, res_arr, C.byref(li), C.byref(ri),
                                  C.byref(ctx))
            n = int(li.size)
            for c in res + [li, ri]:
                libgdf.gdf_column_free(C.byref(c))
            return n
        if not only or "c3_materialise_4_payload_cols" in only:
            ppay2 = ppay * 3
            bpay2 = bpay * 5
            timed("c3_materialise_4_payload_cols", join_materialise4,
                  lambda out: 8.0 * npr + 8.0 * nb + 8.0 * out + 2 * 8.0 * npr + 2 * 8.0 * nb + 5 * 8.0 * out, npr,
                  "C3 + result_cols = [2 x int64 probe payload, key, 2 x int64 build payload]: round 6 -- BOTH probe payload columns travel with the "
                  "tuples (16-byte payload elements, PayCarry mode 4), the key comes out of the probe kernel, the build side's first column travels "
                  "and its second is staged by build row into the LDS image; no gather (rounds 4 - 5: one word per side carried, 32 ms of gathers)")
            del ppay2, bpay2
        del ppay, bpay
    # 1. half of the probe rows miss: count pass + write pass
    probe_half = make_probe_keys(npr, 2 * nb, 0x5EED0012, dev)
    timed("c3_half_hit", lambda: join(probe_half, build), jb(npr, nb), npr, "probe keys in [0, 2e8): 50 % hit rate -> sample rejects the optimistic pass: count + write")
    timed("c3_left_half_hit", lambda: join(probe_half, build, how="left"), jb(npr, nb), npr, "the same as a LEFT join: unmatched probe rows emit (l, -1)")
    del probe_half
    probe_80 = make_probe_keys(npr, nb + nb // 4, 0x5EED0014, dev)
    timed("c3_80pct_hit", lambda: join(probe_80, build), jb(npr, nb), npr, "probe keys in [0, 1.25e8): 80 % hit rate")
    del probe_80
    probe_tenth = make_probe_keys(npr, 10 * nb, 0x5EED0013, dev)
    timed("c3_tenth_hit", lambda: join(probe_tenth, build), jb(npr, nb), npr, "probe keys in [0, 1e9): 10 % hit rate -> one optimistic write pass into per-unit slots + compaction of the pair list")
    del probe_tenth
    # 2. int32 keys
    b32, p32 = build.to(torch.int32), probe.to(torch.int32)
    timed("c3_int32_keys", lambda: join(p32, b32), lambda out: 4.0 * npr + 4.0 * nb + 8.0 * out, npr, "int32 key columns")
    del b32, p32
    # 3. skewed probe side: Zipf(s=1) over the build keys
    pz = torch.empty(npr, dtype=torch.int64, device=dev)
    step = 1 << 26
    for s in range(0, npr, step):
        e = min(npr, s + step)
        i = torch.arange(s, e, dtype=torch.int64, device=dev)
        u = ((splitmix64_torch(i + 0x5EED0021) >> 11) & ((1 << 53) - 1)).double() / float(1 << 53)
        pz[s:e] = build[torch.clamp(torch.exp(u * math.log(nb + 1.0)).long() - 1, 0, nb - 1)]
        del i, u
    timed("c3_zipf_probe", lambda: join(pz, build), jb(npr, nb), npr, "probe keys Zipf(s=1) over the build keys (hottest key: ~5 % of the rows)")
    del pz, probe
    # 4. keys spread over 2^60: WIDE (12-byte) tuples
    i = torch.arange(nb, dtype=torch.int64, device=dev)
    bwide = (splitmix64_torch(i + 0x5EED0031) >> 4) & ((1 << 60) - 1)          # splitmix64 is a bijection: distinct before the shift;
    del i                                                                       # a handful of collisions after it do not matter
    pwide = gather_keys(bwide, npr, 0x5EED0032)
    timed("c3_wide_keys", lambda: join(pwide, bwide), jb(npr, nb), npr, "int64 keys spread over 2^60: round 6 -- ten-byte tuples (six bytes of hash remainder + row, the key's high word beside them), lean probe kernel at two workgroups per CU (rounds 1 - 5: key64 + row, 12 bytes)")
    del pwide, bwide
    # 5. every build key four times (multimap semantics): the probe side shrinks so that the output stays 1e9 pairs
    npd = npr // 4
    bdup = build % (nb // 4)
    pdup = make_probe_keys(npd, nb // 4, 0x5EED0042, dev)
    timed("dup4_build_keys", lambda: join(pdup, bdup), jb(npd, nb), npd, f"{npd} probe rows x {nb} build rows, every build key 4 times: chained multimap units, count + write")
    del pdup, bdup, build

    # C2 and its sparse twin
    n2 = 100_000_000
    vals = make_probe_keys(n2, 1000, 0x5EED0004, dev)
    dense = make_probe_keys(n2, 10_000, 0x5EED0003, dev)
    i = torch.arange(10_000, dtype=torch.int64, device=dev)
    lut = (splitmix64_torch(i + 0x5EED0051) >> 2)                                  # 10 k keys scattered over 2^62
    sparse = lut[dense]
    for name, k, note in (("c2_dense_keys", dense, "10 k keys in [0, 1e4): direct-index path"),
                          ("c2_sparse_keys", sparse, "the same 10 k groups under keys scattered over 2^62: LDS dictionary path (sample -> image -> 2-byte ids -> LDS accumulators)")):
        kc, vc = Column(k), Column(vals)
        timed(name, lambda: int(gdf.api.group_by("sum", [kc], vc, capacity=1 << 20)[1].numel()), lambda out: 16.0 * n2, n2,
              "gdf_group_by_sum int64 keys / int64 values, " + note)


if __name__ == "__main__":
    main()
