#!/usr/bin/env python
"""BASELINE config C5: multi-key gdf_group_by_avg, fp64 values with a 50 %-null validity mask,
keys k0 int64 Zipf(s=1) over 1e6 values x k1 int32 uniform [0,16), through the C ABI with inputs resident in HBM.
Algorithmic bytes = N * (8 + 4 + 8) + N / 8 (value mask).  Checks the result PER GROUP against a torch fp64 scatter_add_ / bincount
over the dense key space (keys, exact counts, averages within 1e-6) next to the size-independent totals.
--null-keys P: the variant with P (1 %) NULL KEYS -- a validity mask on key column 0; a row with a null key belongs
to no group (the reference rejects masks altogether, sqls_ops.cu:1103-1106; semantics of DESIGN.md section 4).
Usage: python tools/bench_c5.py [--rows N] [--reps R] [--null-keys P]"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_c5(n, dev, zipf_values=1_000_000):
    """The C5 relation: k0 int64 Zipf(s=1) over `zipf_values` values (inverse CDF of p(r) ~ 1/r), k1 int32 uniform [0, 16),
    fp64 values uniform [0, 1), value validity 50 % -> (k0, k1, v, ok bool tensor, LSB-first mask bytes)."""
    import torch
    from bench import splitmix64_torch
    k0 = torch.empty(n, dtype=torch.int64, device=dev)
    k1 = torch.empty(n, dtype=torch.int32, device=dev)
    v = torch.empty(n, dtype=torch.float64, device=dev)
    ok = torch.empty(n, dtype=torch.bool, device=dev)
    step = 1 << 26
    M = zipf_values
    for s in range(0, n, step):
        e = min(n, s + step)
        i = torch.arange(s, e, dtype=torch.int64, device=dev)
        u = ((splitmix64_torch(i + 0x5EED0005) >> 11) & ((1 << 53) - 1)).double() / float(1 << 53)
        k0[s:e] = torch.clamp(torch.exp(u * math.log(M + 1.0)).long() - 1, 0, M - 1)
        k1[s:e] = ((splitmix64_torch(i + 0x5EED0006) >> 1) % 16).int()
        v[s:e] = ((splitmix64_torch(i + 0x5EED0007) >> 11) & ((1 << 53) - 1)).double() / float(1 << 53)
        ok[s:e] = (splitmix64_torch(i + 0x5EED0008) >> 63) == 0
        del i, u
    pad = (-n) % 8
    bits = torch.cat([ok, torch.zeros(pad, dtype=torch.bool, device=dev)]).view(-1, 8).to(torch.uint8)
    weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=dev)
    mask = (bits * weights).sum(dim=1, dtype=torch.int32).to(torch.uint8)
    mask = torch.cat([mask, torch.zeros((-mask.numel()) % 64, dtype=torch.uint8, device=dev)])
    return k0, k1, v, ok, mask


def pack_bits(okb):
    """bool tensor -> LSB-first mask bytes, padded to a multiple of 64 bytes"""
    import torch
    n = okb.numel()
    pad = (-n) % 8
    bits = torch.cat([okb, torch.zeros(pad, dtype=torch.bool, device=okb.device)]).view(-1, 8).to(torch.uint8)
    weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=okb.device)
    mask = (bits * weights).sum(dim=1, dtype=torch.int32).to(torch.uint8)
    return torch.cat([mask, torch.zeros((-mask.numel()) % 64, dtype=torch.uint8, device=okb.device)])


def null_key_mask(n, dev, p):
    """key validity: a row's key 0 is null with probability p -> (bool tensor, mask bytes)"""
    import torch
    from bench import splitmix64_torch
    kok = torch.empty(n, dtype=torch.bool, device=dev)
    step = 1 << 26
    for s in range(0, n, step):
        e = min(n, s + step)
        u = (splitmix64_torch(torch.arange(s, e, dtype=torch.int64, device=dev) + 0x5EED0009) >> 11) & ((1 << 53) - 1)
        kok[s:e] = u >= int((1 << 53) * p)
    return kok, pack_bits(kok)


def c5_property_checks(gdf, k0, k1, v, ok, mask, cap, kok=None, kmask=None):
    """Size-independent properties of gdf_group_by_avg / _count over the C5 relation (what a 1e9-row run can be held to):
    group count = number of distinct key pairs, AVG output sorted by key, AVG and COUNT name the same groups, the counts
    add up to the number of valid values, a group is null exactly when its count is 0, and sum(avg * count) = sum of
    the valid values.  -> (checks dict, all good)"""
    import torch
    from libgdf_amd.columns import Column
    n = k0.numel()
    kc = [Column(k0, kmask, null_count=int(n - kok.sum().item())) if kok is not None else Column(k0), Column(k1)]
    vc = Column(v, mask, null_count=int(n - ok.sum().item()))
    if kok is not None:                      # rows with a null key belong to no group: the expectations are taken over the others
        k0, k1, v, ok = k0[kok], k1[kok], v[kok], ok[kok]
    gk, avg, avg_ok = gdf.api.group_by("avg", kc, vc, out_dtype=6, capacity=cap, with_masks=True)
    ck, cnt, _ = gdf.api.group_by("count", kc, vc, out_dtype=4, capacity=cap, with_masks=True)
    pk_avg = gk[0] * 16 + gk[1].long()
    pk_cnt = ck[0] * 16 + ck[1].long()
    order = torch.argsort(pk_cnt)
    cnt_sorted = cnt[order]
    avg_ok = avg_ok.to(avg.device)
    checks = {
        "groups": int(avg.numel()),
        "groups_expected": int(torch.unique(k0 * 16 + k1.long()).numel()),
        "avg_keys_sorted": bool((pk_avg[1:] > pk_avg[:-1]).all().item()),
        "same_keys_avg_and_count": bool(torch.equal(pk_avg, pk_cnt[order])),
        "sum_of_counts": int(cnt.sum().item()), "valid_values": int(ok.sum().item()),
        "null_groups": int((~avg_ok).sum().item()), "zero_count_groups": int((cnt == 0).sum().item()),
        "null_iff_zero_count": bool(torch.equal(~avg_ok, cnt_sorted == 0)),
    }
    total = float((avg.double() * cnt_sorted.double()).sum().item())
    expect = float(v[ok].sum().item())
    checks["sum_avg_times_count_rel_err"] = abs(total - expect) / expect
    # PER GROUP (totals alone would let two groups swap their sums): the key space is dense -- pk = k0 * 16 + k1
    # below 16 * max(k0) + 16 -- so a torch fp64 scatter_add_ / bincount over pk is an independent per-group reference at any size:
    # every group's key present, its count EXACT, its average within 1e-6 relative (north_star's fp tolerance; measured ~1e-13).
    span = int(k0.max().item()) * 16 + 16
    ref_sum = torch.zeros(span, dtype=torch.float64, device=k0.device)
    ref_cnt = torch.zeros(span, dtype=torch.int64, device=k0.device)
    ref_rows = torch.zeros(span, dtype=torch.int64, device=k0.device)
    step = 1 << 27
    for s in range(0, k0.numel(), step):
        pk = k0[s:s + step] * 16 + k1[s:s + step].long()
        okc = ok[s:s + step]
        ref_rows += torch.bincount(pk, minlength=span)
        pkv = pk[okc]
        ref_cnt += torch.bincount(pkv, minlength=span)
        ref_sum.scatter_add_(0, pkv, v[s:s + step][okc])
        del pk, pkv, okc
    ref_keys = torch.nonzero(ref_rows > 0).flatten()
    same_groups = bool(ref_keys.numel() == pk_avg.numel() and torch.equal(ref_keys, pk_avg))
    checks["per_group_keys_match"] = same_groups
    if same_groups:
        want_cnt = ref_cnt[pk_avg]
        checks["per_group_counts_exact"] = bool(torch.equal(want_cnt, cnt_sorted.long()))
        live = want_cnt > 0
        checks["per_group_avg_max_rel_err"] = 0.0
        if bool(live.any().item()):
            want_avg = ref_sum[pk_avg][live] / want_cnt[live].double()
            checks["per_group_avg_max_rel_err"] = float(((avg.double()[live] - want_avg).abs() / want_avg.abs().clamp_min(1e-300)).max().item())
        checks["per_group_null_avg_is_zero"] = bool((avg.double()[~live] == 0).all().item())
    else:
        checks["per_group_counts_exact"] = False
        checks["per_group_avg_max_rel_err"] = float("inf")
        checks["per_group_null_avg_is_zero"] = False
    del ref_sum, ref_cnt, ref_rows
    good = (checks["groups"] == checks["groups_expected"] and checks["avg_keys_sorted"] and checks["same_keys_avg_and_count"]
            and checks["sum_of_counts"] == checks["valid_values"] and checks["null_iff_zero_count"]
            and checks["sum_avg_times_count_rel_err"] < 1e-9
            and checks["per_group_keys_match"] and checks["per_group_counts_exact"] and checks["per_group_avg_max_rel_err"] < 1e-6
            and checks["per_group_null_avg_is_zero"])
    return checks, good


def run_c5(gdf, dev, n=1_000_000_000, reps=3, null_keys=0.0, checks=True):
    """one C5 measurement through the C ABI (inputs resident in HBM, outputs preallocated, the timed region is the C call alone)
    -> the result dict of this tool's JSON line.  Also called by bench.py for its `extra.c5` object."""
    import torch
    from bench import read_profile
    from libgdf_amd.columns import Column, column_array, new_context
    lib = gdf._binding._gdf_cdll
    k0, k1, v, ok, mask = make_c5(n, dev)
    kok = kmask = None
    if null_keys > 0:
        kok, kmask = null_key_mask(n, dev, null_keys)
    kc = [Column(k0, kmask, null_count=int(n - kok.sum().item())) if kok is not None else Column(k0), Column(k1)]
    vc = Column(v, mask, null_count=int(n - ok.sum().item()))
    cap = min(20_000_000, n)
    alg = n * 20.0 + n / 8.0 + (n / 8.0 if kok is not None else 0.0)

    def out_col(tdtype, gdtype):
        return Column(torch.empty(cap, dtype=tdtype, device=dev), torch.zeros((cap + 7) // 8 + 64, dtype=torch.uint8, device=dev), gdtype, size=cap)
    ok0, ok1, oagg = out_col(torch.int64, 4), out_col(torch.int32, 3), out_col(torch.float64, 6)
    ka, oa = column_array(kc), column_array([ok0, ok1])
    ctx = new_context(method=1)
    call = lambda: gdf.libgdf.gdf_group_by_avg(2, ka, vc.ptr, None, oa, oagg.ptr, C.byref(ctx))
    from bench import settle_placement
    settle_placement(gdf, call, 6, 1)       # warm-up: one call, more while the record buffer's placement search is still open (round 6: budgeted per call)
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    lib.gdf_amd_profile_enable(0)
    prof = read_profile(gdf)
    del ok0, ok1, oagg
    checks_d, good = c5_property_checks(gdf, k0, k1, v, ok, mask, cap, kok, kmask) if checks else ({}, True)
    return {"op": "C5 gdf_group_by_avg (int64 Zipf x int32) keys, fp64 values, 50% null" + (f", {null_keys:g} null keys" if kok is not None else ""),
            "rows": n, "ms": dt * 1e3,
            "rows_per_s": n / dt, "algorithmic_GBps": alg / dt / 1e9, "frac_of_8TBps": alg / dt / 8e12,
            "kernels_ms": {k: round(x[0] / reps, 3) for k, x in prof.items()}, "checks": checks_d, "checks_pass": good}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--null-keys", type=float, default=0.0)
    ap.add_argument("--no-checks", action="store_true", help="timing only (LAB ablations that break the result on purpose)")
    ap.add_argument("--place-draws", type=int, default=None, help="librmm gdf_amd_rmm_place_draws: 0 = no placement search (PMC runs)")
    ap.add_argument("--force", action="append", default=[], metavar="NAME[=VALUE]",
                    help="path switches set through gdf_amd_debug_force for the whole run, e.g. --force GDF_GBP_PLAIN_RANK=0")
    a = ap.parse_args()
    import torch
    if a.force:
        os.environ["LIBGDF_AMD_TESTHOOK"] = "1"      # path switches go through libgdf_testhook.so, loaded in front of libgdf.so
    import libgdf_amd as gdf
    from libgdf_amd._binding import rmmOptions_t
    gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
    if a.place_draws is not None:
        gdf._binding._rmm_cdll.gdf_amd_rmm_place_draws(C.c_int(a.place_draws))
    for sw in a.force:
        name, _, value = sw.partition("=")
        gdf.libgdf.gdf_amd_debug_force(name.encode(), (value or "1").encode())
    res = run_c5(gdf, torch.device("cuda", 0), a.rows, a.reps, a.null_keys, not a.no_checks)
    print(json.dumps(res))
    sys.exit(0 if res["checks_pass"] else 1)


if __name__ == "__main__":
    main()
