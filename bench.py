#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: hash-join probe rows/s (+ HBM roofline fraction), 1 B int64 rows.

A "step" is ONE gdf_inner_join call (HASH method) through the C ABI of libgdf.so over synthetic columns
that are already resident in HBM:
  N = 1  : config C3 (BASELINE.json configs[2]) -- probe 1,000,000,000 int64 rows, build 100,000,000
           unique int64 keys (a seeded permutation of [0, N_b)), probe key = splitmix64(seed + i) % N_b, so
           every probe row matches exactly once and N_out = N_p; indices only (result_cols = NULL).
  N > 1  : config C4 -- the same per-rank shard sizes (1e9 probe + 1.25e8 build rows per GPU over a global
           key space), both relations partitioned on key across the ranks and exchanged with RCCL all-to-alls
           (libgdf_amd/multigpu.py: the fused join, else the key shuffle -- the SAME algorithm at every N; what
           the planner would have picked is timed after it and reported as planner_choice), then joined
           locally.  Weak scaling.  An untimed preflight step checks the global pair count and 2^20 sampled
           pairs per rank (equal keys) before anything is timed.
value = probe rows of all ranks / max-over-ranks time.  The roofline object prices the WHOLE call with the algorithmic
bytes of SURVEY.md 8d (8*N_p + 8*N_b + 8*N_out; the partition passes count as overhead) and carries the probe phase
(16 B per probe row over the probe-side launches, live HIP-event timing inside libgdf.so, see csrc/prof.h) and the PMC
traffic of one join; roofline_kernel prices the dominant kernel on its own contract.  cpu_baseline times oracle/gdf_oracle.c (a
single-threaded C port of the reference algorithm) on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec

# per-kernel algorithmic bytes per step (DESIGN.md "Kernels"): what the kernel must move by its own contract.
# tb = bytes per (key, row) tuple: 8 when the keys fit 32 bits (C3/C4: key space <= 2^32), else 12.
# lps = launches per step: jk_hist runs on the build side only (1 launch) when the probe side takes the
# histogram-free layout, on both relations (2 launches) otherwise.
# kb = bytes per key the join reads: 8 (C3: the caller's int64 column), 4 on the multi-GPU path (keys travel narrowed).
KERNEL_BYTES = {
    "jk_hist": lambda npr, nb, tb, lps, kb: kb * nb + (kb * npr if lps > 1.5 else 0.0),   # reads every key once
    # key in, tuple out (the probe side's level-1 tuples are six bytes too on the single-GPU main path: csrc/join.hip L6)
    "jk_scatter1": lambda npr, nb, tb, lps, kb: (kb + tb) * nb + (kb + l6_bytes(npr, nb, tb, kb)) * npr,
    # (six-byte level-2 tuples on the probe side -- csrc/join.hip p6_store -- when the keys are NARROW and the build side needs 2^15
    # partitions: the probe side's level-2 output and the probe kernel's input are then 6 B per tuple)
    "jk_scatter2": lambda npr, nb, tb, lps, kb: (tb + tb) * nb + (l6_bytes(npr, nb, tb, kb) + p6_bytes(nb, tb)) * npr,         # tuple in, tuple out
    "jk_probe_count": lambda npr, nb, tb, lps, kb: tb * nb + p6_bytes(nb, tb) * npr,                    # tuples in
    "jk_probe_write": lambda npr, nb, tb, lps, kb: tb * nb + p6_bytes(nb, tb) * npr + 8.0 * npr,        # tuples in, one int32 index pair per probe row out
    # the sender side of the shuffle (multi-GPU only): int64 keys in; narrowed key + int32 row number out
    "shuffle_hist": lambda npr, nb, tb, lps, kb: 8.0 * (npr + nb),
    "shuffle_scatter": lambda npr, nb, tb, lps, kb: (8.0 + kb + 4.0) * (npr + nb),
    # the stable variant the join uses: keys out, one bit per row and destination instead of a row number
    "stable_count": lambda npr, nb, tb, lps, kb: 8.0 * (npr + nb),
    "stable_scatter": lambda npr, nb, tb, lps, kb: (8.0 + kb + 0.125) * (npr + nb),
    # the fused variant: raw int64 key in, narrowed key + row number out (the row numbers stay with the sender)
    "fj_scatter": lambda npr, nb, tb, lps, kb: (8.0 + 4.0 + 4.0) * (npr + nb),
}


def p6_bytes(nb, tb):
    """bytes per probe-side level-2 tuple: 6 when the join takes the six-byte tuples (NARROW keys, >= 3072 * 2^14 build rows), else tb"""
    return 6.0 if (tb == 8.0 and nb >= 3072 * 2 ** 14) else tb


def l6_bytes(npr, nb, tb, kb):
    """bytes per probe-side level-1 tuple: 6 where the join takes the six-byte level-1 tuples (csrc/join.hip L6: six-byte level-2
    tuples, 2^26 <= probe rows < 2^30 - 2^23, the single-GPU path reading the caller's 8-byte keys), else tb"""
    return 6.0 if (p6_bytes(nb, tb) == 6.0 and kb == 8.0 and 2 ** 26 <= npr < 2 ** 30 - 2 ** 23) else tb


def splitmix64_torch(x):
    """splitmix64 on int64 tensors (two's-complement wrap == uint64 arithmetic); logical shifts emulated."""
    import torch

    def lsr(v, k):
        return (v >> k) & ((1 << (64 - k)) - 1)
    z = x + (-7046029254386353131)            # 0x9E3779B97F4A7C15
    z = (z ^ lsr(z, 30)) * (-4658895280553007687)    # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * (-7723592293110705685)    # 0x94D049BB133111EB
    return z ^ lsr(z, 31)


def make_probe_keys(n, key_space, seed, device, offset=0):
    """probe[i] = (splitmix64(seed + offset + i) >> 1) % key_space, generated in slices to bound temporaries."""
    import torch
    out = torch.empty(n, dtype=torch.int64, device=device)
    step = 1 << 27
    for s in range(0, n, step):
        e = min(n, s + step)
        i = torch.arange(s + offset, e + offset, dtype=torch.int64, device=device) + seed
        out[s:e] = ((splitmix64_torch(i) >> 1) & 0x7FFFFFFFFFFFFFFF) % key_space
    return out


def make_build_keys(n, seed, device):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randperm(n, dtype=torch.int64, device=device, generator=g)


def read_profile(gdf, split_sides=False):
    """{kernel name: (total ms, launches)} since the last reset.  The join tags its probe-side launches "name@probe" (csrc/prof.h);
    split_sides=False folds them into the plain name."""
    lib = gdf._binding._gdf_cdll
    names = ((C.c_char * 64) * 64)()
    ms = (C.c_double * 64)()
    cnt = (C.c_int * 64)()
    lib.gdf_amd_profile_read.restype = C.c_int
    k = lib.gdf_amd_profile_read(names, ms, cnt, 64)
    out = {}
    for i in range(min(k, 64)):
        name = names[i].value.decode()
        if not split_sides:
            name = name.split("@")[0]
        a, b = out.get(name, (0.0, 0))
        out[name] = (a + ms[i], b + cnt[i])
    return out


def settle_placement(gdf, fn, max_calls=6, min_calls=2):
    """Untimed warm-up for a steady-state measurement: call fn() until the pool's placement searches have settled (librmm
    gdf_amd_rmm_place_stats: no entry still exploring) -- since round 6 a call spends a bounded time on candidate blocks and the
    searches of a new shape go on over its first two to four calls (csrc/internal.h PlaceRound); at most max_calls.  Returns the
    number of calls made."""
    st = (C.c_ulonglong * 4)()
    rmm = gdf._binding._rmm_cdll
    made = 0
    for _ in range(max_calls):
        fn()
        made += 1
        rmm.gdf_amd_rmm_place_stats(st)
        if st[3] == 0 and made >= min_calls:
            break
    return made


def library_build_id():
    """sha256 of the kernel SOURCES libgdf.so is built from (csrc/*): the same value here and on the GPU box, and it
    changes whenever a kernel changes -- unlike a hash of the .so, which would differ between two builds of one source."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "libgdf_amd", "csrc", "*"))):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, launches_per_step):
    """HBM bytes per launch of `kernel` (kernel=None: per JOIN, all kernels) from the newest committed rocprofv3 PMC summary
    (profiles/*_pmc_hbm.json).  bench.py cannot collect counters itself (separate --pmc passes, tools/pmc_hbm_json.py); the file
    states how they were collected and corrected and carries the build id of the kernels it measured -- counters of OTHER kernel
    sources are refused (traffic = null) instead of silently going stale."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json")))      # by name: r1_a < r1_k < r2_z (mtimes do not survive a checkout)
    if not files:
        return None, "no profiles/*_pmc_hbm.json"
    mine = library_build_id()
    seen = []
    for path in reversed(files):
        with open(path) as f:
            data = json.load(f)
        rel = os.path.relpath(path, ROOT)
        if data.get("build_id") != mine:
            seen.append(f"{rel} measured build {data.get('build_id')}")
            continue
        if kernel is None:
            ks = data.get("kernels", {})
            return (sum(k["hbm_bytes_per_join_corrected"] for k in ks.values()) if ks else None), rel
        k = data.get("kernels", {}).get(kernel)
        if not k or not launches_per_step:
            return None, rel
        return k["hbm_bytes_per_join_corrected"] / launches_per_step, rel
    return None, f"this is build {mine}; " + "; ".join(seen[:2]) + ": refused"


def cpu_baseline_pandas(sample_probe, sample_build, budget_s=60.0):
    """The CPU path BASELINE.json's north_star names: pandas.merge(how="inner") on int64 keys, the same key
    distribution as C3 at a tenth of its size.  Median of up to three runs (stops early once `budget_s` is spent:
    pandas joins ~2-5 M rows/s on one core)."""
    import numpy as np
    import pandas as pd
    from oracle import oracle
    rng = np.random.RandomState(0x5EED)
    build = rng.permutation(sample_build).astype(np.int64)
    probe = ((oracle.splitmix64(np.arange(sample_probe, dtype=np.uint64) + np.uint64(0x5EED0002)) >> np.uint64(1))
             % np.uint64(sample_build)).astype(np.int64)
    left = pd.DataFrame({"k": probe, "l": np.arange(sample_probe, dtype=np.int32)})
    right = pd.DataFrame({"k": build, "r": np.arange(sample_build, dtype=np.int32)})
    times = []
    while len(times) < 3 and sum(times) < budget_s:
        t0 = time.perf_counter()
        m = left.merge(right, on="k", how="inner")
        times.append(time.perf_counter() - t0)
        assert len(m) == sample_probe
        del m
    dt = sorted(times)[len(times) // 2]
    return {"value": sample_probe / dt, "unit": "rows/s", "cores": 1, "kind": "reference-cpu-path", "impl": f"pandas {pd.__version__} DataFrame.merge",
            "sample": f"pandas.merge(how='inner') of {sample_probe} probe x {sample_build} build int64 rows (C3 / "
                      f"{1_000_000_000 // max(sample_probe, 1)}), median of {len(times)} run(s) = {dt:.1f} s; pandas' hash join is "
                      f"single-threaded: 1 of the host's {os.cpu_count()} cores"}


def cpu_baseline(sample_probe, sample_build):
    """oracle join (single-threaded C port of the reference algorithm) on a bounded sample of C3."""
    import numpy as np
    from oracle import oracle
    rng = np.random.RandomState(0x5EED)
    build = rng.permutation(sample_build).astype(np.int64)
    probe = (oracle.splitmix64(np.arange(sample_probe, dtype=np.uint64) + np.uint64(0x5EED0002)) >> np.uint64(1)) % np.uint64(sample_build)
    probe = probe.astype(np.int64)
    oracle.lib()
    t0 = time.perf_counter()
    li, ri = oracle.join([probe], [build], "inner")
    dt = time.perf_counter() - t0
    assert len(li) == sample_probe
    return {"value": sample_probe / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"oracle/gdf_oracle.c orc_join, {sample_probe} probe x {sample_build} build int64 rows, "
                      f"{dt:.1f} s on 1 of {os.cpu_count()} host cores"}


def cpu_baseline_all_cores(sample_probe, sample_build):
    """SURVEY.md 8(d)'s optional second baseline: an OpenMP radix-partitioned hash join (oracle/gdf_oracle.c orc_join_parallel_i64,
    test / bench infrastructure) on ALL host cores, the same key distribution as C3 on a bounded sample; best of three."""
    import numpy as np
    from oracle import oracle
    rng = np.random.RandomState(0x5EED)
    build = rng.permutation(sample_build).astype(np.int64)
    probe = ((oracle.splitmix64(np.arange(sample_probe, dtype=np.uint64) + np.uint64(0x5EED0002)) >> np.uint64(1)) % np.uint64(sample_build)).astype(np.int64)
    oracle.lib()
    best, used = None, 0
    for _ in range(3):
        t0 = time.perf_counter()
        li, ri, used = oracle.join_parallel_i64(probe, build)
        dt = time.perf_counter() - t0
        assert len(li) == sample_probe
        best = dt if best is None else min(best, dt)
    return {"value": sample_probe / best, "unit": "rows/s", "cores": used, "kind": "port",
            "sample": f"oracle/gdf_oracle.c orc_join_parallel_i64 (OpenMP radix-partitioned hash join, not reference code), {sample_probe} probe x "
                      f"{sample_build} build int64 rows, best of 3 = {best:.2f} s on {used} threads of {os.cpu_count()} host cores"}


def extra_configs(gdf, dev):
    """BASELINE configs C2 (gdf_group_by_sum, 1e8 int64 keys, 1e4 groups) and C5 (gdf_group_by_avg, 1e9 rows, Zipf keys, 50 % null
    values) through the C ABI, inputs resident in HBM: {ms, frac of 8 TB/s on the config's algorithmic bytes,
    checks_pass}.  C2's check: group count, key set and the total of the sums against torch reductions over the same inputs
    (bit-exact integers); C5's: tools/bench_c5.py's size-independent properties."""
    import torch
    from libgdf_amd.columns import Column, column_array, new_context
    out = {}
    try:
        n = 100_000_000
        keys = make_probe_keys(n, 10000, 0x5EED0003, dev)          # (splitmix64(seed + i) >> 1) % 10000, as tools/bench_ops.py
        vals = make_probe_keys(n, 1000, 0x5EED0004, dev)
        cap = 16384
        okey = Column(torch.empty(cap, dtype=torch.int64, device=dev), None, 4, size=cap)
        oagg = Column(torch.empty(cap, dtype=torch.int64, device=dev), None, 4, size=cap)
        kc, vc = Column(keys), Column(vals)
        ka, oa = column_array([kc]), column_array([okey])
        ctx = new_context(method=1)
        call = lambda: gdf.libgdf.gdf_group_by_sum(1, ka, vc.ptr, None, oa, oagg.ptr, C.byref(ctx))
        call()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            call()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        ng = int(oagg.size)
        assert 0 < ng <= cap
        gk, ga = okey.data[:ng], oagg.data[:ng]
        want = torch.zeros(10000, dtype=torch.int64, device=dev).scatter_add_(0, keys, vals)
        good = ng == int((torch.bincount(keys, minlength=10000) > 0).sum().item()) and bool(torch.equal(want[gk], ga)) and \
            int(torch.unique(gk).numel()) == ng
        out["c2"] = {"op": "C2 gdf_group_by_sum, 1e8 int64 keys, 1e4 groups, HASH", "ms": dt * 1e3, "frac": n * 16.0 / dt / 8e12,
                     "algorithmic_bytes": n * 16.0, "groups": ng, "checks_pass": bool(good)}
        del keys, vals, kc, vc, okey, oagg, want
        torch.cuda.empty_cache()
    except Exception as e:                         # noqa: BLE001 -- reported, never fatal for the headline line
        out["c2"] = {"error": f"{type(e).__name__}: {e}", "checks_pass": False}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_c5
        r = bench_c5.run_c5(gdf, dev, 1_000_000_000, 3, 0.0, True)
        out["c5"] = {"op": r["op"], "ms": r["ms"], "frac": r["frac_of_8TBps"], "algorithmic_bytes": 20.375e9,
                     "kernels_ms": r["kernels_ms"], "checks_pass": bool(r["checks_pass"])}
    except Exception as e:                         # noqa: BLE001
        out["c5"] = {"error": f"{type(e).__name__}: {e}", "checks_pass": False}
    try:
        out["ops"] = extra_ops(gdf, dev)
    except Exception as e:                         # noqa: BLE001
        out["ops"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def wide_unique_keys(n, seed, device):
    """n DISTINCT int64 keys spread uniformly over [0, 2^62): a bijection of the 62-bit word applied to seed + i (multiplications by odd
    constants and xor-shifts are permutations of Z / 2^62)."""
    import torch
    M = (1 << 62) - 1
    x = (torch.arange(n, dtype=torch.int64, device=device) + seed) & M
    x = (x * 0x1E3779B97F4A7C15) & M
    x = x ^ (x >> 31)
    x = (x * 0x3F58476D1CE4E5B9) & M
    x = x ^ (x >> 29)
    x = (x * 0x14D049BB133111EB) & M
    return x ^ (x >> 32)


def extra_shapes(gdf, dev, headline_ms, npr=1_000_000_000, nb=100_000_000):
    """Shapes next to the headline, through the same C-ABI call:
    wide_keys   -- C3 with genuinely 64-bit keys, uniform over 2^62 (the headline's keys fit 32 bits): ms, fraction of 8 TB/s on the
                   same algorithmic bytes, properties of the result (pair count, no probe row twice, a 2^20-pair sample joins equal keys);
    shape_sweep -- 20 joins whose probe relation has a DIFFERENT size each time, 0.9e9 ... 1.1e9 rows (a caller whose relations change
                   from query to query; round 5's placed blocks were keyed on the exact size and searched anew every time): mean ms
                   and its ratio to the headline."""
    import torch
    from libgdf_amd import gdf_column, libgdf, new_context
    from libgdf_amd.columns import Column, column_array
    out = {}
    ctx = new_context()
    on = (C.c_int * 1)(0)

    def join(pcol, bcol, keep=False):
        li, ri = gdf_column(), gdf_column()
        libgdf.gdf_inner_join(column_array([pcol]), 1, on, column_array([bcol]), 1, on, 1, 0, None, C.byref(li), C.byref(ri), C.byref(ctx))
        n = int(li.size)
        if keep:
            return n, li, ri
        libgdf.gdf_column_free(C.byref(li))
        libgdf.gdf_column_free(C.byref(ri))
        return n

    def timed(fn, warm, reps):
        settle_placement(gdf, fn, max(warm, 6))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    try:
        build = wide_unique_keys(nb, 0x5EED0031, dev)
        pick = make_probe_keys(npr, nb, 0x5EED0032, dev)
        probe = build[pick]
        del pick
        pcol, bcol = Column(probe), Column(build)
        ms = timed(lambda: join(pcol, bcol), 5, 10)
        n, li, ri = join(pcol, bcol, keep=True)
        l = torch.empty(n, dtype=torch.int32, device=dev)
        r = torch.empty(n, dtype=torch.int32, device=dev)
        gdf.api._hipMemcpyDtoD(l.data_ptr(), li.data, n * 4)
        gdf.api._hipMemcpyDtoD(r.data_ptr(), ri.data, n * 4)
        libgdf.gdf_column_free(C.byref(li)); libgdf.gdf_column_free(C.byref(ri))
        ok = n == npr and int(l.long().sum().item()) == npr * (npr - 1) // 2 and int(l.min()) == 0 and int(l.max()) == npr - 1
        pos = torch.randint(0, n, (1 << 20,), device=dev)
        ok = ok and bool((probe[l[pos].long()] == build[r[pos].long()]).all().item()) and int(r.min()) >= 0 and int(r.max()) < nb
        ab = 8.0 * npr + 8.0 * nb + 8.0 * n
        out["wide_keys"] = {"op": f"C3 gdf_inner_join, int64 keys uniform over 2^62: {npr} probe x {nb} build rows, unique build keys, 100% hit",
                            "ms": ms, "frac": ab / (ms * 1e-3) / 8e12, "algorithmic_bytes": ab, "out_rows": n, "checks_pass": bool(ok)}
        del probe, build, pcol, bcol, l, r, pos
        torch.cuda.empty_cache()
    except Exception as e:                         # noqa: BLE001
        out["wide_keys"] = {"error": f"{type(e).__name__}: {e}", "checks_pass": False}
    try:
        big = int(npr * 1.1)
        build = make_build_keys(nb, 0x5EED0001, dev)
        probe = make_probe_keys(big, nb, 0x5EED0002, dev)
        bcol = Column(build)
        import random
        rnd = random.Random(6)
        sizes = [int(npr * (0.9 + 0.2 * rnd.random())) for _ in range(20)]
        times, good = [], True
        for sz in sizes:
            pcol = Column(probe[:sz])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = join(pcol, bcol)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
            good = good and n == sz
        per_row = [t / sz * npr for t, sz in zip(times, sizes)]
        out["shape_sweep"] = {"op": "20 C3 joins, probe rows drawn from 0.9e9 ... 1.1e9 (a different size every call), in call order",
                              "mean_ms": sum(times) / len(times), "mean_ms_scaled_to_1e9_rows": sum(per_row) / len(per_row),
                              "ratio_to_headline": (sum(per_row) / len(per_row)) / headline_ms if headline_ms else None,
                              "max_ms": max(times), "calls_ms": [round(t, 3) for t in times], "probe_rows": sizes, "checks_pass": bool(good)}
        del probe, build
        torch.cuda.empty_cache()
    except Exception as e:                         # noqa: BLE001
        out["shape_sweep"] = {"error": f"{type(e).__name__}: {e}", "checks_pass": False}
    return out


def extra_ops(gdf, dev, n=1_000_000_000, reps=3):
    """SURVEY 8d's micro-metrics for the other operators north_star names, at 1e9 int64 rows through the C ABI (as tools/bench_ops.py):
    hash partition P = 256, inclusive prefix sum, compare + stencil compaction at 10 % selectivity.  {ms, frac of 8 TB/s on the
    algorithmic bytes, checks_pass (a size-independent property of each result)}."""
    import numpy as np
    import torch
    from libgdf_amd.columns import Column
    out = {}
    keys = make_probe_keys(n, 10000, 0x5EED0003, dev)
    vals = make_probe_keys(n, 1000, 0x5EED0004, dev)
    kc, vc = Column(keys), Column(vals)

    def timed(fn):
        r = fn()
        del r
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
            if _ < reps - 1:
                del r
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, r
    dt, res = timed(lambda: gdf.api.prefixsum(vc, True))
    scan = res
    ok = int(scan[-1].item()) == int(vals.sum().item()) and int(scan[0].item()) == int(vals[0].item())
    out["prefixsum_i64"] = {"ms": dt * 1e3, "frac": 16.0 * n / dt / 8e12, "algorithmic_bytes": 16.0 * n, "checks_pass": bool(ok)}
    del res, scan
    thr = 899
    kept = int((vals > thr).sum().item())

    def filt():
        st = gdf.api.comparison(vc, np.int64(thr), 4)
        return gdf.api.apply_stencil(vc, st)
    dt, res = timed(filt)
    got = res.data[: res.size]
    ok = int(got.numel()) == kept and bool((got > thr).all().item()) and int(got.sum().item()) == int(vals[vals > thr].sum().item())
    fb = 8.0 * n + 2.0 * n + 8.0 * n + 8.0 * kept
    out["compare_stencil_10pct"] = {"ms": dt * 1e3, "frac": fb / dt / 8e12, "algorithmic_bytes": fb, "kept": kept, "checks_pass": bool(ok)}
    del res, got
    torch.cuda.empty_cache()
    dt, res = timed(lambda: gdf.api.hash_partition([kc, vc], [0], 256))
    cols, offs = res
    pk, pv = cols[0].data, cols[1].data
    offs_t = torch.as_tensor(np.asarray(offs, dtype=np.int64), device=dev)
    ok = int(pk.numel()) == n and int(pk.sum().item()) == int(keys.sum().item()) and int(pv.sum().item()) == int(vals.sum().item()) and \
        int(offs_t.numel()) == 256 and bool((offs_t[1:] >= offs_t[:-1]).all().item())
    pb = (16.0 + 16.0 + 8.0) * n
    out["hash_partition_p256"] = {"ms": dt * 1e3, "frac": pb / dt / 8e12, "algorithmic_bytes": pb, "checks_pass": bool(ok)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=6,
                    help="untimed steps, each timed on its own and reported as warmup_calls_ms / first_call_ms.  The join compares physical "
                         "placements of its scratch and output blocks on calibration runs inside its first calls -- up to 16 / 8 / 6 candidates "
                         "per block, most searches settle after four to six -- under a per-call time budget (6 ms in a process's first searching "
                         "call, 24 ms after it: csrc/internal.h PlaceRound), so the searches are over after two to four calls; the default covers that")
    ap.add_argument("--place-draws", type=int, default=None,
                    help="A/B switch: challengers the pool draws per placed scratch block (library default 4; 0 = never re-draw)")
    ap.add_argument("--probe-rows", type=int, default=1_000_000_000)
    ap.add_argument("--build-rows", type=int, default=None)
    ap.add_argument("--cpu-sample", type=int, default=20_000_000, help="probe rows of the oracle-port CPU baseline sample (0 = skip)")
    ap.add_argument("--pandas-sample", type=int, default=100_000_000, help="probe rows of the pandas.merge CPU baseline (0 = skip)")
    ap.add_argument("--strategy", choices=["exchange", "auto", "fused", "shuffle", "broadcast"], default="exchange",
                    help="multi-GPU join.  exchange (default): the key-partitioned all-to-all join BASELINE.json names, at EVERY world size -- "
                         "the fused variant (rank split in the join's level-1 regroup, 4-byte keys in fixed-size blocks) and, where the shape does "
                         "not fit it, the plain key shuffle -- so that a 1 -> 8 GPU curve measures one algorithm; what the planner "
                         "(libgdf_amd.multigpu.choose_join_strategy: broadcast at 2 GPUs, shuffle at 4, fused at 8) would have run instead is timed "
                         "after it and reported in the extra field planner_choice.  auto: the planner's choice as `value`.  fused / shuffle / "
                         "broadcast: that strategy")
    ap.add_argument("--extra", type=int, default=1,
                    help="1 (default, single GPU only): after the headline's timed region, also measure BASELINE configs C2 and C5 through the C ABI "
                         "and append them as `extra.c2` / `extra.c5` (ms, roofline fraction, checks_pass); 0 = skip")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the multi-GPU (C4) code path even at world size 1 (1-rank RCCL group): measures its local passes")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run, exactly
        # the command line the driver would have used; rank 0 of the children prints the JSON line
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.device_count() > local_rank, f"rank {rank} needs cuda:{local_rank}, this node has {torch.cuda.device_count()} GPU(s)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or args.force_distributed
    if distributed:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            # a lost peer must end the run, not hang it: RCCL's watchdog aborts a collective stuck for 5 minutes
            import datetime
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=5))

    import libgdf_amd as gdf
    from libgdf_amd._binding import rmmOptions_t
    from libgdf_amd.columns import Column
    # pool mode: the multi-GB partition buffers are recycled between steps instead of hipMalloc'ed
    opts = rmmOptions_t(1, 0, False)
    gdf.librmm.rmmInitialize(C.byref(opts))
    if args.place_draws is not None:
        gdf._binding._rmm_cdll.gdf_amd_rmm_place_draws(C.c_int(args.place_draws))

    npr = args.probe_rows
    nb = args.build_rows if args.build_rows is not None else (npr // 8 if distributed else npr // 10)
    key_space = nb * world
    lib = gdf._binding._gdf_cdll

    if not distributed:
        build = make_build_keys(nb, 0x5EED0001, dev)
        probe = make_probe_keys(npr, key_space, 0x5EED0002, dev)
        pcol, bcol = Column(probe), Column(build)

        from libgdf_amd import gdf_column, libgdf, new_context
        from libgdf_amd.columns import column_array
        ctx = new_context()                       # {0, GDF_HASH, 0, 0, 0}
        la, ra = column_array([pcol]), column_array([bcol])
        on = (C.c_int * 1)(0)

        def step():
            # exactly one C-ABI call; the library-allocated index columns are released, not copied
            li, ri = gdf_column(), gdf_column()
            libgdf.gdf_inner_join(la, 1, on, ra, 1, on, 1, 0, None, C.byref(li), C.byref(ri), C.byref(ctx))
            n = int(li.size)
            libgdf.gdf_column_free(C.byref(li))
            libgdf.gdf_column_free(C.byref(ri))
            return n
        workload = f"C3 gdf_inner_join HASH: {npr} probe x {nb} build int64 rows, unique build keys, 100% hit, indices only"
    else:
        from libgdf_amd import multigpu
        # rank r owns build keys r, r+world, ... of a global permutation-free key space: key = perm_r * world + r
        build = make_build_keys(nb, 0x5EED0001 + rank, dev) * world + rank
        probe = make_probe_keys(npr, key_space, 0x5EED0002, dev, offset=rank * npr)

        last_pairs = [None]                        # the preflight (only) looks at the pairs of the step it just ran
        keep_pairs = [False]

        def done(pairs):
            if keep_pairs[0]:
                last_pairs[0] = pairs
            return pairs.numel()

        def step_broadcast():
            return done(multigpu.broadcast_inner_join(probe, build))

        def step_fused():
            pairs = multigpu.fused_inner_join(probe, build)
            if pairs is None:                      # None on EVERY rank: the shape did not fit the fixed-size blocks
                raise RuntimeError("fused_inner_join declined this shape")
            return done(pairs)

        def step_shuffle():
            return done(multigpu.distributed_inner_join(probe, build))

        def sampled_pairs_join_equal_keys(pairs, k=1 << 20):
            """About k of this rank's pairs, resolved to (owner rank << 40 | row) on both sides (a collective for the fused
            join: every rank calls this), must name rows with EQUAL keys.  The keys are functions of (rank, row) -- the
            generators above -- so any rank can recompute them without another exchange."""
            pg, bg = pairs.sample_global_ids(k)
            if pg.numel() == 0:
                return None
            pr, prow = pg >> 40, pg & ((1 << 40) - 1)
            br, brow = bg >> 40, bg & ((1 << 40) - 1)
            if int(prow.max()) >= npr or int(brow.max()) >= nb or int(pr.max()) >= world or int(br.max()) >= world:
                return "a sampled pair names a row that does not exist"
            pk = ((splitmix64_torch(prow + pr * npr + 0x5EED0002) >> 1) & 0x7FFFFFFFFFFFFFFF) % key_space
            bk = torch.empty_like(pk)
            for r in range(world):                 # rank r's build keys: its seeded permutation, regenerated here
                sel = br == r
                if bool(sel.any()):
                    bk[sel] = make_build_keys(nb, 0x5EED0001 + r, dev)[brow[sel]] * world + r
            bad = int((pk != bk).sum())
            return None if bad == 0 else f"{bad} of {pg.numel()} sampled pairs join UNEQUAL keys"

        steps = {"fused": (step_fused, f"C4 partitioned hash join: {npr} probe + {nb} build int64 rows per GPU, key space {key_space}, rank split "
                                       f"fused into the join's level-1 regroup at the sender, RCCL exchange of 4-byte keys in fixed-size blocks, "
                                       f"level 2 + LDS probe at the receiver"),
                 "shuffle": (step_shuffle, f"C4 partitioned hash join: {npr} probe + {nb} build int64 rows per GPU, key space {key_space}, "
                                           f"RCCL all-to-all shuffle + local gdf_inner_join"),
                 "broadcast": (step_broadcast, f"C4 rows, broadcast variant: {npr} probe + {nb} build int64 rows per GPU, key space {key_space}, "
                                               f"RCCL all-gather of the build keys + local gdf_inner_join")}
        planner = multigpu.choose_join_strategy(world, npr, nb)
        planned = {"exchange": "fused", "auto": planner}.get(args.strategy, args.strategy)
        # Preflight (untimed, before the warmup): one step of the planned strategy, checked against what this workload must
        # produce -- every probe key is a build key on exactly one rank, so the ranks' pair counts add up to the probe rows --
        # and a sample of 2^20 pairs per rank, resolved to global row ids, must join equal keys.
        # A strategy that raises or miscounts ON ANY RANK is dropped by all of them together and the next one is tried;
        # the JSON line says which ran and why.  (A failure inside a collective can still take the job down: then RCCL's
        # watchdog ends it.)
        order = [planned] + [k for k in ("fused", "shuffle", "broadcast") if k != planned]
        if args.strategy == "exchange":
            order = ["fused", "shuffle"]           # never the broadcast: the curve is the all-to-all join at every N
        preflight = []
        strategy = None
        for cand in order:
            err = None
            got = 0
            try:
                keep_pairs[0] = True
                got = steps[cand][0]()
                keep_pairs[0] = False
                err = sampled_pairs_join_equal_keys(last_pairs[0])
            except Exception as e:                 # noqa: BLE001 -- reported, not swallowed: see "preflight" in the output
                err = f"{type(e).__name__}: {e}"
            keep_pairs[0] = False
            last_pairs[0] = None
            flags = torch.tensor([1.0 if err else 0.0, float(got)], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(flags)
            bad, total = int(flags[0].item()), int(flags[1].item())
            if not bad and total != npr * world:
                err = f"pair count {total} != {npr * world} probe rows"
                bad = 1
            preflight.append({"strategy": cand, "ok": not bad, "error": err if err else (f"failed on {bad} other rank(s)" if bad else None)})
            if not bad:
                strategy = cand
                break
        if strategy is None:
            raise SystemExit("bench.py: no multi-GPU join strategy passed its preflight: " + json.dumps(preflight))
        step, workload = steps[strategy]

    def sync():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    out_rows = 0
    # the untimed calls, each under its own wall clock: the FIRST call of a process pays the cold allocations, the kernels' first
    # launches and the first candidates of the placement searches (the headline is a steady-state number; this is
    # what the call in front of it cost)
    warm_ms = []
    for _ in range(args.warmup):
        sync()
        w0 = time.perf_counter()
        out_rows = step()
        sync()
        warm_ms.append((time.perf_counter() - w0) * 1e3)
    if distributed:
        multigpu.reset_stats()
    lib.gdf_amd_profile_reset()
    lib.gdf_amd_profile_enable(1)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_rows = step()
    sync()
    dt = time.perf_counter() - t0
    lib.gdf_amd_profile_enable(0)
    prof_sides = read_profile(gdf, split_sides=True)
    prof = {}
    for k_, v_ in prof_sides.items():
        a_, b_ = prof.get(k_.split("@")[0], (0.0, 0))
        prof[k_.split("@")[0]] = (a_ + v_[0], b_ + v_[1])

    # what the planner would have run at this world size, when that is another strategy than the one `value` reports
    planner_choice = None
    if distributed and args.strategy == "exchange" and planner != strategy:
        pstep = steps[planner][0]
        err = None
        try:
            pstep()
            sync()
            p0 = time.perf_counter()
            for _ in range(args.steps):
                pstep()
            sync()
            pdt = time.perf_counter() - p0
        except Exception as e:                     # noqa: BLE001 -- reported in the JSON line
            err, pdt = f"{type(e).__name__}: {e}", 0.0
        pt = torch.tensor([pdt, 1.0 if err else 0.0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(pt, op=dist.ReduceOp.MAX)
        planner_choice = {"strategy": planner, "ms_per_step": float(pt[0].item()) / args.steps * 1e3 if not pt[1].item() else None,
                          "value": npr * world * args.steps / float(pt[0].item()) if (pt[0].item() > 0 and not pt[1].item()) else None,
                          "error": err}

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    rows = torch.tensor([float(out_rows)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rows, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3

    def kernel_roofline():
        """this rank's dominant kernel (by total HIP-event time inside the timed region) priced against the HBM peak"""
        roofline = None
        if prof:
            name, (tot_ms, launches) = max(prof.items(), key=lambda kv: kv[1][0])
            per_launch_ms = tot_ms / max(launches, 1)
            launches_per_step = launches / args.steps
            # jk_* kernels run once per relation per step: the per-step byte count is split over those launches
            fn = KERNEL_BYTES.get(name)
            if fn is not None:
                tb = 8.0 if key_space < 2 ** 32 else 12.0
                kb = 8.0 if not distributed else (4.0 if key_space < 2 ** 31 - 1 else 8.0)
                step_bytes = fn(float(npr), float(nb), tb, launches_per_step, kb)
                bytes_per_launch = step_bytes / launches_per_step
                achieved = bytes_per_launch / (per_launch_ms * 1e-3) / 1e9
                traffic, src = (pmc_traffic(name, launches_per_step) if (not distributed and npr == 1_000_000_000) else (None, None))
                roofline = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                            "algorithmic_bytes_per_launch": bytes_per_launch,
                            "avg_launch_ms": per_launch_ms, "launches_per_step": launches_per_step}
        return roofline

    comm_ranks = None
    if distributed:
        try:
            tr = multigpu.transport_for(None)
            n_comm, r_comm = tr.communicator_ranks() if hasattr(tr, "communicator_ranks") else (None, None)
            seen = torch.tensor([n_comm if n_comm is not None else -1, -(n_comm if n_comm is not None else -1),
                                 1 if r_comm == rank else 0], dtype=torch.int64, device=dev)
            if world > 1:
                dist.all_reduce(seen, op=dist.ReduceOp.MIN)
            # min(n) == max(n) on all ranks and every rank's communicator rank is its RANK: one number; else what was seen
            comm_ranks = int(seen[0]) if (int(seen[0]) == -int(seen[1]) and int(seen[2]) == 1) else {"min": int(seen[0]), "max": -int(seen[1]), "ranks_match": bool(int(seen[2]))}
        except Exception as e:                     # noqa: BLE001 -- reported in the line
            comm_ranks = f"{type(e).__name__}: {e}"
    mine = {"rank": rank, "roofline": kernel_roofline(),
            "kernels_ms_per_step": {k: v[0] / args.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}
    if distributed:
        mine["exchange_bytes_sent_per_step"] = multigpu.STATS["bytes_sent"] / args.steps
        mine["exchange_messages_per_step"] = multigpu.STATS["messages"] / args.steps
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    line = None
    if rank == 0:
        total_probe = npr * world
        value = total_probe * args.steps / dt
        kernel_roof = mine["roofline"]
        out_per_gpu = rows.item() / world
        e2e_bytes = 8.0 * npr + 8.0 * nb + 8.0 * out_per_gpu
        e2e = e2e_bytes / (ms_per_step * 1e-3) / 1e9
        # the WHOLE call against the HBM peak on its algorithmic bytes (8 N_p + 8 N_b + 8 N_out; partition passes are
        # overhead with zero algorithmic bytes), the probe phase on 16 B per probe row over the probe-side launches (tagged @probe by
        # the library: live HIP events), and the PMC traffic of one whole join.  The per-kernel contract object of rounds 1-4
        # (dominant kernel on its own bytes) is `roofline_kernel`.
        probe_ms = sum(v[0] for k_, v in prof_sides.items() if k_.endswith("@probe")) / args.steps
        probe_bytes = 8.0 * npr + 8.0 * out_per_gpu
        join_traffic, join_src = (pmc_traffic(None, 1.0) if (not distributed and npr == 1_000_000_000) else (None, None))
        roofline = {"bound": "hbm", "kernel": "whole gdf_inner_join call (all launches of one step)", "achieved": e2e, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": e2e / HBM_PEAK_GBS, "traffic": join_traffic, "traffic_source": join_src,
                    "algorithmic_bytes_per_launch": e2e_bytes, "avg_launch_ms": ms_per_step, "launches_per_step": 1.0,
                    "probe_phase": ({"achieved": probe_bytes / (probe_ms * 1e-3) / 1e9, "frac": probe_bytes / (probe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "algorithmic_bytes": probe_bytes, "probe_side_kernels_ms": probe_ms,
                                     "kernels_ms": {k_: v[0] / args.steps for k_, v in sorted(prof_sides.items()) if k_.endswith("@probe")}}
                                    if probe_ms > 0 else None)}
        result = {
            "metric": "hash-join probe rows/s", "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": workload, "probe_rows_per_gpu": npr, "build_rows_per_gpu": nb,
                       "out_rows": int(rows.item()), "parallelism": f"key-partitioned x{world}"},
            "first_call_ms": warm_ms[0] if warm_ms else None,
            "warmup_calls_ms": [round(x, 3) for x in warm_ms],
            "roofline": roofline,
            "roofline_kernel": kernel_roof,
            "kernels_ms_per_step": mine["kernels_ms_per_step"],
        }
        if not distributed:
            stats = (C.c_ulonglong * 4)()
            gdf._binding._rmm_cdll.gdf_amd_rmm_place_stats(stats)
            result["placement"] = {"challengers_drawn": int(stats[0]), "promoted": int(stats[1]), "entries": int(stats[2]),
                                   "still_exploring": int(stats[3])}
        if distributed:
            # the exchange, per GPU and step: what went to RCCL, and what the busiest xGMI link would need for it at the rate the
            # planner assumes (a rank reaches each peer over ONE link) -- an estimate, the links cannot be timed from in here
            sent = max(r["exchange_bytes_sent_per_step"] for r in per_rank)
            result["config"]["strategy"] = strategy
            # how many ranks the collectives really spanned: the library's own RCCL communicator (the transport the fused join talks
            # through) and torch.distributed's group, each as seen from rank 0 and agreed on by all ranks
            result["config"]["nranks"] = {"world_size_env": world, "torch_distributed": dist.get_world_size(), "rccl_communicator": comm_ranks}
            result["config"]["strategy_planned"] = planned
            result["config"]["planner_would_pick"] = planner
            if planner_choice is not None:
                result["planner_choice"] = planner_choice
            result["config"]["preflight"] = preflight
            result["exchange"] = {"bytes_sent_per_gpu_per_step": sent, "messages_per_gpu_per_step": max(r["exchange_messages_per_step"] for r in per_rank),
                                  "busiest_link_ms_at_assumed_rate": sent / max(world - 1, 1) / multigpu.XGMI_LINK_BYTES_PER_S * 1e3,
                                  "assumed_link_GBps": multigpu.XGMI_LINK_BYTES_PER_S / 1e9,
                                  "planner_estimate_ms": {k: v * 1e3 for k, v in multigpu.estimate_join_seconds(world, npr, nb).items()}}
            result["per_rank"] = per_rank
        if world == 1 and not distributed and args.extra and npr == 1_000_000_000:
            # the other two single-GPU BASELINE configs, measured AFTER the headline's timed region with its inputs released
            # (a driver-run line, not only profiles/, carries them)
            step = pcol = bcol = la = ra = probe = build = None      # (the step closure holds the columns)
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            result["extra"] = extra_configs(gdf, dev)
            try:
                result["extra"]["shapes"] = extra_shapes(gdf, dev, ms_per_step)
            except Exception as e:                 # noqa: BLE001 -- an extra, never fatal for the headline line
                result["extra"]["shapes"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.pandas_sample > 0:
            result["cpu_baseline"] = cpu_baseline_pandas(args.pandas_sample, max(args.pandas_sample // 10, 1))
        if world == 1 and args.cpu_sample > 0:
            port = cpu_baseline(args.cpu_sample, max(args.cpu_sample // 10, 1))
            result["cpu_baseline_oracle_port"] = port
            result.setdefault("cpu_baseline", port)
            try:
                result["cpu_baseline_all_cores"] = cpu_baseline_all_cores(100_000_000, 10_000_000)
            except Exception as e:                 # noqa: BLE001 -- an extra, never fatal for the headline line
                result["cpu_baseline_all_cores"] = {"error": f"{type(e).__name__}: {e}"}
        line = json.dumps(result)
    if distributed:
        # RCCL's version banner (NCCL_DEBUG=VERSION on the GPU boxes) sits in C stdio's buffer since communicator creation and
        # would otherwise come out AFTER the JSON line at exit: push it out now, on every rank, then print
        import ctypes
        ctypes.CDLL(None).fflush(None)
        dist.barrier()
    if rank == 0:
        print(line, flush=True)
    if distributed:
        multigpu.close_transports()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
