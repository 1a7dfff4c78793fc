"""-m gpu: the multi-GPU join with its REAL device functions at world sizes 2 and 3 -- on one GPU.

The boxes the tests run on have one GPU and RCCL refuses two ranks on one device, so the ranks are processes that share
cuda:0 and talk through gloo; the three communication calls of libgdf_amd/multigpu.py are staged through host memory for
that.  Everything else is the production path: gdf_amd_shuffle_partition_stable at fan-out 2 / 3, the bitmap exchange,
the prepared build side, the accumulated probe slices, global row ids from the received bitmaps."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stage_through_host():
    """gloo moves host tensors: wrap the collectives multigpu.py uses so that device tensors take a detour."""
    from libgdf_amd import multigpu
    a2a, allred = dist.all_to_all_single, dist.all_reduce

    def all_to_all_single(out, inp, group=None):
        o = torch.empty(out.shape, dtype=out.dtype)
        a2a(o, inp.cpu(), group=group)
        out.copy_(o)

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None):
        c = t.cpu()
        allred(c, op=op, group=group)
        t.copy_(c)

    def all_to_all_v(recv, send, recv_split, send_split, group, async_op):
        world, me = dist.get_world_size(group), dist.get_rank(group)
        hs, hr = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        ops, so, ro = [], 0, 0
        for r in range(world):
            ns, nr = int(send_split[r]), int(recv_split[r])
            if r == me:
                hr[ro:ro + nr].copy_(hs[so:so + ns])
            else:
                if ns:
                    ops.append(dist.P2POp(dist.isend, hs[so:so + ns], r, group))
                if nr:
                    ops.append(dist.P2POp(dist.irecv, hr[ro:ro + nr], r, group))
            so += ns
            ro += nr
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        recv.copy_(hr)
        return []

    dist.all_to_all_single, dist.all_reduce, multigpu._all_to_all_v = all_to_all_single, all_reduce, all_to_all_v


def _shards(world, big):
    rs = np.random.RandomState(77)
    npr, nb = (6_000_000, 1_200_000) if big else (30_000, 4_000)
    space = nb * world * 5 // 4
    builds = [rs.permutation(space)[: nb + 13 * r].astype(np.int64) * world + r for r in range(world)]      # disjoint key sets per rank
    probes = [rs.randint(0, space * world, size=npr + 101 * r).astype(np.int64) for r in range(world)]
    return probes, builds


def _worker(rank, world, port, big, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from libgdf_amd import multigpu
    _stage_through_host()
    probes, builds = _shards(world, big)
    p = torch.from_numpy(probes[rank]).cuda()
    b = torch.from_numpy(builds[rank]).cuda()
    pairs = multigpu.distributed_inner_join(p, b)
    pg, bg = pairs.global_ids()
    bpairs = multigpu.broadcast_inner_join(p, b)
    bpg, bbg = bpairs.global_ids()
    # group-by-sum of (key % 1000, key): local pre-aggregation, exchange of the partial sums, final aggregation
    gk, gv = multigpu.distributed_group_by_sum(p % 1000, p)
    others = {}
    for op in ("min", "max", "count", "avg"):
        ok, ov = multigpu.distributed_group_by(op, p % 1000, p)
        others[op] = (ok.cpu().numpy(), ov.cpu().numpy())
    q.put((rank, len(pairs.probe_pos), pg.cpu().numpy(), bg.cpu().numpy(), bpg.cpu().numpy(), bbg.cpu().numpy(),
           gk.cpu().numpy(), gv.cpu().numpy(), others))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,big", [(2, False), (3, False), (2, True)])
def test_device_path_at_world_sizes_2_and_3(world, big):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, big, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    probes, builds = _shards(world, big)
    # expected pairs in global ids: (rank << 40 | row) of every probe row whose key some rank's build relation holds
    allb = np.concatenate(builds)
    bid = np.concatenate([(r << 40) + np.arange(len(bk), dtype=np.int64) for r, bk in enumerate(builds)])
    order = np.argsort(allb)
    sb, sid = allb[order], bid[order]
    exp = []
    for r, pk in enumerate(probes):
        at = np.searchsorted(sb, pk)
        at[at == len(sb)] = 0
        rows = np.flatnonzero(sb[at] == pk)
        exp.append(np.stack([(r << 40) + rows, sid[at[rows]]], axis=1))
    exp = np.concatenate(exp)
    exp = exp[np.lexsort(exp.T[::-1])]
    for a, b in ((2, 3), (4, 5)):
        got = np.concatenate([np.stack([res[a], res[b]], axis=1) for res in results])
        got = got[np.lexsort(got.T[::-1])]
        np.testing.assert_array_equal(got, exp)
    if big:
        assert all(res[1] == 1 for res in results)          # the received slices were accumulated and probed once
    allp = np.concatenate(probes)
    sums = np.zeros(1000, dtype=np.int64)
    np.add.at(sums, allp % 1000, allp)
    ek = np.unique(allp % 1000)
    ev = sums[ek]
    gk = np.concatenate([res[6] for res in results])
    gv = np.concatenate([res[7] for res in results])
    order = np.argsort(gk)
    np.testing.assert_array_equal(gk[order], ek)             # every group on exactly one rank
    np.testing.assert_array_equal(gv[order], ev)
    import pandas as pd
    ref = pd.DataFrame({"k": allp % 1000, "v": allp}).groupby("k")["v"]
    for op, exp_col in (("min", ref.min()), ("max", ref.max()), ("count", ref.count()), ("avg", ref.mean())):
        k = np.concatenate([res[8][op][0] for res in results])
        v = np.concatenate([res[8][op][1] for res in results])
        o = np.argsort(k)
        np.testing.assert_array_equal(k[o], exp_col.index.values)
        if op == "avg":
            np.testing.assert_allclose(v[o], exp_col.values, rtol=1e-12)
        else:
            np.testing.assert_array_equal(v[o], exp_col.values)
