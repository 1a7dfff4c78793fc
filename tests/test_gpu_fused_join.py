"""-m gpu: the FUSED multi-GPU join (gdf_amd_fj_*: the sender runs the join's level-1 regroup, the receiver continues at
level 2) on one GPU.

A rank's send buffer is `world` blocks, one per destination; a receive buffer is `world` blocks, one per sender, in the same
layout.  Feeding a sender's whole buffer back as the receive buffer therefore emulates `world` senders on one device: the
receiver does not look at the rank bits, so the result must be the plain join of the two relations -- for every world
size, with the probe relation arriving in slices, against the oracle."""
import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


def _fused_self_join(gdf, probe, build, world, slices):
    import torch
    from libgdf_amd import api
    from libgdf_amd.columns import Column
    lo, hi = int(build.min()), int(build.max())
    nb, npr = len(build), len(probe)
    step = (npr + slices - 1) // slices
    lay_b = api.fj_plan(world, nb * world, nb)               # (a rank's share of world x nb rows is nb: this "rank" receives all of it)
    lay_p = api.fj_plan(world, nb * world, step, max(1.0, npr / nb))
    assert lay_b is not None and lay_p is not None
    assert (world << lay_b.coarse_bits) <= 1024 and 1 <= lay_b.fine_bits - lay_b.coarse_bits <= 8
    tb = torch.from_numpy(build).cuda()
    tp = torch.from_numpy(probe).cuda()
    bk, brows, bfill, over = api.fj_send(Column(tb), lo, hi, lay_b, 0)
    assert not over
    kept = int(bfill[:lay_b.nregions].sum())
    assert kept == nb                                            # every build key is inside its own range
    b = api.FjBuild(bk, bfill, lo, lay_b, nb)
    acc = b.accumulate(npr)
    prows = []
    per_buf = world * lay_p.block
    for i in range(slices):
        a0, a1 = min(npr, i * step), min(npr, (i + 1) * step)
        pk, prow, pfill, over = api.fj_send(Column(tp[a0:a1]), lo, hi, lay_p, a0)
        assert not over
        acc.add_recv(pk, pfill, lay_p, i * per_buf)
        prows.append(prow)
    li, ri = acc.finish()
    li, ri = li.long(), ri.long()
    rows_b = brows.materialize()[ri].cpu().numpy().astype(np.int64)
    which = li // per_buf
    rows_p = np.empty(li.numel(), dtype=np.int64)
    for i in range(slices):
        sel = (which == i)
        if bool(sel.any()):
            rows_p[sel.cpu().numpy()] = prows[i].materialize()[li[sel] - i * per_buf].cpu().numpy()
    # the owner of a received key is the block it sits in: block index = mulhi(hash, world) of the key, checked on the build side
    assert int((ri // lay_b.block).max()) < world
    b.close()
    return rows_p, rows_b


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("dtype", [np.int64, np.int32])
def test_fused_join_equals_plain_join(gdf, world, dtype):
    rs = np.random.RandomState(100 + world)
    nb, npr = 60_000, 400_000
    base = (1 << 40) if dtype == np.int64 else -5000
    build = (rs.permutation(nb * 2)[:nb] + base).astype(dtype)                   # unique keys
    probe = (rs.randint(-1000, nb * 2 + 1000, size=npr) + base).astype(dtype)    # some outside the build range, some misses inside
    gp, gb = _fused_self_join(gdf, probe, build, world, slices=3)
    el, er = oracle.join([probe], [build], "inner")
    got = np.stack([gp, gb], axis=1)
    exp = np.stack([el, er], axis=1)
    np.testing.assert_array_equal(got[np.lexsort(got.T[::-1])], exp[np.lexsort(exp.T[::-1])])


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("base", [(1 << 32) - 30_000, (7 << 32) - 5, -20_000], ids=["across-2^32", "just-below-7*2^32", "across-zero"])
def test_fused_join_keys_whose_raw_values_straddle_a_2_32_boundary(gdf, world, base, force_path):
    """The sender (fj_scatter) and the receiver's level 2 hash key32 + lo without 64-bit arithmetic: key_fold's high word is lo's own or
    one more (the carry), two fold constants to choose from.  Build keys on both sides of a multiple of 2^32 (and of zero: lo negative)
    take both constants; against the oracle, and -- power-of-two worlds -- with the rank remap as a multiply instead of a shift
    (GDF_FJ_NO_POW2) the sender puts every row into the same (rank, coarse partition, XCD) region."""
    import torch
    from libgdf_amd import api
    from libgdf_amd.columns import Column
    rs = np.random.RandomState(world * 31 + (base & 0xff))
    nb, npr = 60_000, 300_000
    build = (rs.permutation(nb * 2)[:nb] + base).astype(np.int64)
    probe = (rs.randint(-500, nb * 2 + 500, size=npr) + base).astype(np.int64)
    gp, gb = _fused_self_join(gdf, probe, build, world, slices=2)
    el, er = oracle.join([probe], [build], "inner")
    got = np.stack([gp, gb], axis=1)
    exp = np.stack([el, er], axis=1)
    np.testing.assert_array_equal(got[np.lexsort(got.T[::-1])], exp[np.lexsort(exp.T[::-1])])
    if world & (world - 1) == 0:
        lo, hi = int(build.min()), int(build.max())
        lay = api.fj_plan(world, nb * world, npr, max(1.0, npr / nb))
        regions = []
        for off in (None, "1"):
            force_path("GDF_FJ_NO_POW2", off)
            _, rows, fill, over = api.fj_send(Column(torch.from_numpy(probe).cuda()), lo, hi, lay, 0)
            assert not over
            pos = rows.pos.cpu().numpy().astype(np.int64) & 0xffffffff
            regions.append(np.where(pos == 0xffffffff, -1, pos // lay.cap))
        np.testing.assert_array_equal(regions[0], regions[1])


@pytest.mark.parametrize("world", [1, 2, 3, 4, 6, 8])
@pytest.mark.parametrize("dtype", [np.int64, np.int32])
@pytest.mark.parametrize("keys", ["unique", "twice"])
def test_fused_join_on_six_byte_tuples(gdf, world, dtype, keys, force_path):
    """The receiver's level 2 at the headline geometry (2^15 fine partitions per rank, forced on a small relation): six-byte
    tuples of (hash remainder, position), csrc/join.hip p6_store.  The fed-back buffer holds the keys of EVERY rank, so the
    remainder has to keep what the rank remap shifts out of the hash (p6_low): an odd world is a bijection, a power of two
    stores the owner in the emptied low bits, world = 6 keeps eight-byte tuples.  Against the oracle and against the eight-byte
    path (GDF_JK_NO_P6); repeated build keys take the multimap kernels on the same tuples."""
    rs = np.random.RandomState(300 + world)
    nb, npr = 150_000, 1_200_000
    force_path("GDF_JK_FORCE_FB", "15")
    base = (5 << 34) + 99 if dtype == np.int64 else -70_000
    build = (rs.permutation(nb * 2)[:nb] + base).astype(dtype)
    if keys == "twice":
        build[: nb // 2] = build[nb // 2:]
    probe = (rs.randint(-1000, nb * 2 + 1000, size=npr) + base).astype(dtype)
    el, er = oracle.join([probe], [build], "inner")
    exp = np.stack([el, er], axis=1)
    exp = exp[np.lexsort(exp.T[::-1])]
    for eight_bytes in (False, True):
        force_path("GDF_JK_NO_P6", "1" if eight_bytes else None)
        gp, gb = _fused_self_join(gdf, probe, build, world, slices=2)
        got = np.stack([gp, gb], axis=1)
        np.testing.assert_array_equal(got[np.lexsort(got.T[::-1])], exp)
    force_path("GDF_JK_NO_P6", None)


def test_fused_join_duplicate_build_keys_and_large(gdf):
    rs = np.random.RandomState(7)
    nb, npr = 3_000_000, 9_000_000
    build = rs.randint(0, nb // 2, size=nb).astype(np.int64)                     # every key ~2 times: multimap units
    probe = rs.randint(0, nb // 2, size=npr).astype(np.int64)
    gp, gb = _fused_self_join(gdf, probe, build, 8, slices=4)
    assert bool((probe[gp] == build[gb]).all())
    mult = np.bincount(build, minlength=nb // 2)
    assert len(gp) == int(mult[probe].sum())
    assert len(np.unique(gp * nb + gb)) == len(gp)


def test_fused_plan_declines_what_it_cannot_take(gdf):
    from libgdf_amd import api
    assert api.fj_plan(64, 10**9, 10**8) is None                                 # 64 ranks x 128 coarse partitions > 1024 bins
    assert api.fj_plan(8, 4 * 10**9, 5 * 10**8) is None                          # a rank's share needs a third partitioning level
    lay = api.fj_plan(8, 10**9, 125 * 10**6)
    assert lay is not None and lay.fine_bits == 15 and lay.coarse_bits == 7 and lay.world * (1 << lay.coarse_bits) == 1024


@pytest.mark.parametrize("n", [1, 777, 32768, 32769, 200_001])
def test_send_buffer_and_positions_describe_the_same_rows(gdf, n):
    """gdf_amd_fj_send's contract: out_pos[i] is where row i's narrowed key went (0xffffffff for rows outside [lo, hi]); the
    fill counters add up to the rows sent; every region holds only keys of its (rank, coarse partition) bin -- checked through
    the receiver: joining the buffer against the same keys finds every sent row exactly once."""
    import torch
    from libgdf_amd import api
    from libgdf_amd.columns import Column
    rs = np.random.RandomState(n)
    keys = rs.permutation(4 * n + 10)[:n].astype(np.int64) + 10**12
    lo, hi = int(keys.min()) + (1 if n > 2 else 0), int(keys.max())           # (n > 2: the smallest key lies outside the range)
    world = 4
    lay = api.fj_plan(world, max(n * world, 1), n)
    assert lay is not None
    kb, rows, fill, over = api.fj_send(Column(torch.from_numpy(keys).cuda()), lo, hi, lay, 5)
    assert not over
    pos = rows.pos.cpu().numpy().astype(np.int64) & 0xffffffff
    sent = (keys >= lo) & (keys <= hi)
    assert np.array_equal(pos != 0xffffffff, sent)
    assert int(fill[:lay.nregions].sum()) == int(sent.sum())
    assert len(np.unique(pos[sent])) == int(sent.sum())                         # no two rows share a position
    buf = kb.cpu().numpy().astype(np.int64) & 0xffffffff
    np.testing.assert_array_equal(buf[pos[sent]] + lo, keys[sent])              # the position holds the row's narrowed key
    # positions lie inside the filled part of their region
    region, off = pos[sent] // lay.cap, pos[sent] % lay.cap
    assert bool((off < fill.cpu().numpy()[region]).all())
    inv = rows.materialize().cpu().numpy()
    np.testing.assert_array_equal(inv[pos[sent]], np.nonzero(sent)[0] + 5)      # row_base = 5
    assert int((inv >= 0).sum()) == int(sent.sum())


def test_send_reports_regions_that_overflow(gdf):
    """Skewed keys: one key value repeated far beyond a region's capacity sets *overflowed (the caller then takes the shuffle)."""
    import torch
    from libgdf_amd import api
    from libgdf_amd.columns import Column
    n = 1_000_000
    keys = np.full(n, 12345, dtype=np.int64)
    keys[::3] = np.arange(0, n, 3)
    lay = api.fj_plan(8, 8 * n, n)
    assert lay is not None and lay.cap < n // 2
    _, _, _, over = api.fj_send(Column(torch.from_numpy(keys).cuda()), 0, n, lay, 0)
    assert over


# ---- gdf_amd_dist_inner_join: the whole fused join behind ONE C call (include/gdf/gdf_amd_ext.h) ----
def _dist_join_one_rank(gdf, transport, probe, build, chunks):
    """-> sorted (probe row, build row) pairs of a world-1 distributed join, or None when the library declined"""
    import torch
    from libgdf_amd import api
    from libgdf_amd.columns import Column
    tp, tb = torch.from_numpy(probe).cuda(), torch.from_numpy(build).cuda()
    got = api.dist_inner_join(Column(tp), Column(tb), transport, chunks)
    if got is None:
        return None
    ppos, bpos, li, ri, info = got
    assert info.world == 1 and 1 <= info.chunks <= max(chunks, 1)
    assert info.block_p == (8 << info.coarse_bits_p) * info.cap_p and info.block_b == (8 << info.coarse_bits_b) * info.cap_b
    per_buf, step = info.world * info.block_p, max(int(info.slice_rows), 1)
    li, ri = li.tensor().long(), ri.tensor().long()
    # position -> local row: invert the senders' maps (world 1: this rank sent everything it received)
    brow = api.FjRows(bpos, 0, info.block_b + api.FJ_DUMP_ELEMS).materialize()
    rows_b = brow[ri].cpu().numpy().astype(np.int64)
    rows_p = np.empty(li.numel(), dtype=np.int64)
    which = (li // per_buf).cpu().numpy()
    n = len(probe)
    for c in range(info.chunks):
        a, b = min(n, c * step), min(n, (c + 1) * step)
        sel = which == c
        if sel.any():
            prow = api.FjRows(ppos[a:b], a, per_buf + api.FJ_DUMP_ELEMS).materialize()
            rows_p[sel] = prow[li[torch.from_numpy(sel).cuda()] - c * per_buf].cpu().numpy()
    return sort_pairs(rows_p, rows_b)


def sort_pairs(a, b):
    o = np.lexsort((b, a))
    return a[o], b[o]


@pytest.mark.parametrize("wire", ["rccl", "callbacks"])
@pytest.mark.parametrize("dtype", [np.int64, np.int32])
@pytest.mark.parametrize("chunks", [1, 4])
def test_dist_inner_join_c_entry_one_rank(gdf, wire, dtype, chunks):
    """gdf_amd_dist_inner_join at world 1 through both wires -- the library's own RCCL communicator (gdf_amd_rccl_transport_create:
    ncclCommInitRank on a 128-byte id, ncclSend / ncclRecv groups on its own stream) and a transport of host-staged callbacks --
    against the oracle: plan, sender regroup, exchange of equal blocks + fill counters, receiver level 2 + probe, agreement, all
    inside the C call.  A fifth of the probe keys miss; keys start at an offset (they travel as key - lo)."""
    import torch.distributed as dist
    from libgdf_amd import api
    rs = np.random.RandomState(chunks + len(wire))
    nb, npr = 40_000, 700_000
    build = (rs.permutation(nb * 5 // 4)[:nb] + 1_000_000).astype(dtype)
    probe = (rs.randint(0, nb * 5 // 4, size=npr) + 1_000_000).astype(dtype)
    if wire == "rccl":
        tr = api.RcclTransport(api.RcclTransport.unique_id(), 1, 0)
    else:
        import os
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        assert not dist.is_initialized()
        dist.init_process_group("gloo", rank=0, world_size=1)
        tr = api.CallbackTransport()
    try:
        got = _dist_join_one_rank(gdf, tr, probe, build, chunks)
        assert got is not None
        el, er = oracle.join([probe], [build], "inner")
        np.testing.assert_array_equal(got[0], el)
        np.testing.assert_array_equal(got[1], er)
    finally:
        tr.close()
        if wire != "rccl":
            dist.destroy_process_group()          # (other tests of this session make their own default group)


@pytest.mark.parametrize("case", ["keys-wider-than-31-bits", "one-key-holds-a-third", "empty-build", "empty-probe"])
def test_dist_inner_join_c_entry_declines_collectively(gdf, case):
    """Shapes the fused path cannot take come back as *declined = 1 (on every rank: the answer is an all-reduce) with no result
    columns, not as an error: keys whose global range does not narrow to 31 bits, skewed keys that overflow the fixed-size
    regions, an empty relation.  The caller then takes the key shuffle (libgdf_amd/multigpu.py)."""
    from libgdf_amd import api
    rs = np.random.RandomState(3)
    nb, npr = 50_000, 900_000
    build = rs.permutation(nb).astype(np.int64)
    probe = rs.randint(0, nb, size=npr).astype(np.int64)
    if case == "keys-wider-than-31-bits":
        build[0] = 1 << 40
    elif case == "one-key-holds-a-third":
        probe[rs.rand(npr) < 0.34] = 7
    elif case == "empty-build":
        build = build[:0]
    else:
        probe = probe[:0]
    tr = api.RcclTransport(api.RcclTransport.unique_id(), 1, 0)
    try:
        assert _dist_join_one_rank(gdf, tr, probe, build, 4) is None
    finally:
        tr.close()
