"""Key-partitioned multi-GPU join / group-by: one process per GPU, RCCL all-to-all over xGMI.

The reference is single-GPU (no NCCL/MPI anywhere), so this layer has no
counterpart there; it sits ABOVE the unchanged per-GPU ``gdf_*`` C ABI:

  0. 8-byte keys whose global build-side range fits 31 bits travel as 4 bytes (the ``gdf_amd_narrow_keys`` image;
     one all-reduce of the build min / max decides);
  1. every rank hash-partitions each relation on the join key into ``world`` partitions, placed exactly as the
     public ``gdf_hash_partition`` places them (Murmur3 & (P-1) / % P) -- a different hash from the one the local
     join partitions on, so rank placement and local partitioning are uncorrelated.  The payload that travels
     with the keys of a (sender, receiver) pair is ONE BITMAP of the sender's rows: the partition is stable, so the
     j-th key of the segment is the j-th set bit -- 1 bit instead of a 4-byte row number per row on the links; the
     owner rank is implied by the segment.
     Narrowing and partitioning are ONE pass pair over the raw keys (``gdf_amd_shuffle_partition_stable``);
  2. the ``world x world`` send-count matrix is exchanged (one tiny all-to-all);
  3. one ``all_to_all_single`` per column moves partition r to rank r (xGMI is point-to-point, an
     all-to-all drives all 7 links of a GPU at once, so each column goes out as ONE large collective);
  4. every rank joins what it received: the received build relation is partitioned once
     (``gdf_amd_join_build_create``), every received probe slice is partitioned as it arrives into ONE set of fine
     partitions (``gdf_amd_join_probe_add``) and the lot is probed once at the end; the pairs are those of
     ``gdf_inner_join``.  (Shapes the accumulator declines are probed slice by slice.)  The result is a :class:`ShardedPairs`:
     index pairs into the RECEIVED tables plus what is needed to name the original rows
     (``(owner rank, local row)``), resolved lazily -- an 8-byte global id per output row would double the
     output traffic of the timed path;
  5. the probe relation goes through steps 1-4 in slices, software-pipelined: the all-to-all of slice c runs
     on RCCL's stream while slice c+1 is partitioned and slice c-1 is joined on the library's stream.

``shuffle_fn`` / ``prepare_fn`` / ``join_fn`` (and ``partition_fn`` of the group-by) are injectable so the exchange
logic is testable on CPU with the gloo backend (tests/test_multigpu_gloo.py): there they are numpy oracle functions,
here the C ABI.

Not done yet: fusing the rank split into the join's own radix partitioning (the
receiver would continue from 8-byte packed tuples instead of re-reading raw keys).
"""
from __future__ import annotations

from ._binding import GDFError, GDF_COLUMN_SIZE_TOO_BIG, GDF_UNSUPPORTED_METHOD

# what makes the layer change its PLAN (slice-by-slice joins, or the key shuffle instead of the fused blocks) rather than fail:
# the library declining a shape, and its 31-bit position space running out for an accumulated relation -- anything else (a HIP
# error, out of memory) is a fault and is re-raised
_PLAN_CHANGE = (GDF_UNSUPPORTED_METHOD, GDF_COLUMN_SIZE_TOO_BIG)

# bytes this rank handed to RCCL (remote segments only; a rank's own segment is a device copy) since the last reset --
# bench.py reports them per step next to the link-bound estimate
STATS = {"bytes_sent": 0, "messages": 0}


def reset_stats():
    STATS["bytes_sent"] = 0
    STATS["messages"] = 0


def _device_partition(keys, payload, world):
    """gdf_hash_partition over (key, payload) on the key column -> (keys_out, payload_out, offsets list)."""
    from . import api
    from .columns import Column
    outs, offsets = api.hash_partition([Column(keys), Column(payload)], [0], world)
    return outs[0].data, outs[1].data, offsets


def _device_shuffle(keys, row_base, world, narrow):
    """gdf_amd_shuffle_partition_stable: (keys [narrowed to int32 when narrow=(lo, hi)] partitioned on the key as
    gdf_hash_partition would place them, each partition in input order; the bitmaps that say which rows each partition
    took; offsets list).  Beyond 64 ranks: gdf_amd_shuffle_partition with a row-number column."""
    from . import api
    from .columns import Column
    if world <= 64:
        return api.shuffle_partition_stable(Column(keys), world, narrow=narrow)
    return api.shuffle_partition(Column(keys), world, row_base=row_base, narrow=narrow)


def _device_prepare(build_keys):
    """The received build relation, partitioned once (gdf_amd_join_build_create)."""
    from . import api
    from .columns import Column
    return api.JoinBuild([Column(build_keys)])


def _device_inner_join(probe_keys, build):
    from .columns import Column
    # the pairs stay in the library's buffers until somebody asks for them (ShardedPairs.global_ids): copying 8 B per
    # output row into torch tensors would add a quarter to the local HBM traffic of a slice join
    return build.probe([Column(probe_keys)], how="inner", copy=False)


def _device_narrow(keys, lo, hi):
    """gdf_amd_narrow_keys: int64 keys -> int32 (key - lo), -1 outside [lo, hi]."""
    import torch
    from ._binding import libgdf
    from .columns import Column
    out = torch.empty(keys.numel(), dtype=torch.int32, device=keys.device)
    cin, cout = Column(keys), Column(out)                              # keep the structs alive across the call
    libgdf.gdf_amd_narrow_keys(cin.ptr, int(lo), int(hi), cout.ptr)
    return out


def _narrow_range(probe_keys, build_keys, group):
    """(lo, hi) when 8-byte keys can travel and join as 4-byte ones (a third less to ship and re-read): the GLOBAL
    build-side range fits 31 bits.  Probe keys outside it cannot match any build key and become -1.  Else None."""
    import torch
    import torch.distributed as dist
    if probe_keys.dtype != torch.int64 or build_keys.dtype != torch.int64:
        return None
    big = torch.iinfo(torch.int64).max
    if build_keys.numel():
        lo, hi = torch.aminmax(build_keys)
        mm = torch.stack([lo, -hi])
    else:
        mm = torch.tensor([big, big], dtype=torch.int64, device=build_keys.device)
    dist.all_reduce(mm, op=dist.ReduceOp.MIN, group=group)            # [global min, -(global max)]
    lo, hi = int(mm[0]), -int(mm[1])
    if lo > hi or hi - lo >= (1 << 31) - 1:
        return None
    return lo, hi


class Received:
    """One relation after the exchange: keys and the segment bounds (rows ``bounds[r]:bounds[r+1]`` came from rank r),
    plus what names the senders' local rows: either their row numbers (``rows``, 4 bytes per row on the links) or, from a
    STABLE partition, one bitmap per sender (``bitmaps[r]``: bit i set iff row ``row_base[r] + i`` of rank r came here;
    the j-th key of segment r is the j-th set bit -- 1 bit per row on the links)."""

    def __init__(self, keys, rows, bounds, bitmaps=None, row_base=None):
        self.keys, self.rows, self.bounds, self.bitmaps, self.row_base = keys, rows, bounds, bitmaps, row_base
        self._rows_cache = {}

    def owner_of(self, positions):
        """Owner rank of received positions (int64 tensor)."""
        import torch
        b = torch.tensor(self.bounds[1:], dtype=torch.int64, device=positions.device)
        return torch.bucketize(positions.long(), b, right=True)

    def _rows_of(self, r):
        """Local row numbers of segment r in arrival order (= input order of the sender): the positions of the set bits."""
        import torch
        if r not in self._rows_cache:
            bits = self.bitmaps[r].contiguous().view(torch.uint8)                     # little-endian words: byte b holds rows 8b .. 8b+7
            shifts = torch.arange(8, dtype=torch.uint8, device=bits.device)
            ones = ((bits.unsqueeze(1) >> shifts) & 1).flatten()
            self._rows_cache[r] = torch.nonzero(ones).flatten() + int(self.row_base[r])
        return self._rows_cache[r]

    def global_ids(self, positions):
        """(owner rank << 40) | local row, as int64."""
        import torch
        pos = positions.long()
        owner = self.owner_of(pos)
        if self.bitmaps is None:
            return (owner << 40) | self.rows[pos].long()
        starts = torch.tensor(self.bounds, dtype=torch.int64, device=pos.device)
        within = pos - starts[owner]
        rows = torch.empty_like(pos)
        for r in range(len(self.bounds) - 1):
            sel = owner == r
            if bool(sel.any()):
                rows[sel] = self._rows_of(r)[within[sel]]
        return (owner << 40) | rows


class _ConcatReceived:
    """Several received slices seen as one relation: position p belongs to the slice whose range holds it."""

    def __init__(self, parts):
        self.parts = parts
        self.starts = [0]
        for r in parts:
            self.starts.append(self.starts[-1] + int(r.keys.numel()))

    def global_ids(self, positions):
        import torch
        pos = positions.long()
        starts = torch.tensor(self.starts, dtype=torch.int64, device=pos.device)
        which = torch.bucketize(pos, starts[1:], right=True)
        out = torch.empty_like(pos)
        for i, r in enumerate(self.parts):
            sel = which == i
            if bool(sel.any()):
                out[sel] = r.global_ids(pos[sel] - self.starts[i])
        return out


class ShardedPairs:
    """This rank's share of a distributed join, one entry per probe chunk: ``probe_pos[c][i]`` / ``build_pos[c][i]``
    index chunk c's received probe table / the received build table."""

    def __init__(self, probes, build, probe_pos, build_pos):
        self.probes, self.build, self.probe_pos, self.build_pos = probes, build, probe_pos, build_pos

    def numel(self):
        return sum(int(p.numel()) for p in self.probe_pos)

    def global_ids(self):
        import torch
        as_tensor = lambda x: x.tensor() if hasattr(x, "tensor") else x
        pg = [r.global_ids(as_tensor(p)) for r, p in zip(self.probes, self.probe_pos)]
        bg = [self.build.global_ids(as_tensor(b)) for b in self.build_pos]
        return torch.cat(pg), torch.cat(bg)

    def sample_global_ids(self, k):
        """global ids of about ``k`` evenly spaced pairs of this rank's share (all of them when it has fewer): what a caller
        needs to spot-check a billion-pair result without resolving every position (bench.py's preflight)."""
        import torch
        as_tensor = lambda x: x.tensor() if hasattr(x, "tensor") else x
        total = max(self.numel(), 1)
        pg, bg = [], []
        for r, p, b in zip(self.probes, self.probe_pos, self.build_pos):
            p, b = as_tensor(p), as_tensor(b)
            n = int(p.numel())
            if n == 0:
                continue
            take = min(n, max(1, int(k) * n // total))
            at = torch.div(torch.arange(take, dtype=torch.int64, device=p.device) * n, take, rounding_mode="floor")
            pg.append(r.global_ids(p[at]))
            bg.append(self.build.global_ids(b[at]))
        if not pg:
            empty = torch.empty(0, dtype=torch.int64)
            return empty, empty
        return torch.cat(pg), torch.cat(bg)


# Largest single message handed to RCCL.  Measured on this image (RCCL 2.26.6 inside torch 2.10, 1 rank sending to
# itself): a send/recv pair of 2.0e9 bytes or more delivers only its first half, 2^30 bytes are fine
# (tools/rccl_message_size_check.py).  Every message therefore travels in pieces of at most 2^29 bytes, and a rank's own
# segment never enters RCCL at all.
_MAX_MESSAGE_BYTES = 1 << 29


def _all_to_all_v(recv, send, recv_split, send_split, group, async_op):
    """Variable all-to-all of one column: the local segment is a device copy, every remote segment a sequence of
    bounded isend / irecv pairs issued as one batch.  Returns the list of in-flight works (empty if none)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    me = dist.get_rank(group)
    piece = max(1, _MAX_MESSAGE_BYTES // send.element_size())
    ops = []
    so = ro = 0
    for r in range(world):
        ns, nr = int(send_split[r]), int(recv_split[r])
        if r == me:
            recv[ro:ro + nr].copy_(send[so:so + ns])
        else:
            peer = dist.get_global_rank(group, r) if group is not None else r
            for a in range(0, ns, piece):
                ops.append(dist.P2POp(dist.isend, send[so + a:so + min(ns, a + piece)], peer, group))
                STATS["messages"] += 1
            STATS["bytes_sent"] += ns * send.element_size()
            for a in range(0, nr, piece):
                ops.append(dist.P2POp(dist.irecv, recv[ro + a:ro + min(nr, a + piece)], peer, group))
        so += ns
        ro += nr
    if not ops:
        return []
    works = dist.batch_isend_irecv(ops)
    if not async_op:
        for w in works:
            w.wait()
        return []
    return list(works)


class _Exchange:
    """One relation (or one chunk of it) on its way to its owner ranks: the partitioned send buffers, the receive
    buffers and the in-flight collectives.  ``finish()`` waits for them and returns the :class:`Received`.

    ``partitioned`` = (keys, payload, offsets).  A 1-D payload travels row by row next to its key (row numbers, group-by
    values).  A 2-D payload is the bitmap set of a STABLE partition (``payload[r]`` = the rows that go to rank r): rank r
    gets ``payload[r]`` whole, and ``row_base`` (the first row of this chunk in the sender's shard) with it."""

    def __init__(self, partitioned, group, async_op, row_base=0):
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(group)
        pk, pp, offsets = partitioned
        n = pk.numel()
        bounds = list(offsets) + [n]
        stable = pp.dim() == 2
        words = int(pp.shape[1]) if stable else 0
        # the count matrix, one row per rank: keys for you, and (stable) the length of my bitmap and my chunk's first row
        mine = [[bounds[r + 1] - bounds[r], words, int(row_base)] for r in range(world)]
        send_info = torch.tensor(mine, dtype=torch.int64, device=pk.device).flatten()
        recv_info = torch.empty_like(send_info)
        dist.all_to_all_single(recv_info, send_info, group=group)
        send_split = [m[0] for m in mine]
        info = recv_info.view(world, 3).tolist()
        recv_split = [int(i[0]) for i in info]
        total = int(sum(recv_split))
        self.keep = (pk, pp)                                                   # send buffers stay alive until finish()
        self.rk = torch.empty(total, dtype=pk.dtype, device=pk.device)
        self.works = _all_to_all_v(self.rk, pk, recv_split, send_split, group, async_op)
        self.bitmaps = self.row_base = self.rp = None
        if stable:
            rwords = [int(i[1]) for i in info]
            self.row_base = [int(i[2]) for i in info]
            flat = torch.empty(int(sum(rwords)), dtype=pp.dtype, device=pk.device)
            send_bits = pp.reshape(-1)                                         # a copy unless pp is contiguous
            self.keep = (pk, pp, send_bits)
            self.works += _all_to_all_v(flat, send_bits, rwords, [words] * world, group, async_op)
            self.bitmaps, at = [], 0
            for w in rwords:
                self.bitmaps.append(flat[at:at + w])
                at += w
        else:
            self.rp = torch.empty(total, dtype=pp.dtype, device=pk.device)
            self.works += _all_to_all_v(self.rp, pp, recv_split, send_split, group, async_op)
        self.bounds = [0]
        for c in recv_split:
            self.bounds.append(self.bounds[-1] + int(c))

    def finish(self):
        for w in self.works:
            if w is not None:
                w.wait()
        self.keep = None
        return Received(self.rk, self.rp, self.bounds, self.bitmaps, self.row_base)


def exchange_by_key(keys, payload, partition_fn=_device_partition, group=None):
    """Send every (key, payload) row to rank ``hash(key) mod world``.  Returns (keys, payload, bounds)."""
    import torch.distributed as dist
    r = _Exchange(partition_fn(keys, payload, dist.get_world_size(group)), group, async_op=False).finish()
    return r.keys, r.rows, r.bounds


def distributed_inner_join(probe_keys, build_keys, shuffle_fn=_device_shuffle, join_fn=_device_inner_join, group=None,
                           chunks=4, prepare_fn=_device_prepare):
    """Inner join of two row-sharded relations on one integer key column.

    Every rank passes its shard of both relations and gets back a :class:`ShardedPairs` with its share of
    the join; ``result.numel()`` summed over the ranks is the size of the global join.

    The probe relation travels in ``chunks`` slices: while slice c is on the links (asynchronous all-to-all on
    RCCL's own stream), slice c+1 is being hash-partitioned and slice c-1 joined against the received build
    relation on the library's stream, so that the exchange hides behind the local HBM passes instead of adding
    to them.

    ``shuffle_fn(keys, row_base, world, narrow)`` -> (partitioned keys, their row numbers or the bitmaps of a stable
    partition, offsets);
    ``prepare_fn(build_keys)`` -> whatever ``join_fn(probe_keys, prepared)`` takes as its build side (None: the keys).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = probe_keys.numel()
    narrow = _narrow_range(probe_keys, build_keys, group)
    # Every slice is a collective (count matrix + batched point-to-point), so ALL ranks must run the same number of
    # them whatever their own row count: the slice count follows the LARGEST shard, and a rank that runs out of rows
    # sends empty slices.
    sizes = torch.empty(world, dtype=torch.int64, device=probe_keys.device)
    dist.all_gather_into_tensor(sizes, torch.tensor([n], dtype=torch.int64, device=probe_keys.device), group=group)
    sizes = [int(x) for x in sizes.tolist()]
    n_total = sum(sizes)
    build_x = _Exchange(shuffle_fn(build_keys, 0, world, narrow), group, async_op=True, row_base=0)
    chunks = max(1, min(int(chunks), max(sizes)))
    step = (n + chunks - 1) // chunks
    build = prepared = acc = None
    probes, ppos, bpos = [], [], []

    def join_slice(r):
        """One received probe slice against the prepared build side: added to the accumulator (one probe pass at the
        very end) when the library can do that, joined on its own otherwise."""
        nonlocal acc
        probes.append(r)
        if acc is not None:
            try:
                acc.add([_as_column(r.keys)])
                return
            except GDFError as e:                   # a partition outgrew its room (skew): every slice on its own after all
                if e.errcode not in _PLAN_CHANGE:
                    raise                           # a real fault (HIP error, out of memory ...) is not a plan change
                acc = None
                for earlier in probes[:-1]:
                    li, ri = join_fn(earlier.keys, prepared)
                    ppos.append(li); bpos.append(ri)
        li, ri = join_fn(r.keys, prepared)
        ppos.append(li); bpos.append(ri)

    pending = None
    for c in range(chunks):
        lo, hi = min(n, c * step), min(n, (c + 1) * step)
        x = _Exchange(shuffle_fn(probe_keys[lo:hi], lo, world, narrow), group, async_op=True, row_base=lo)   # partition c, then start moving it
        if build is None:
            build = build_x.finish()
            prepared = prepare_fn(build.keys) if prepare_fn is not None else build.keys
            if hasattr(prepared, "accumulate"):
                # what this rank will receive: its share of all probe rows (the key hash spreads them evenly)
                try:
                    acc = prepared.accumulate(n_total // world + 1)
                except GDFError as e:
                    if e.errcode not in _PLAN_CHANGE:
                        raise
                    acc = None
        if pending is not None:                                                       # join c-1 while c moves
            join_slice(pending.finish())
        pending = x
    join_slice(pending.finish())
    if acc is not None:
        try:
            li, ri = acc.finish(copy=False)
            result = ShardedPairs([_ConcatReceived(probes)], build, [li], [ri])
        except GDFError as e:
            if e.errcode not in _PLAN_CHANGE:
                raise
            acc = None
            for r in probes:
                li, ri = join_fn(r.keys, prepared)
                ppos.append(li); bpos.append(ri)
            result = ShardedPairs(probes, build, ppos, bpos)
    else:
        result = ShardedPairs(probes, build, ppos, bpos)
    if hasattr(prepared, "close"):
        prepared.close()
    return result


def _as_column(t):
    from .columns import Column
    return Column(t)


class _LocalRows:
    """Rows that never left their rank (the probe relation of a broadcast join): position == local row."""

    def __init__(self, rank):
        self.rank = rank

    def global_ids(self, positions):
        return (positions.long() + (self.rank << 40))


class _GatheredRows:
    """A relation gathered from all ranks in rank order: position p belongs to the rank whose segment holds it and is
    that rank's local row p - start(segment)."""

    def __init__(self, keys, bounds):
        self.keys, self.bounds = keys, bounds

    def global_ids(self, positions):
        import torch
        pos = positions.long()
        starts = torch.tensor(self.bounds, dtype=torch.int64, device=pos.device)
        owner = torch.bucketize(pos, starts[1:], right=True)
        return (owner << 40) | (pos - starts[owner])


def _all_gather_v(local, group):
    """Every rank's 1-D tensor, concatenated in rank order -> (gathered tensor, bounds).  Bounded point-to-point
    messages like the all-to-all; the local piece is a device copy."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    me = dist.get_rank(group)
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    counts = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = [int(c) for c in counts.tolist()]
    bounds = [0]
    for c in counts:
        bounds.append(bounds[-1] + c)
    out = torch.empty(bounds[-1], dtype=local.dtype, device=local.device)
    limit = max(1, _MAX_MESSAGE_BYTES // local.element_size())
    ops = []
    for r in range(world):
        if r == me:
            out[bounds[r]:bounds[r + 1]].copy_(local)
            continue
        peer = dist.get_global_rank(group, r) if group is not None else r
        for a in range(0, local.numel(), limit):
            ops.append(dist.P2POp(dist.isend, local[a:min(local.numel(), a + limit)], peer, group))
            STATS["messages"] += 1
        STATS["bytes_sent"] += local.numel() * local.element_size()
        for a in range(0, counts[r], limit):
            ops.append(dist.P2POp(dist.irecv, out[bounds[r] + a:bounds[r] + min(counts[r], a + limit)], peer, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return out, bounds


def _device_join_columns(probe_keys, build_keys):
    from . import api
    from .columns import Column
    return api.join([Column(probe_keys)], [Column(build_keys)], how="inner", copy=False)


def broadcast_inner_join(probe_keys, build_keys, join_fn=_device_join_columns, group=None, narrow_fn=_device_narrow):
    """The same join without moving the probe relation: every rank gathers the whole build relation (keys only -- a
    gathered row's owner and local row number follow from its position) and joins its own probe shard against it.

    Bytes into a GPU: (world - 1) x its share of the build keys, against (world - 1) / world x (probe + build) x
    (key + row number) for the shuffle.  For C4 (build = probe / 8) that is 3.5 GB instead of 7.9 GB at 8 GPUs and
    0.5 GB instead of 4.5 GB at 2, where ONE xGMI link carries the whole exchange; the price is a local join against a
    ``world`` times larger build relation.  ``distributed_inner_join`` (the shuffle the north star names) stays the
    default of bench.py; this is the planner's other choice."""
    import torch.distributed as dist
    me = dist.get_rank(group)
    narrow = _narrow_range(probe_keys, build_keys, group)
    if narrow is not None and narrow_fn is not None:
        probe_keys, build_keys = narrow_fn(probe_keys, *narrow), narrow_fn(build_keys, *narrow)
    gathered, bounds = _all_gather_v(build_keys, group)
    li, ri = join_fn(probe_keys, gathered)
    return ShardedPairs([_LocalRows(me)], _GatheredRows(gathered, bounds), [li], [ri])


# ---- fused join: the sender runs the join's level-1 regroup -----------------------------------------------------------------
class FusedPairs:
    """This rank's share of a fused_inner_join: index pairs into its RECEIVE buffers (probe positions are numbered across
    the slices' buffers).  ``global_ids()`` turns them into (owner rank << 40) | row; it is a COLLECTIVE -- the row numbers
    stayed with the senders (they never travel in the timed path) and are fetched with one exchange per buffer."""

    def __init__(self, probe_layout, build_layout, group, probe_pos, build_pos, build_rows_buf, probe_rows_bufs, keep=None):
        self.probe_layout, self.build_layout, self.group = probe_layout, build_layout, group
        self.probe_pos, self.build_pos = probe_pos, build_pos
        self._build_rows, self._probe_rows = build_rows_buf, probe_rows_bufs
        self._keep = keep                                            # the build handle: the pairs index its receive buffer

    def numel(self):
        return int(self.probe_pos.numel())

    def _rows_of(self, rows_buf, layout):
        """the row numbers of the keys this rank received from every sender: the senders' blocks for this rank, sender-major"""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        blk = layout.block
        if hasattr(rows_buf, "materialize"):                         # api.FjRows: position -> row, inverted only now
            rows_buf = rows_buf.materialize()
        recv = torch.empty(world * blk, dtype=rows_buf.dtype, device=rows_buf.device)
        _all_to_all_v(recv, rows_buf[:world * blk], [blk] * world, [blk] * world, self.group, async_op=False)
        return recv

    def sample_global_ids(self, k):
        """global ids of about ``k`` evenly spaced pairs (see ShardedPairs.sample_global_ids); COLLECTIVE like global_ids()."""
        return self.global_ids(sample=int(k))

    def global_ids(self, sample=None):
        import torch
        as_tensor = lambda x: x.tensor() if hasattr(x, "tensor") else x
        build_pos, probe_pos = as_tensor(self.build_pos), as_tensor(self.probe_pos)
        if sample is not None and int(probe_pos.numel()) > sample > 0:
            n = int(probe_pos.numel())
            at = torch.div(torch.arange(sample, dtype=torch.int64, device=probe_pos.device) * n, sample, rounding_mode="floor")
            build_pos, probe_pos = build_pos[at], probe_pos[at]
        b = build_pos.long()
        brows = self._rows_of(self._build_rows, self.build_layout)
        bg = ((b // self.build_layout.block) << 40) | brows[b].long()
        blk = self.probe_layout.block
        per_buf = self.probe_layout.world * blk
        p = probe_pos.long()
        pg = torch.empty_like(p)
        which = p // per_buf
        for i, rows_buf in enumerate(self._probe_rows):            # every rank walks all slices: the exchange is collective
            prow = self._rows_of(rows_buf, self.probe_layout)
            sel = which == i
            if bool(sel.any()):
                q = p[sel] - i * per_buf
                pg[sel] = ((q // blk) << 40) | prow[q].long()
        return pg, bg


def _fj_send(keys, lo, hi, layout, row_base):
    from . import api
    from .columns import Column
    return api.fj_send(Column(keys), lo, hi, layout, row_base)


def _fj_build(recv_keys, recv_fill, lo, layout, expected_rows):
    from . import api
    return api.FjBuild(recv_keys, recv_fill, lo, layout, expected_rows)


def _fj_plan(world, build_total, rows_max, rows_per_key=1.0):
    from . import api
    return api.fj_plan(world, build_total, rows_max, rows_per_key)


_TRANSPORTS = {}            # process group -> the gdf_amd_transport the C entry point talks through


def transport_for(group=None):
    """The gdf_amd_transport of a torch.distributed group, made once: the library's own RCCL communicator when the group's backend
    is nccl (its unique id travels over the group's object broadcast), Python callbacks that stage the blocks through host memory
    for any other backend (gloo: ranks sharing one GPU in the tests)."""
    import torch.distributed as dist
    from . import api
    key = group if group is not None else "default"
    t = _TRANSPORTS.get(key)
    if t is None:
        if dist.get_backend(group) == "nccl":
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            box = [api.RcclTransport.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            t = api.RcclTransport(box[0], world, rank)
        else:
            # (LIBGDF_AMD_NO_A2AV: test switch, read HERE in the Python layer -- the transport's optional all_to_all_v stays NULL and
            # the C entry points take their equal-block exchange)
            import os
            t = api.CallbackTransport(group, with_all_to_all_v=not os.environ.get("LIBGDF_AMD_NO_A2AV"))
        _TRANSPORTS[key] = t
    return t


def close_transports():
    """destroy the cached transports (before the process group goes away)"""
    for t in _TRANSPORTS.values():
        t.close()
    _TRANSPORTS.clear()


import atexit as _atexit
_atexit.register(close_transports)      # an RCCL communicator must not outlive the HIP runtime's own teardown


def _fused_inner_join_c(probe_keys, build_keys, group, chunks):
    from . import api
    from .columns import Column
    tr = transport_for(group)
    got = api.dist_inner_join(Column(probe_keys), Column(build_keys), tr, chunks)
    if got is None:
        return None
    ppos, bpos, li, ri, info = got
    lay_p = api.FjLayout(info.world, info.fine_bits_p, info.coarse_bits_p, info.cap_p)
    lay_b = api.FjLayout(info.world, info.fine_bits_b, info.coarse_bits_b, info.cap_b)
    n, step = probe_keys.numel(), max(int(info.slice_rows), 1)
    brows = api.FjRows(bpos, 0, info.world * lay_b.block + api.FJ_DUMP_ELEMS)
    prows = []
    for c in range(info.chunks):
        a, b = min(n, c * step), min(n, (c + 1) * step)
        prows.append(api.FjRows(ppos[a:b], a, info.world * lay_p.block + api.FJ_DUMP_ELEMS))
    STATS["bytes_sent"] += 4 * (info.world - 1) * (lay_b.block + lay_b.regions_per_rank + info.chunks * (lay_p.block + lay_p.regions_per_rank))
    STATS["messages"] += 2 * (info.world - 1) * (1 + info.chunks)
    return FusedPairs(lay_p, lay_b, group, li, ri, brows, prows)


def fused_inner_join(probe_keys, build_keys, group=None, chunks=4, plan_fn=_fj_plan, send_fn=_fj_send, build_fn=_fj_build):
    """Inner join of two row-sharded relations on one integer key column with the rank split FUSED into the join's own
    partitioning (csrc/join.hip "FUSED multi-GPU join"): every rank regroups its rows by (owner rank, coarse partition on that
    rank) -- the join's level-1 pass, writing narrowed 4-byte keys -- ships each rank its block (4 B per row on the links, fixed
    block sizes: no count exchange before the data) and continues at level 2 on what it received.  Against the key shuffle
    (distributed_inner_join) a rank saves one pass over its rows on each side of the links.

    Returns a :class:`FusedPairs`, or ``None`` when the shape does not fit (keys that do not narrow to 32 bits, a world size /
    relation size outside gdf_amd_fj_plan's range, skewed keys that overflow the fixed-size regions) -- ON ALL RANKS, so that
    the caller can fall back to ``distributed_inner_join`` collectively."""
    import torch
    import torch.distributed as dist
    if plan_fn is _fj_plan and send_fn is _fj_send and build_fn is _fj_build and probe_keys.is_cuda:
        # the product path: ONE C call (gdf_amd_dist_inner_join, include/gdf/gdf_amd_ext.h) plans, regroups, exchanges and joins;
        # what follows below is the same protocol in Python, kept as its executable specification for the CPU tests
        # (tests/test_multigpu_gloo.py run it over gloo with numpy stand-ins for the three device steps)
        return _fused_inner_join_c(probe_keys, build_keys, group, chunks)
    world = dist.get_world_size(group)
    dev = probe_keys.device
    narrow = _narrow_range(probe_keys, build_keys, group)
    mine = torch.tensor([probe_keys.numel(), build_keys.numel()], dtype=torch.int64, device=dev)
    sizes = torch.empty(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes = sizes.view(world, 2).tolist()
    p_max, b_max = max(int(x[0]) for x in sizes), max(int(x[1]) for x in sizes)
    p_total, b_total = sum(int(x[0]) for x in sizes), sum(int(x[1]) for x in sizes)
    if narrow is None or b_total == 0 or p_total == 0:
        return None
    lo, hi = narrow
    chunks = max(1, min(int(chunks), p_max))
    step_max = (p_max + chunks - 1) // chunks
    lay_b = plan_fn(world, b_total, b_max, 1.0)
    lay_p = plan_fn(world, b_total, step_max, max(1.0, p_total / max(b_total, 1)))
    if lay_b is None or lay_p is None:
        return None
    # result positions are 31-bit and count the blocks' room and empty regions: checked here, from numbers every rank has,
    # before anything is exchanged (a rank that found out later raised alone and left its peers in a collective)
    if chunks * world * lay_p.block >= 2 ** 31 - 1 or world * lay_b.block >= 2 ** 31 - 1:
        return None

    def agree(flag):
        """True on every rank iff `flag` is False on all of them (one tiny all-reduce): a region overflow anywhere sends
        EVERY rank back to the shuffle"""
        t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return int(t.item()) == 0

    def exchange(keys_buf, fill, layout, async_op):
        blk, rpr = layout.block, layout.regions_per_rank
        rk = torch.empty(world * blk, dtype=keys_buf.dtype, device=dev)
        rf = torch.empty(world * rpr + 2, dtype=fill.dtype, device=dev)        # (+ 2: the library's flag / tile-count words are elsewhere; slack)
        works = _all_to_all_v(rk, keys_buf[:world * blk], [blk] * world, [blk] * world, group, async_op)
        works += _all_to_all_v(rf[:world * rpr], fill[:world * rpr], [rpr] * world, [rpr] * world, group, async_op)
        return rk, rf, works

    def wait(works):
        for w in works:
            if w is not None:
                w.wait()

    # ---- build relation ----
    bk, brows, bfill, over = send_fn(build_keys, lo, hi, lay_b, 0)
    if not agree(over):
        return None
    rbk, rbf, works = exchange(bk, bfill, lay_b, async_op=True)
    # ---- probe relation, sliced and software-pipelined as in distributed_inner_join ----
    n = probe_keys.numel()
    step = (n + chunks - 1) // chunks
    build = acc = None
    probe_rows, failed = [], False
    pending = None
    per_buf = world * lay_p.block

    def add(received, index):
        """queue one received probe buffer for level 2; False: the library declined (a plan change, settled by agree() below)"""
        if acc is None or failed:
            return True
        try:
            acc.add_recv(received[0], received[1], lay_p, index * per_buf)
            return True
        except GDFError as e:
            if e.errcode not in _PLAN_CHANGE:
                raise
            return False

    for c in range(chunks):
        a, b = min(n, c * step), min(n, (c + 1) * step)
        pk, prow, pfill, over = send_fn(probe_keys[a:b], lo, hi, lay_p, a)
        failed = failed or over
        probe_rows.append(prow)
        # the slice goes on the wire BEFORE anybody asks whether it overflowed (the agreement used to sit in front of the
        # first exchange -- a blocking all-reduce on the happy path of every join); an overflowed buffer is just useless
        x = exchange(pk, pfill, lay_p, async_op=True) + (pk, pfill)              # the send buffers stay alive with the works
        if c == 0 and not agree(over):
            # a region overflowed on some rank's FIRST slice (skewed probe keys are usually skewed everywhere): every rank leaves
            # now, before three more slices are regrouped, shipped and partitioned for nothing
            wait(works)
            wait(x[2])
            return None
        if build is None:
            wait(works)
            try:
                build = build_fn(rbk, rbf, lo, lay_b, b_total // world + 1)
                acc = build.accumulate(p_total // world + 1)
            except GDFError as e:
                if e.errcode not in _PLAN_CHANGE:
                    raise
                failed = True
        if pending is not None:
            wait(pending[2])
            failed = failed or not add(pending, c - 1)
        pending = x
    wait(pending[2])
    failed = failed or not add(pending, chunks - 1)
    li = ri = None
    if acc is not None and not failed:
        try:
            li, ri = acc.finish(copy=False)
        except GDFError as e:
            if e.errcode not in _PLAN_CHANGE:
                raise
            failed = True
    ok = agree(failed or li is None)
    if build is not None and not ok:
        build.close()
    return FusedPairs(lay_p, lay_b, group, li, ri, brows, probe_rows, keep=build) if ok else None


# ---- planner ------------------------------------------------------------------------------------------------------------
# Cost model of the two join strategies for one rank, in seconds.  Constants are measurements of this library on one
# MI355X (profiles/): the local passes per row, and what one xGMI link sustains in one direction
# (76.8 GB/s peak per link and direction; 60 GB/s assumed -- the links have never been measured from here, the 1-GPU
# boxes have none).  xGMI is point-to-point: with `world` GPUs a rank talks to each peer over ONE link, so an all-to-all
# of V bytes per rank puts V / world on every link, and the time is that of the busiest link, not of the aggregate.
XGMI_LINK_BYTES_PER_S = 60e9
_SHUFFLE_LOCAL_S_PER_ROW = 15.1e-12      # sender split + receiver partition + probe, per row of (probe + build): 17.0 ms at C4 shard sizes
_FUSED_LOCAL_S_PER_ROW = 11.3e-12        # sender level 1 + receiver level 2 + probe: 12.7 ms at C4 shard sizes (tools/sim_c4_fused.py)
_FUSED_BYTES_PER_ROW = 4.45              # 4-byte keys in fixed-size regions: + 6 sigma of room (11 % at ten probe rows per key)
_JOIN_S_PER_PROBE_ROW = 9.6e-12          # gdf_inner_join, NARROW keys: 10.6 ms for 1e9 x 1e8
_JOIN_S_PER_BUILD_ROW = 10e-12
_LEVEL3_BUILD_ROWS = 1.6e8               # larger build relations take a third partitioning level (csrc/join.hip refine_side):
_LEVEL3_S_PER_PROBE_ROW = 5e-12          # 1e9 x 2.5e8 / 5e8 / 1e9 rows: 17.2 / 20.6 / 29.1 ms
_LEVEL3_S_PER_BUILD_ROW = 4e-12
_NARROW_S_PER_ROW = 2.4e-12              # gdf_amd_narrow_keys over both relations (broadcast variant): 2.7 ms per 1.125e9 rows


def estimate_join_seconds(world, probe_rows, build_rows, key_bytes=4.125):
    """-> {"shuffle": s, "broadcast": s} for per-rank shard sizes `probe_rows` / `build_rows` (the larger of the exchange on
    the busiest link and the local passes, i.e. assuming they overlap)."""
    rows = probe_rows + build_rows
    shuffle_link = rows * key_bytes / max(world, 1) / XGMI_LINK_BYTES_PER_S if world > 1 else 0.0
    shuffle = max(shuffle_link, _SHUFFLE_LOCAL_S_PER_ROW * rows)
    fused_link = rows * _FUSED_BYTES_PER_ROW / max(world, 1) / XGMI_LINK_BYTES_PER_S if world > 1 else 0.0
    fused = max(fused_link, _FUSED_LOCAL_S_PER_ROW * rows)
    gathered = build_rows * world
    bcast_link = build_rows * 4.0 / XGMI_LINK_BYTES_PER_S if world > 1 else 0.0        # every peer's shard arrives over its own link
    local = _NARROW_S_PER_ROW * rows + _JOIN_S_PER_PROBE_ROW * probe_rows + _JOIN_S_PER_BUILD_ROW * gathered
    if gathered > _LEVEL3_BUILD_ROWS:
        local += _LEVEL3_S_PER_PROBE_ROW * probe_rows + _LEVEL3_S_PER_BUILD_ROW * gathered
    return {"shuffle": shuffle, "fused": fused, "broadcast": max(bcast_link, local)}


def choose_join_strategy(world, probe_rows, build_rows):
    """"fused" (sender-side level 1, 4-byte keys in fixed-size blocks), "shuffle" (stable key split + bitmaps, RCCL all-to-all)
    or "broadcast" (all-gather the build keys, probe rows stay home) -- whichever the cost model above expects to finish
    first.  With C4's shard sizes: broadcast at 2 GPUs (either exchange of the probe side would push > 2 GB through the one
    link between them), the shuffle at 4 (both exchanges are link-bound there and the shuffle sends exact sizes, the fused blocks
    carry 11 % of room), the fused exchange at 8 (local passes bound both: 12.7 against 17.0 ms)."""
    est = estimate_join_seconds(world, probe_rows, build_rows)
    moving = min(("fused", "shuffle"), key=lambda k: est[k])                  # (fused falls back to the shuffle when its shape checks fail)
    # moving the probe relation rests on the ASSUMED link rate; leaving it at home does not: the exchange has to win by 10 %
    return moving if est[moving] < 0.9 * est["broadcast"] else "broadcast"


def planned_inner_join(probe_keys, build_keys, group=None, shuffle_kw=None, broadcast_kw=None, fused_kw=None):
    """distributed_inner_join or broadcast_inner_join, chosen by choose_join_strategy from the GLOBAL shard sizes (one
    all-gather of two numbers, so that every rank takes the same branch)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mine = torch.tensor([probe_keys.numel(), build_keys.numel()], dtype=torch.int64, device=probe_keys.device)
    sizes = torch.empty(2 * world, dtype=torch.int64, device=probe_keys.device)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes = sizes.view(world, 2).tolist()
    p = max(int(x[0]) for x in sizes)
    b = max(int(x[1]) for x in sizes)
    strategy = choose_join_strategy(world, p, b)
    if strategy == "broadcast":
        return broadcast_inner_join(probe_keys, build_keys, group=group, **(broadcast_kw or {}))
    if strategy == "fused":
        pairs = fused_inner_join(probe_keys, build_keys, group=group, **(fused_kw or {}))
        if pairs is not None:                          # None on ALL ranks: the shape did not fit, take the shuffle together
            return pairs
    return distributed_inner_join(probe_keys, build_keys, group=group, **(shuffle_kw or {}))


def distributed_gather(ids, columns, valids=None, take_fn=None, group=None):
    """The multi-GPU face of the joins' ``result_cols`` step (reference, per rank: src/join/joining.cu:375-479 gathers the relations'
    columns by the index columns): ``ids`` -- int64 GLOBAL row ids as the distributed joins produce them ((owner rank << 40) | local row,
    -1 for the missing side of an unmatched row) -- name rows of a row-sharded relation; ``columns`` are THIS rank's shard of it
    (``valids``: a bool tensor or None per column).  Returns a list of (values, valid bools) per column, ``ids.numel()`` rows each: the
    named rows, fetched from the ranks that own them; null where the id is -1 or the source row is null.  COLLECTIVE.

    On the device this is ONE C call (gdf_amd_dist_gather, csrc/dist_ops.hip).  With ``take_fn`` given -- the CPU gloo tests' numpy
    stand-in -- the same request / response protocol runs here as its executable specification:
    ``take_fn(column, valid, rows)`` -> (values, valid flags) reads the local shard at ``rows`` (a row of -1 is a null)."""
    import torch
    import torch.distributed as dist
    valids = valids or [None] * len(columns)
    if take_fn is None and ids.is_cuda:
        from . import api
        from .columns import Column, mask_from_bools

        def col(t, ok):
            if ok is None:
                return Column(t)
            okn = ok.cpu().numpy().astype(bool)
            return Column(t, torch.from_numpy(mask_from_bools(okn)).to(t.device), null_count=int(len(okn) - okn.sum()))
        return api.dist_gather(ids, [col(c, v) for c, v in zip(columns, valids)], transport_for(group))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    # 1. every id becomes (owner, local row, position); a missing side is asked of the caller itself as row -1
    owner = torch.where(ids >= 0, ids >> 40, torch.full_like(ids, rank))
    row = torch.where(ids >= 0, ids & ((1 << 40) - 1), torch.full_like(ids, -1))
    assert bool(((owner >= 0) & (owner < world)).all()), "an id names no rank"
    pos = torch.argsort(owner, stable=True)              # 2. split by owner: the local rows travel, the positions stay
    asked = torch.bincount(owner, minlength=world)
    everyone = [None] * world
    dist.all_gather_object(everyone, [row[pos][owner[pos] == r].clone() for r in range(world)], group=group)
    req = [everyone[src][rank] for src in range(world)]
    # 3. the owner serves every sender in the order its requests arrived, and the answers go back the way they came
    answers = [[take_fn(c, v, q) for c, v in zip(columns, valids)] for q in req]
    dist.all_gather_object(everyone, answers, group=group)
    out = []
    for j, c in enumerate(columns):                      # 4. what came back lies in owner order, as the kept positions do
        vals = torch.cat([everyone[o][rank][j][0] for o in range(world)])
        flags = torch.cat([everyone[o][rank][j][1] for o in range(world)])
        assert vals.numel() == ids.numel() and all(everyone[o][rank][j][0].numel() == int(asked[o]) for o in range(world))
        v = torch.zeros(ids.numel(), dtype=c.dtype)
        f = torch.zeros(ids.numel(), dtype=torch.bool)
        v[pos] = vals
        f[pos] = flags
        out.append((v, f))
    return out


def distributed_group_by_multi(op, keys, values, key_valids=None, value_valid=None, group_fn=None, owner_fn=None, group=None):
    """``gdf_group_by_<op>`` of a row-sharded relation over SEVERAL key columns, validity masks honoured (BASELINE configuration C5 across
    ranks; reference shape: sqls_ops.cu:1085-1363, row hash of gdf_table.cuh:704-854).  ``keys``: list of tensors; ``key_valids``: list of
    bool tensors or None per key column; ``value_valid``: bool tensor or None.  A row with a null in any key column is dropped, a null
    value is skipped, a group without a valid value reports 0 and valid=False (COUNT: 0, valid).  Returns this rank's groups:
    (list of key tensors, aggregates, bool tensor of valid aggregates), in ascending key order.

    On the device this is ONE C call (gdf_amd_dist_group_by_multi, csrc/dist_ops.hip).  With ``group_fn`` / ``owner_fn`` given -- the CPU
    gloo tests' numpy stand-ins -- the same protocol runs here as its executable specification:
    ``group_fn(op, keys, values, key_valids, value_valid)`` -> (key tensors, aggregate, valid bools) is one local masked group-by with its
    result sorted by key; ``owner_fn(keys, world)`` -> the owner rank of every row (row hash % world)."""
    import torch
    import torch.distributed as dist
    if group_fn is None and owner_fn is None and values.is_cuda:
        from . import api
        from .columns import Column, mask_from_bools

        def col(t, ok):
            if ok is None:
                return Column(t)
            okn = ok.cpu().numpy().astype(bool)
            return Column(t, torch.from_numpy(mask_from_bools(okn)).to(t.device), null_count=int(len(okn) - okn.sum()))
        kv = key_valids or [None] * len(keys)
        return api.dist_group_by_multi(op, [col(k, v) for k, v in zip(keys, kv)], col(values, value_valid), transport_for(group))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sop = "sum" if op == "avg" else op
    vals = _widen(values) if op == "avg" else values
    # 1. two local masked group-bys over the same rows, both sorted by key: the partial aggregate S and the number of VALID values C
    gk, cnt, _ = group_fn("count", keys, values, key_valids, value_valid)
    s = None
    if op != "count":
        gk2, s, _ = group_fn(sop, keys, vals, key_valids, value_valid)
        assert all(torch.equal(a, b) for a, b in zip(gk, gk2))
    # 2. (key columns, S, C) travel to the owner of the row hash; no mask travels (S is 0 exactly where C is 0)
    owner = owner_fn(gk, world)
    cols = list(gk) + ([s] if s is not None else []) + [cnt]
    pieces = [[c[owner == r].clone() for c in cols] for r in range(world)]
    everyone = [None] * world
    dist.all_gather_object(everyone, pieces, group=group)
    mine = [torch.cat([everyone[src][rank][j] for src in range(world)]) for j in range(len(cols))]
    rk, rc = mine[: len(keys)], mine[-1]
    # 3. the owner combines: S over the partials that had a valid value (the same masked operator), C by a sum
    fk, fc, _ = group_fn("sum", rk, rc, None, None)
    if op == "count":
        return fk, fc, torch.ones(fc.numel(), dtype=torch.bool)
    fk2, fs, ok = group_fn(sop, rk, mine[len(keys)], None, rc > 0)
    assert all(torch.equal(a, b) for a, b in zip(fk, fk2))
    if op == "avg":
        fs = torch.where(fc > 0, fs.double() / fc.clamp(min=1).double(), torch.zeros_like(fs, dtype=torch.float64))
    return fk, fs, ok


def distributed_group_by_sum(keys, values, group_fn=None, partition_fn=_device_partition, group=None):
    """Group-by-sum of a row-sharded (key, value) relation: local pre-aggregation, exchange of the partial
    aggregates by key hash (far fewer rows than the input), final aggregation on the owner rank."""
    if group_fn is None and partition_fn is _device_partition and keys.is_cuda:
        return distributed_group_by("sum", keys, values, group=group)          # the C entry point
    if group_fn is None:
        def group_fn(k, v):
            from . import api
            from .columns import Column
            gk, ga = api.group_by("sum", [Column(k)], Column(v))
            return gk[0].clone(), ga.clone()
    k1, v1 = group_fn(keys, values)
    k2, v2, _ = exchange_by_key(k1, v1, partition_fn, group)
    return group_fn(k2, v2)


def _device_group(op, k, v, out_dtype=None):
    """One local gdf_group_by_<op>.  COUNT is typed by its OUTPUT column in the library (the reference's rule), so the
    partial counts are asked for as int64 -- in the value dtype an int8 column would wrap at 128 rows per group."""
    from . import api
    from ._binding import GDF_INT64
    from .columns import Column
    if op == "count" and out_dtype is None:
        out_dtype = GDF_INT64
    gk, ga = api.group_by(op, [Column(k)], Column(v), out_dtype=out_dtype)
    return gk[0].clone(), ga.clone()


def _widen(values):
    """The accumulator type of a distributed AVG: int64 for integer values, float64 for floating-point ones (what the
    single-GPU gdf_group_by_avg accumulates in) -- partial SUMs in a narrow value dtype would wrap before they meet."""
    import torch
    if values.dtype in (torch.float32, torch.float64):
        return values if values.dtype == torch.float64 else values.double()
    return values if values.dtype == torch.int64 else values.long()


def distributed_group_by(op, keys, values, group_fn=_device_group, partition_fn=_device_partition, group=None):
    """``gdf_group_by_<op>`` (sum, min, max, count, avg) of a row-sharded (key, value) relation on one key column:
    every rank pre-aggregates its shard, the partial aggregates travel to the rank ``hash(key) % world`` owns, and are
    combined there -- partial sums / minima / maxima by the same operator, partial counts by a sum, AVG as the quotient of
    the combined sums and counts (a float64 column).  Returns this rank's groups: (keys, aggregates)."""
    if group_fn is _device_group and partition_fn is _device_partition and keys.is_cuda:
        # the product path: ONE C call (gdf_amd_dist_group_by, include/gdf/gdf_amd_ext.h; csrc/dist_ops.hip) pre-aggregates,
        # partitions, exchanges over the group's gdf_amd_transport and combines; what follows is the same protocol in Python, kept
        # as its executable specification for the CPU tests (tests/test_multigpu_gloo.py: numpy stand-ins for the device steps)
        from . import api
        from .columns import Column
        return api.dist_group_by(op, Column(keys), Column(values), transport_for(group))
    if op in ("sum", "min", "max"):
        k1, v1 = group_fn(op, keys, values)
        k2, v2, _ = exchange_by_key(k1, v1, partition_fn, group)
        return group_fn(op, k2, v2)
    if op == "count":
        k1, c1 = group_fn("count", keys, values)
        k2, c2, _ = exchange_by_key(k1, c1, partition_fn, group)
        return group_fn("sum", k2, c2)
    if op == "avg":
        values = _widen(values)
        ks, ss = distributed_group_by("sum", keys, values, group_fn, partition_fn, group)
        kc, cc = distributed_group_by("count", keys, values, group_fn, partition_fn, group)
        os_, oc = ks.argsort(), kc.argsort()               # the two results name the same groups; align them by key
        return ks[os_], ss[os_].double() / cc[oc].double()
    raise ValueError(f"unknown aggregation {op!r}")
