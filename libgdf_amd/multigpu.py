"""Key-partitioned multi-GPU join / group-by: one process per GPU, RCCL all-to-all over xGMI.

The reference is single-GPU (no NCCL/MPI anywhere), so this layer has no
counterpart there; it sits ABOVE the unchanged per-GPU ``gdf_*`` C ABI:

  1. every rank hash-partitions each relation on the join key into ``world`` partitions with the public
     ``gdf_hash_partition`` (Murmur3 & (P-1) / % P) -- a different hash from the mix64 the local join
     partitions on, so rank placement and local partitioning are uncorrelated.  The payload that travels
     with a key is its 4-byte LOCAL row number; the owner rank is implied by the segment it arrives in;
  2. the ``world x world`` send-count matrix is exchanged (one tiny all-to-all);
  3. one ``all_to_all_single`` per column moves partition r to rank r (xGMI is point-to-point, an
     all-to-all drives all 7 links of a GPU at once, so each column goes out as ONE large collective);
  4. every rank joins what it received with ``gdf_inner_join``.  The result is a :class:`ShardedPairs`:
     index pairs into the RECEIVED tables plus what is needed to name the original rows
     (``(owner rank, local row)``), resolved lazily -- an 8-byte global id per output row would double the
     output traffic of the timed path.

``partition_fn`` / ``join_fn`` are injectable so the exchange logic is testable on CPU with the gloo
backend (tests/test_multigpu_gloo.py): there they are numpy oracle functions, here the C ABI.

Not done yet: fusing the rank split into the join's own radix partitioning (the
receiver would continue from 8-byte packed tuples instead of re-reading raw keys) and pipelining the
exchange against it; by the volume arithmetic that is what >= 6x at 8 GPUs needs.
"""
from __future__ import annotations


def _device_partition(keys, payload, world):
    """gdf_hash_partition over (key, payload) on the key column -> (keys_out, payload_out, offsets list)."""
    from . import api
    from .columns import Column
    outs, offsets = api.hash_partition([Column(keys), Column(payload)], [0], world)
    return outs[0].data, outs[1].data, offsets


def _device_inner_join(probe_keys, build_keys):
    from . import api
    from .columns import Column
    return api.join([Column(probe_keys)], [Column(build_keys)], how="inner")


class Received:
    """One relation after the exchange: keys, the senders' local row numbers, and the segment bounds
    (rows ``bounds[r]:bounds[r+1]`` came from rank r)."""

    def __init__(self, keys, rows, bounds):
        self.keys, self.rows, self.bounds = keys, rows, bounds

    def owner_of(self, positions):
        """Owner rank of received positions (int64 tensor)."""
        import torch
        b = torch.tensor(self.bounds[1:], dtype=torch.int64, device=positions.device)
        return torch.bucketize(positions.long(), b, right=True)

    def global_ids(self, positions):
        """(owner rank << 40) | local row, as int64."""
        return (self.owner_of(positions) << 40) | self.rows[positions.long()].long()


class ShardedPairs:
    """This rank's share of a distributed join: ``probe_pos[i]`` / ``build_pos[i]`` index the received tables."""

    def __init__(self, probe, build, probe_pos, build_pos):
        self.probe, self.build, self.probe_pos, self.build_pos = probe, build, probe_pos, build_pos

    def numel(self):
        return self.probe_pos.numel()

    def global_ids(self):
        return self.probe.global_ids(self.probe_pos), self.build.global_ids(self.build_pos)


def exchange_by_key(keys, payload, partition_fn=_device_partition, group=None):
    """Send every (key, payload) row to rank ``hash(key) mod world``.  Returns (keys, payload, bounds)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    pk, pp, offsets = partition_fn(keys, payload, world)
    n = keys.numel()
    bounds = list(offsets) + [n]
    send_counts = torch.tensor([bounds[r + 1] - bounds[r] for r in range(world)], dtype=torch.int64, device=keys.device)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)          # the count matrix, one row per rank
    send_split = send_counts.tolist()
    recv_split = recv_counts.tolist()
    total = int(sum(recv_split))
    rk = torch.empty(total, dtype=keys.dtype, device=keys.device)
    rp = torch.empty(total, dtype=payload.dtype, device=keys.device)
    dist.all_to_all_single(rk, pk, recv_split, send_split, group=group)
    dist.all_to_all_single(rp, pp, recv_split, send_split, group=group)
    rb = [0]
    for c in recv_split:
        rb.append(rb[-1] + int(c))
    return rk, rp, rb


def distributed_inner_join(probe_keys, build_keys, partition_fn=_device_partition, join_fn=_device_inner_join, group=None):
    """Inner join of two row-sharded relations on one integer key column.

    Every rank passes its shard of both relations and gets back a :class:`ShardedPairs` with its share of
    the join.  ``len(result.probe_pos)`` summed over the ranks is the size of the global join.
    """
    import torch
    dev = probe_keys.device
    probe_rows = torch.arange(probe_keys.numel(), dtype=torch.int32, device=dev)
    build_rows = torch.arange(build_keys.numel(), dtype=torch.int32, device=dev)
    pk, prow, pb = exchange_by_key(probe_keys, probe_rows, partition_fn, group)
    bk, brow, bb = exchange_by_key(build_keys, build_rows, partition_fn, group)
    li, ri = join_fn(pk, bk)
    return ShardedPairs(Received(pk, prow, pb), Received(bk, brow, bb), li, ri)


def distributed_group_by_sum(keys, values, group_fn=None, partition_fn=_device_partition, group=None):
    """Group-by-sum of a row-sharded (key, value) relation: local pre-aggregation, exchange of the partial
    aggregates by key hash (far fewer rows than the input), final aggregation on the owner rank."""
    if group_fn is None:
        def group_fn(k, v):
            from . import api
            from .columns import Column
            gk, ga = api.group_by("sum", [Column(k)], Column(v))
            return gk[0].clone(), ga.clone()
    k1, v1 = group_fn(keys, values)
    k2, v2, _ = exchange_by_key(k1, v1, partition_fn, group)
    return group_fn(k2, v2)
