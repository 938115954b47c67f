"""Sharding of independent reads / read pairs over the GPUs of a node, and the whole-job reduction
of per-rank results.  Mirrors the only parallel axis of the reference (multiprocessing.Pool over
files / pairs, decode.py:158-162, pair_decode.py:292-297): items are independent, so there is no
data-path collective; torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used for the
barrier, the timing reduce and the optional gather of results only."""
import os

__all__ = ["env_rank_world", "shard_range", "shard_seeds", "shard_by_cost", "job_aggregate", "gather_strings",
           "gather_in_order"]


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_items, rank, world):
    """Contiguous, near-equal split of n_items: rank r owns [lo, hi).  Disjoint and covering."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_seeds(per_rank, rank):
    """Weak scaling: every rank decodes `per_rank` items of its own; global item id = seed."""
    return range(rank * per_rank, (rank + 1) * per_rank)


def shard_by_cost(costs, world):
    """Greedy longest-processing-time split of items with unequal cost (frames U + V of a pair before its envelope
    exists, envelope cells after): items sorted by decreasing cost, each to the least loaded rank.  Returns one
    index list per rank, each in decreasing cost (the persistent kernels pull work in index order, so a rank's long
    pairs start first and its tail is made of short ones).  Deterministic: ties keep input order."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += costs[i]
    return shards


def job_aggregate(dist, elapsed_s, units, device=None):
    """(max elapsed over ranks, per-unit sums over ranks).  dist may be None (single process)."""
    if dist is None or not dist.is_initialized():
        return elapsed_s, list(units)
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor(list(units), dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t[0]), [float(x) for x in u]


def gather_strings(dist, local_strings, dst=0):
    """Host-side gather of variable-length results in global item order (what the reference's
    Pool callback does with FASTA records, pair_decode.py:272-283)."""
    if dist is None or not dist.is_initialized():
        return list(local_strings)
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(list(local_strings), out, dst=dst)
    if out is None:
        return None
    return [s for part in out for s in part]


def gather_in_order(dist, local_items, local_indices, n_items, dst=0):
    """Gather results of a shard_by_cost split back into INPUT order on rank dst (None elsewhere)."""
    if dist is None or not dist.is_initialized():
        out = [None] * n_items
        for i, x in zip(local_indices, local_items):
            out[i] = x
        return out
    parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object((list(local_indices), list(local_items)), parts, dst=dst)
    if parts is None:
        return None
    out = [None] * n_items
    for idx, items in parts:
        for i, x in zip(idx, items):
            out[i] = x
    return out
