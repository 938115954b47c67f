"""Sharding of independent reads / read pairs over the GPUs of a node, and the whole-job reduction
of per-rank results.  Mirrors the only parallel axis of the reference (multiprocessing.Pool over
files / pairs, decode.py:158-162, pair_decode.py:292-297): items are independent, so there is no
data-path collective; torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used for the
barrier, the timing reduce and the optional gather of results only."""
import os

__all__ = ["env_rank_world", "shard_range", "shard_seeds", "job_aggregate", "gather_strings"]


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
