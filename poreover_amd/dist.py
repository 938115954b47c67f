"""Sharding of independent reads / read pairs over the GPUs of a node, and the whole-job reduction
of per-rank results.  Mirrors the only parallel axis of the reference (multiprocessing.Pool over
files / pairs, decode.py:158-162, pair_decode.py:292-297): items are independent, so there is no
data-path collective; torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used for the
barrier, the timing reduce and the optional gather of results only."""
import os
import subprocess
import sys

__all__ = ["env_rank_world", "shard_range", "shard_seeds", "shard_by_cost", "job_aggregate", "gather_strings",
           "gather_in_order", "visible_devices", "plan_devices", "run_sharded", "decode_distributed"]


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
    if dist.get_backend() == "gloo":
        device = None   # (host tensors: a timing figure and a few counts)
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


# ------------------------------------------------------------------------------------------------
# The decode drivers' fan-out over the GPUs of a node.  Two launch styles, one plan (shard_by_cost):
#   * `torchrun --nproc-per-node N -m poreover_amd pair-decode ...` (or bench.py): WORLD_SIZE ranks exist already;
#     rank r decodes shard r on device LOCAL_RANK, results are gathered on rank 0 (decode_distributed);
#   * a plain `python -m poreover_amd pair-decode ...` on a node with several GPUs: the driver spawns one worker
#     process per device (run_sharded), which is what the reference's multiprocessing.Pool does with CPU cores
#     (pair_decode.py:292-297, decode.py:158-162).
# The parent of spawned workers must not hold a HIP context (a process that has initialised the GPU may not
# fork / exec on this platform), so devices are COUNTED in a child process too.

def visible_devices():
    """Indices of the HIP devices this process may use: POREOVER_DEVICES="0,2,3" if set, else every visible
    device — counted by a short-lived child process so that the caller stays free of GPU state."""
    env = os.environ.get("POREOVER_DEVICES")
    if env is not None:
        return [int(x) for x in env.split(",") if x.strip() != ""]
    code = ("import sys; sys.path.insert(0, %r); from poreover_amd import _lib; "
            "print(_lib.load(require_gpu=False).po_device_count())" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
        return list(range(int(out.stdout.strip().splitlines()[-1])))
    except Exception:
        return []


def plan_devices(n_items, devices=None, max_workers=None):
    """The devices a job of n_items independent items is spread over: at most one worker per item, at most
    max_workers (the drivers pass --threads when it is > 1)."""
    if devices is None:
        devices = visible_devices() if n_items > 1 else [0]
    devices = list(devices)
    if max_workers is not None and max_workers > 0:
        devices = devices[:max_workers]
    return devices[:max(1, n_items)] if devices else []


def _shard_worker(rank, device, fn, items, extra):
    """Body of one spawned worker: bind the process to its device, decode its shard."""
    os.environ["POREOVER_DEVICE"] = str(device)
    if device is not None and device >= 0:
        from . import _lib
        _lib.set_device(int(device))
    return fn(items, extra)


def run_sharded(items, costs, fn, devices, extra=None, bind_device=True):
    """Decode `items` on several devices: greedy longest-first split by `costs`, one SPAWNED process per entry of
    `devices` (an index may repeat: two workers then share that GPU), fn(list_of_items, extra) -> list of results
    in each, results returned in INPUT order.  fn and extra must be picklable (module-level function)."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    world = len(devices)
    shards = shard_by_cost(costs, world)
    out = [None] * len(items)
    with ProcessPoolExecutor(max_workers=world, mp_context=mp.get_context("spawn")) as pool:
        futs = [pool.submit(_shard_worker, r, devices[r] if bind_device else None, fn, [items[i] for i in shards[r]], extra)
                for r in range(world) if shards[r]]
        live = [r for r in range(world) if shards[r]]
        for r, fu in zip(live, futs):
            res = fu.result()
            if len(res) != len(shards[r]):
                raise RuntimeError("worker %d returned %d results for %d items" % (r, len(res), len(shards[r])))
            for i, x in zip(shards[r], res):
                out[i] = x
    return out


def decode_distributed(items, costs, fn, extra=None, dst=0, backend="gloo"):
    """One process per GPU exists already (torchrun): every rank computes the same plan, decodes its own shard with
    fn(list_of_items, extra) and the records are gathered, in input order, on rank dst (None elsewhere).  The gather
    moves Python objects (strings), so it runs over gloo whatever the compute devices are; no data-path collective."""
    import torch.distributed as dist
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
    shards = shard_by_cost(costs, world)
    mine = shards[rank]
    res = fn([items[i] for i in mine], extra) if mine else []
    if len(res) != len(mine):
        raise RuntimeError("rank %d decoded %d of %d items" % (rank, len(res), len(mine)))
    return gather_in_order(dist if world > 1 else None, res, mine, len(items), dst=dst)
