"""The N > 1 path on CPU: two processes over gloo exercise the same sharding / reduction code that
bench.py and the drivers use with RCCL on GPUs."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["PO_REPO"])
import torch.distributed as dist
from poreover_amd import dist as podist
rank, local_rank, world = podist.env_rank_world()
dist.init_process_group("gloo")
lo, hi = podist.shard_range(11, rank, world)
mine = ["item%02d" % i for i in range(lo, hi)]                 # stands in for decoded sequences
tmax, (units, bases) = podist.job_aggregate(dist, 1.0 + rank, [hi - lo, 10 * (hi - lo)])
allstr = podist.gather_strings(dist, mine)
seeds = list(podist.shard_seeds(5, rank))
costs = [5, 1, 9, 3, 3, 7, 2, 8, 1, 1, 6]                      # unequal pairs: greedy LPT split, results back in input order
idx = podist.shard_by_cost(costs, world)[rank]
ordered = podist.gather_in_order(dist, ["r%d" % costs[i] for i in idx], idx, len(costs))
dist.barrier()
if rank == 0:
    print(json.dumps({"tmax": tmax, "units": units, "bases": bases, "all": allstr, "seeds0": seeds, "ordered": ordered}))
dist.destroy_process_group()
'''


def test_shard_range_properties():
    from poreover_amd.dist import shard_range
    for n in (0, 1, 7, 8, 10000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(180)
def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PO_REPO=REPO, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         env=env, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["tmax"] == 2.0                       # max over ranks of (1.0, 2.0)
    assert res["units"] == 11 and res["bases"] == 110
    assert res["all"] == ["item%02d" % i for i in range(11)]   # global order preserved
    assert res["seeds0"] == [0, 1, 2, 3, 4]
    assert res["ordered"] == ["r%d" % c for c in [5, 1, 9, 3, 3, 7, 2, 8, 1, 1, 6]]


def test_shard_by_cost_properties():
    import random
    from poreover_amd.dist import shard_by_cost
    rnd = random.Random(3)
    for world in (1, 2, 4, 8):
        costs = [rnd.randint(1, 9000) for _ in range(257)]
        shards = shard_by_cost(costs, world)
        assert sorted(i for sh in shards for i in sh) == list(range(len(costs)))      # disjoint and covering
        loads = [sum(costs[i] for i in sh) for sh in shards]
        assert max(loads) - min(loads) <= max(costs)                                   # the LPT bound
        for sh in shards:
            assert [costs[i] for i in sh] == sorted((costs[i] for i in sh), reverse=True)
