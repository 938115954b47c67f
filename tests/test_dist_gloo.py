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
dist.barrier()
if rank == 0:
    print(json.dumps({"tmax": tmax, "units": units, "bases": bases, "all": allstr, "seeds0": seeds}))
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
