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


# ---------------------------------------------------------------------------------------------------
# the drivers' sharded decode path on REAL fixture pairs (the reference's own outputs are the expectation)
PAIR_WORKER = r'''
import os, sys, json, argparse, pickle
sys.path.insert(0, os.environ["PO_REPO"]); sys.path.insert(0, os.path.join(os.environ["PO_REPO"], "tests"))
from poreover_amd.decoding import pair_decode
from _cpu_pair_decode import tagged_decode
a = argparse.Namespace(**json.loads(os.environ["PO_ARGS"]))
pairs = json.loads(os.environ["PO_PAIRS"])
res = pair_decode.decode_pairs(pairs, a, decode_fn=tagged_decode)     # WORLD_SIZE = 2: this rank's shard, gathered on rank 0
if int(os.environ["RANK"]) == 0:
    with open(os.environ["PO_OUT"], "wb") as f:
        pickle.dump(res, f)
else:
    assert res is None
import torch.distributed as dist
dist.barrier(); dist.destroy_process_group()
'''


def _fixture_pairs(tmp_path, golden, golden_inputs):
    import numpy as np
    recs = [r for r in golden["pairs"] if r["kind"] == "poreover"]
    pairs = []
    for r in recs:
        for k in ("y1", "y2"):
            np.save(tmp_path / ("p%d_%s.npy" % (r["index"], k)), np.exp(golden_inputs["pair%d_%s" % (r["index"], k)]))
        pairs.append(["p%d_y1.npy" % r["index"], "p%d_y2.npy" % r["index"]])
    args = dict(dir=str(tmp_path), basecaller="poreover", reverse_complement=False, out=str(tmp_path / "o"), threads=1,
                method="envelope", single="viterbi", logging="info", debug=False, algorithm="beam", alignment="banded",
                beam_width=5, debug_envelope=False, diagonal_envelope=False, diagonal_width=50, padding=5,
                skip_matches=False, skip_threshold=10, beam_search_method="row_col", window=200)
    return recs, pairs, args


def _check_against_reference(recs, results):
    for r, x in zip(recs, results):
        run = r["runs"]["row_col_w5_banded"]
        assert len(x) == run["n_out"]
        if len(x) == 3:   # the reference's own consensus and 1-D basecalls for this pair
            assert "".join(x[1].split("\n")[1:]) == "".join(run["fasta_2d"].split("\n")[1:])
            seqs = lambda t: [l for l in t.split("\n") if not l.startswith(">")]    # (record names differ: file names)
            assert seqs(x[0]) == seqs(run["fasta_1d"])


@pytest.mark.timeout(300)
def test_two_ranks_shard_real_pairs(tmp_path, golden, golden_inputs, oracle):
    """torchrun-style launch (WORLD_SIZE = 2, gloo): each rank decodes its cost-balanced shard of the fixture pairs,
    rank 0 gathers the records in input order; equal to the single-process result and to the reference's outputs"""
    import argparse, json, pickle
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from _cpu_pair_decode import oracle_decode_pairs
    recs, pairs, args = _fixture_pairs(tmp_path, golden, golden_inputs)
    script = tmp_path / "pair_worker.py"
    script.write_text(PAIR_WORKER)
    outp = tmp_path / "gathered.pkl"
    env = dict(os.environ, PO_REPO=REPO, MASTER_ADDR="127.0.0.1", PO_ARGS=json.dumps(args), PO_PAIRS=json.dumps(pairs),
               PO_OUT=str(outp))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    with open(outp, "rb") as f:
        gathered = pickle.load(f)
    single = oracle_decode_pairs(pairs, argparse.Namespace(**args))
    assert [g[0] for g in gathered] == single                       # input order, identical records
    assert len({g[1] for g in gathered}) == 2                        # two processes did the decoding
    _check_against_reference(recs, [g[0] for g in gathered])


@pytest.mark.timeout(300)
def test_spawned_workers_shard_real_pairs(tmp_path, golden, golden_inputs, oracle):
    """plain launch on a multi-GPU node: one SPAWNED worker per device (two here; the injected CPU decode stands in
    for the engine), same plan, same gather"""
    import argparse
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from _cpu_pair_decode import oracle_decode_pairs, tagged_decode
    from poreover_amd.decoding import pair_decode
    recs, pairs, args = _fixture_pairs(tmp_path, golden, golden_inputs)
    ns = argparse.Namespace(**args)
    res = pair_decode.decode_pairs(pairs, ns, devices=[0, 1], decode_fn=tagged_decode)
    assert [r[0] for r in res] == oracle_decode_pairs(pairs, ns)
    assert len({r[1] for r in res}) == 2 and os.getpid() not in {r[1] for r in res}
    _check_against_reference(recs, [r[0] for r in res])
