"""GPU parity at BATCH scale (BASELINE config 4's per-GPU job in miniature): the first 1024 pairs of the bench
workload (synth_pair(seed, T=4000), W = 5, row_col) decoded by ONE po_pair_decode_batch call, compared pair by
pair with the committed digests of the CPU oracle's results (tests/golden/batch_digest.json, made by
tests/golden/make_batch_digest.py).  Any pair whose digest differs is decoded again by the oracle and must stay
inside the stated tolerance: identical Viterbi basecalls, consensus within 0.1 % edit distance over the batch."""
import json
import os
from multiprocessing import get_context

import numpy as np
import pytest

from conftest import GOLDEN_DIR, REPO

pytestmark = pytest.mark.gpu

NPAIRS = 1024


def _gen(seed):
    from poreover_amd.synth import synth_pair
    return synth_pair(seed, T=4000)


def _digest(seq1, seq2, cons):
    import hashlib
    return hashlib.md5(("%s|%s|%s" % (seq1, seq2, cons if cons is not None else "")).encode()).hexdigest()[:10]


def levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


@pytest.fixture(scope="module")
def workload():
    # generated in worker processes started with "spawn" (this pytest process holds a HIP context)
    with get_context("spawn").Pool(min(8, os.cpu_count() or 1)) as pool:
        pairs = pool.map(_gen, range(NPAIRS), chunksize=16)
    with open(os.path.join(GOLDEN_DIR, "batch_digest.json")) as f:
        dig = json.load(f)
    assert dig["T"] == 4000 and dig["beam_width"] == 5 and dig["method"] == "row_col" and len(dig["records"]) >= NPAIRS
    return pairs, dig["records"][:NPAIRS]


@pytest.mark.parametrize("route", ["default", "reg", "legacy"])
def test_batch_1024_pairs_vs_oracle_digest(workload, oracle, route, monkeypatch):
    from poreover_amd import _lib, batch
    _lib.load()
    _lib.set_pair_route({"reg": "reg", "legacy": "legacy"}.get(route, "auto"))
    pairs, recs = workload
    try:
        got = batch.pair_decode_batch([p[0] for p in pairs], [p[1] for p in pairs], "poreover", 5, "row_col")
    finally:
        _lib.set_pair_route("auto")
    assert len(got) == NPAIRS
    bad = []
    for i, (g, r) in enumerate(zip(got, recs)):
        st, l1, l2, lc, dg = r
        if (g["status"], g["length1"], g["length2"], len(g["consensus"] or "")) != (st, l1, l2, lc) or \
                _digest(g["seq1"], g["seq2"], g["consensus"]) != dg:
            bad.append(i)
    edits = 0
    for i in bad:   # beam search near-ties may legitimately differ in the last ulp of exp / log: bounded by edit distance
        want = oracle.pair_decode(pairs[i][0], pairs[i][1], "poreover", 5, "row_col")
        assert got[i]["status"] == want["status"], i
        assert (got[i]["seq1"], got[i]["seq2"]) == (want["seq1"], want["seq2"]), i      # Viterbi: bit-exact
        edits += levenshtein(got[i]["consensus"] or "", want["consensus"] or "")
    total = sum(r[3] for r in recs)
    assert edits <= 0.001 * total, "%d pairs differ, %d edits in %d consensus bases" % (len(bad), edits, total)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "batch_scale_%s.json" % route), "w") as f:
        json.dump({"pairs": NPAIRS, "route": route, "digest_mismatches": len(bad), "edits": edits, "consensus_bases": total}, f)


def test_more_pairs_than_workgroups_mixed_lengths(oracle):
    """6 000 pairs of very different lengths (T 200 - 3 000) in one pair_decode_batch call: more than the 4 096 resident
    workgroups of beam2d_kernel, so the launch takes its pairs longest first (pair_order_kernel).  Every pair must come
    back in its own place with its own result: 24 distinct pairs, replicated in random order, against the oracle."""
    from poreover_amd import batch
    from poreover_amd.synth import synth_pair
    rng = np.random.default_rng(17)
    base = []
    for i in range(24):
        y1, y2 = synth_pair(1000 + i, T=int(np.exp(rng.uniform(np.log(200), np.log(3000)))))
        w = oracle.pair_decode(y1, y2, "poreover", 5, "row_col")
        base.append((y1, y2, w))
    idx = rng.integers(len(base), size=6000)
    got = batch.pair_decode_batch([base[i][0] for i in idx], [base[i][1] for i in idx], "poreover", 5, "row_col")
    assert len(got) == len(idx)
    for k, i in enumerate(idx):
        w = base[i][2]
        assert got[k]["status"] == w["status"] and (got[k]["consensus"] or "") == (w["consensus"] or "") and got[k]["seq1"] == w["seq1"], (k, int(i))


# ---- the secondary legs of bench.py at test size, against the oracle's committed digests (tests/golden/secondary_digest.json,
# made by make_secondary_digest.py): the occupancies the small parity tests never reach
@pytest.fixture(scope="module")
def secondary_digests():
    with open(os.path.join(GOLDEN_DIR, "secondary_digest.json")) as f:
        return json.load(f)["legs"]


def _gen_ff_read(i):
    from poreover_amd.synth import synth_read
    return synth_read(500000 + i, 4000, 0, True)


def _gen_ff_pair(i):
    from poreover_amd.synth import synth_pair
    return synth_pair(700000 + i, T=4000, flipflop=True)


def _md5(*parts):
    import hashlib
    return hashlib.md5("|".join(p if p is not None else "" for p in parts).encode()).hexdigest()[:10]


@pytest.mark.parametrize("leg", ["config2", "config5"])
def test_beam1d_1000_reads_one_launch_vs_oracle_digest(workload, secondary_digests, leg):
    """BASELINE configs 2 and 5 as ONE launch of 1 000 reads (one wave per SIMD: the occupancy bench.py times) — every string
    against the oracle's digest"""
    from poreover_amd import batch
    want = secondary_digests[leg]
    n = len(want)
    if leg == "config2":
        pairs, _ = workload
        reads = [p[0] for p in pairs][:n]
        model = "ctc"
    else:
        with get_context("spawn").Pool(min(8, os.cpu_count() or 1)) as pool:
            reads = pool.map(_gen_ff_read, range(n), chunksize=16)
        model = "ctc_flipflop"
    n = min(n, len(reads))
    got = batch.beam_search_batch(reads[:n], beam_width=10, model=model)
    bad = [i for i in range(n) if [0, len(got[i]), _md5(got[i])] != want[i]]
    assert bad == [], (leg, bad[:5])


@pytest.mark.parametrize("leg,kind,W,method", [("pair_bonito_W5", "bonito", 5, "row_col"), ("pair_row_W5", "poreover", 5, "row"),
                                               ("pair_row_col_W10", "poreover", 10, "row_col"), ("pair_flipflop_W5", "flipflop", 5, "row_col"),
                                               ("pair_row_W25", "poreover", 25, "row"), ("pair_row_col_W25", "poreover", 25, "row_col")])
def test_secondary_pair_legs_512_pairs_vs_oracle_digest(workload, secondary_digests, leg, kind, W, method):
    """the other pair-decode configurations bench.py times (Bonito's tree model, method row, W = 10, flip-flop pairs, and — round 6 —
    the Python API's literal defaults W = 25 / row with row_col beside it), 512 pairs in one call each, whole stage chain: statuses,
    lengths and strings against the oracle's digests"""
    from poreover_amd import _lib, batch
    n = 512
    want = secondary_digests[leg][:n]
    if kind == "flipflop":
        with get_context("spawn").Pool(min(8, os.cpu_count() or 1)) as pool:
            pairs = pool.map(_gen_ff_pair, range(n), chunksize=16)
    else:
        pairs = workload[0][:n]
    got = batch.pair_decode_stream([p[0] for p in pairs], [p[1] for p in pairs], kind, W, method, strict=False)
    bad = []
    for i, (g, r) in enumerate(zip(got, want)):
        if g["status"] in (0, _lib.SKIP_LENGTH, _lib.SKIP_IDENTITY):
            rec = [g["status"], g["length1"], g["length2"], len(g["consensus"] or ""), _md5(g["seq1"], g["seq2"], g["consensus"])]
        else:   # (refused as the reference's own assertion refuses it: the oracle records the code with empty strings)
            rec = [g["status"], 0, 0, 0, _md5("", "", "")]
        if rec != r:
            bad.append(i)
    assert bad == [], (leg, bad[:5], [got[i]["status"] for i in bad[:5]])
