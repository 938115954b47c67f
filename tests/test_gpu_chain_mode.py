"""GPU: the CLOSED-FORM chain mode of the register-state pair kernel (po_set_chain_mode(PO_CHAIN_CLOSED_FORM): a new element's
window as one exp, a prefix sum and one log per time instead of the reference's serial logaddexp chain, PrefixTree.h:518-531).
Its values are not the reference's bits (they differ by ~ 1e-12), so these tests use north_star's budget the way
test_gpu_batch_scale does: Viterbi strings identical, consensus within 0.1 % edit distance — and print how many pairs differ
(so far: none).  The default mode (serial) is what every other test of the suite runs."""
import json
import os
from multiprocessing import get_context

import numpy as np
import pytest

from conftest import GOLDEN_DIR, REPO
from test_gpu_batch_scale import _digest, _gen, levenshtein

pytestmark = pytest.mark.gpu


@pytest.fixture()
def chain_mode():
    from poreover_amd import _lib
    _lib.load()
    yield _lib.set_chain_mode
    _lib.set_chain_mode("serial")


def test_chain_mode_switch(chain_mode):
    from poreover_amd import _lib
    assert _lib.get_chain_mode() == "serial"        # the default: the reference's chain
    chain_mode("closed_form")
    assert _lib.get_chain_mode() == "closed_form"
    with pytest.raises(_lib.EngineError):
        _lib.check(_lib.load(False).po_set_chain_mode(7), "po_set_chain_mode")
    chain_mode("serial")
    assert _lib.get_chain_mode() == "serial"


@pytest.mark.parametrize("mode", ["closed_form", "closed_guard3"])
@pytest.mark.parametrize("W,T,n", [(5, 4000, 256), (3, 900, 64), (6, 1500, 64), (10, 2000, 48), (12, 800, 32)])
def test_closed_form_within_edit_budget(chain_mode, oracle, mode, W, T, n):
    """both lane layouts (W <= 6 / 7..12), full-size reads; closed_guard3 = the 600-nat guard at 3 nats, so that most steps with
    new elements take the hand-over to the general scan (the serial chain) in the middle of the closed form's work"""
    from poreover_amd import _lib, batch
    from poreover_amd.synth import synth_pair
    if mode == "closed_guard3":
        n = max(16, n // 4)
    pairs = [synth_pair(1000 * W + i, T=T) for i in range(n)]
    chain_mode(mode)
    _lib.set_pair_route("reg")
    try:
        got = batch.pair_decode_batch([p[0] for p in pairs], [p[1] for p in pairs], "poreover", W, "row_col")
    finally:
        _lib.set_pair_route("auto")
        chain_mode("serial")
    differ, edits, total = 0, 0, 0
    for (y1, y2), g in zip(pairs, got):
        want = oracle.pair_decode(y1, y2, "poreover", W, "row_col")
        assert g["status"] == want["status"]
        assert (g["seq1"], g["seq2"]) == (want["seq1"], want["seq2"])      # Viterbi: bit-exact
        total += len(want["consensus"] or "")
        if (g["consensus"] or "") != (want["consensus"] or ""):
            differ += 1
            edits += levenshtein(g["consensus"] or "", want["consensus"] or "")
    print("chain mode %s, W = %d, T = %d: %d of %d pairs differ from the oracle, %d edits in %d consensus bases" % (mode, W, T, differ, n, edits, total))
    assert edits <= 0.001 * total, "%d pairs differ, %d edits in %d consensus bases" % (differ, edits, total)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "chain_mode_%s_W%d.json" % (mode, W)), "w") as f:
        json.dump({"mode": mode, "W": W, "T": T, "pairs": n, "pairs_that_differ": differ, "edits": edits, "consensus_bases": total}, f)


def test_closed_form_1024_pairs_vs_digest(chain_mode, oracle):
    """the first 1024 pairs of the bench workload against the committed digests, like test_batch_1024_pairs_vs_oracle_digest"""
    from poreover_amd import batch
    with get_context("spawn").Pool(min(8, os.cpu_count() or 1)) as pool:
        pairs = pool.map(_gen, range(1024), chunksize=16)
    with open(os.path.join(GOLDEN_DIR, "batch_digest.json")) as f:
        recs = json.load(f)["records"][:1024]
    chain_mode("closed_form")
    try:
        got = batch.pair_decode_batch([p[0] for p in pairs], [p[1] for p in pairs], "poreover", 5, "row_col")
    finally:
        chain_mode("serial")
    bad = [i for i, (g, r) in enumerate(zip(got, recs)) if _digest(g["seq1"], g["seq2"], g["consensus"]) != r[4]]
    edits = 0
    for i in bad:
        want = oracle.pair_decode(pairs[i][0], pairs[i][1], "poreover", 5, "row_col")
        assert (got[i]["seq1"], got[i]["seq2"]) == (want["seq1"], want["seq2"]), i
        edits += levenshtein(got[i]["consensus"] or "", want["consensus"] or "")
    total = sum(r[3] for r in recs)
    print("closed form, 1024 bench pairs: %d differ, %d edits in %d bases" % (len(bad), edits, total))
    assert edits <= 0.001 * total
