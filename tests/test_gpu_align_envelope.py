"""GPU parity of the stand-alone alignment and envelope entry points (poreover.align, envelope.py)
vs the reference's golden alignments / envelopes and the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from poreover_amd import _lib
    _lib.load()
    return _lib


def test_alignments_golden(eng, golden):
    from poreover_amd import align, batch
    recs = golden["nw"]
    full = batch.align_batch([(r["s1"], r["s2"]) for r in recs], band_width=0)
    banded = batch.align_batch([(r["s1"], r["s2"]) for r in recs], band_width=500)
    for r, f, b in zip(recs, full, banded):
        assert list(f) == r["full"], (r["s1"][:20], "full")
        assert list(b) == r["banded"], (r["s1"][:20], "banded")
    narrow = [r for r in recs if "banded30" in r]
    got = batch.align_batch([(r["s1"], r["s2"]) for r in narrow], band_width=30)
    for r, g in zip(narrow, got):
        assert list(g) == r["banded30"]
    a1, a2, mat = align.global_pair("ACGTACGTTT", "ACGTCGTTTA")
    assert ("".join(a1), "".join(a2)) == ("ACGTACGTTT-", "ACGT-CGTTTA") and mat.shape == (11, 11) and mat.dtype == np.int32
    b1, b2 = align.global_pair_banded("ACGTACGTTT", "ACGTCGTTTA")
    assert ("".join(b1), "".join(b2)) == ("-ACGTACG-T-TT", "AACGT-CGTTT-A")     # the reference's banded quirk
    a1, a2, _ = align.global_pair("AC", "AC", match=5)        # score arguments as upstream (golden: test_gpu_prefix.py)
    assert ("".join(a1), "".join(a2)) == ("AC", "AC")


def test_envelope_golden(eng, golden, golden_inputs):
    from poreover_amd.decoding import envelope
    for rec in golden["pairs"]:
        run = rec["runs"].get("row_col_w5_banded")
        if not run or run["n_out"] != 3:
            continue
        y1, y2 = golden_inputs["pair%d_y1" % rec["index"]], golden_inputs["pair%d_y2" % rec["index"]]
        aln = np.array([list(run["alignment"][0]), list(run["alignment"][1])])
        cols = envelope.get_alignment_columns(aln)
        env = envelope.build_envelope(y1, y2, cols, rec["map1"], rec["map2"], padding=5)
        assert env.tolist() == run["envelope"], rec["index"]


def test_envelope_helpers_and_padding(eng, oracle, golden, golden_inputs):
    from poreover_amd.decoding import envelope
    rec = golden["pairs"][0]
    run = rec["runs"]["row_col_w5_banded"]
    y1, y2 = golden_inputs["pair%d_y1" % rec["index"]], golden_inputs["pair%d_y2" % rec["index"]]
    cols = envelope.get_alignment_columns(np.array([list(run["alignment"][0]), list(run["alignment"][1])]))
    for pad in (0, 5, 150):
        want = oracle.build_envelope(len(y1), len(y2), run["alignment"][0], run["alignment"][1], rec["map1"], rec["map2"], pad)
        assert np.array_equal(envelope.build_envelope(y1, y2, cols, rec["map1"], rec["map2"], padding=pad), want)
    e = np.full((6, 2), -1)
    envelope.add_block((1, 3, 4, 9), e)
    assert e.tolist() == [[-1, -1], [3, 9], [3, 9], [3, 9], [-1, -1], [-1, -1]]
    assert envelope.offset_envelope(np.array([[0, 4], [2, 6], [5, 9]]), (1, 3, 2, 9)).tolist() == [[0, 4], [3, 7]]


def test_global_pair_dp_matrix_golden(eng, golden):
    """align.global_pair's third return value (align.pyx:34-52,98): the dense DP matrix, against what the reference returned
    (tests/golden/make_golden_nw_matrix.py: sha-256 of the int32 bytes for the 24 + 5 alignment cases, four small ones in full)"""
    import hashlib
    import json
    from conftest import GOLDEN_DIR
    from poreover_amd import batch
    from poreover_amd.align import align
    with open(os.path.join(GOLDEN_DIR, "nw_matrix_golden.json")) as f:
        G = json.load(f)
    with open(os.path.join(GOLDEN_DIR, "extra_golden.json")) as f:
        extra = json.load(f)["align_scores"]
    mats = batch.nw_matrix_batch([(r["s1"], r["s2"]) for r in golden["nw"]])
    assert len(mats) == len(G["nw"]) == 24
    for m, w in zip(mats, G["nw"]):
        assert list(m.shape) == w["shape"] and m.dtype == np.int32
        assert [int(m[0, 0]), int(m[0, -1]), int(m[-1, 0]), int(m[-1, -1])] == w["corners"] and m[-1].tolist() == w["last_row"]
        assert hashlib.sha256(np.ascontiguousarray(m).tobytes()).hexdigest() == w["sha256"]
    for c, w in zip(extra, G["align_scores"]):
        m = align.global_pair(c["s1"], c["s2"], *c["scores"])[2]
        assert hashlib.sha256(np.ascontiguousarray(m).tobytes()).hexdigest() == w["sha256"], c["scores"]
    for c in G["small"]:
        m = align.global_pair(c["s1"], c["s2"], *c["scores"])[2]
        assert m.tolist() == c["matrix"], (c["s1"], c["s2"])
