"""Pin the CPU oracle to the reference's outputs on REAL signal: the reference's own sample reads
(data/reads/read1.npy, read2.npy — 62 000 and 75 600 frames of float32 logits; fixtures made by
tests/golden/make_golden_real.py, which runs the reference's decode / pair_decode on them).
SURVEY.md §8(c) item 8: Viterbi md5 65ca2452895c, pair-decode (--reverse_complement, W = 5) row_col
b25482af8a0f / row 7ae38059c326, envelope of 62 000 rows and 1 490 722 cells, widest band 257."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, hexf


@pytest.fixture(scope="module")
def real():
    with open(os.path.join(GOLDEN_DIR, "real_golden.json")) as f:
        g = json.load(f)
    return g, dict(np.load(os.path.join(GOLDEN_DIR, "real_inputs.npz")))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _consensus(fasta):
    return "".join(fasta.split("\n")[1:])


def test_load_logits_bits(oracle, real):
    g, inp = real
    y1 = oracle.load_logits(inp["read1_logits"]).astype(np.float64)
    y2 = oracle.load_logits(inp["read2_logits"]).astype(np.float64)
    assert [list(y1.shape), list(y2.shape)] == g["log_prob_shape"]
    assert [hexf(x) for x in g["log_prob_rows"]["read1_0"]] == y1[0].tolist()
    assert [hexf(x) for x in g["log_prob_rows"]["read2_31337"]] == y2[31337].tolist()
    assert [_sha(y1), _sha(y2)] == g["log_prob_sha256"]
    assert _sha(oracle.reverse_complement(y2)) == g["log_prob_revcomp_sha256"]


def test_viterbi_real(oracle, real):
    g, inp = real
    y1 = oracle.load_logits(inp["read1_logits"]).astype(np.float64)
    y2 = oracle.load_logits(inp["read2_logits"]).astype(np.float64)
    s1, p1 = oracle.viterbi_decode(y1)
    s2, p2 = oracle.viterbi_decode(y2)
    assert s1 == g["viterbi1"] and s2 == g["viterbi2_forward"]
    assert hashlib.md5(s1.encode()).hexdigest()[:12] == "65ca2452895c" and len(s1) == 6618
    assert hashlib.md5(s2.encode()).hexdigest()[:12] == "f44652777c20" and len(s2) == 6580
    assert [_sha(p1.astype(np.int8)), _sha(p2.astype(np.int8))] == g["path_sha256"]
    assert oracle.viterbi_decode(oracle.reverse_complement(y2))[0] == g["viterbi2_revcomp"]
    lo, hi = g["segment"]
    seg = y1[lo:hi]
    assert oracle.viterbi_decode(seg)[0] == g["segment_viterbi"]
    assert oracle.cpp_forward(seg, g["segment_viterbi"]) == hexf(g["segment_forward_viterbi"])
    assert oracle.cpp_beam_search(seg, 10, model_="ctc_merge_repeats") == g["segment_beam_merge_w10"]


def test_beam1d_real_w5(oracle, real):
    g, inp = real
    y1 = oracle.load_logits(inp["read1_logits"]).astype(np.float64)
    assert oracle.cpp_beam_search(y1, 5) == g["beam1d_read1"]["5"]


@pytest.mark.parametrize("method", ["row_col", "row"])
def test_pair_decode_real_revcomp(oracle, real, method):
    g, inp = real
    y1 = oracle.load_logits(inp["read1_logits"]).astype(np.float64)
    y2 = oracle.reverse_complement(oracle.load_logits(inp["read2_logits"]).astype(np.float64))
    want = g["pair_revcomp"][method + "_w5"]
    res = oracle.pair_decode(y1, y2, beam_width=5, method=method)
    assert res["seq1"] == g["viterbi1"] and res["seq2"] == g["viterbi2_revcomp"]
    assert res["sequence_identity"] == hexf(want["summary"]["sequence_identity"])
    assert np.array_equal(res["envelope"], inp["envelope"])
    assert int((inp["envelope"][:, 1] - inp["envelope"][:, 0]).sum()) == 1490722
    assert res["consensus"] == _consensus(want["fasta_2d"])
    assert hashlib.md5(res["consensus"].encode()).hexdigest()[:12] == {"row_col": "b25482af8a0f", "row": "7ae38059c326"}[method]
