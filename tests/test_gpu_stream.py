"""GPU: the pipelined host layer (po_pipeline_pair_decode, batch.pair_decode_stream) — host arrays in the
basecaller's own form (float32 logits / uint8 traces / float64 log-probabilities) in, strings out, in waves over three
slots — against the one-shot batched call, the oracle and the reference's outputs on its real sample reads."""
import argparse
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from poreover_amd.synth import synth_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from poreover_amd import _lib, batch
    _lib.load()
    return batch


def _cons(fasta):
    return "".join(fasta.split("\n")[1:])


@pytest.mark.parametrize("kind", ["poreover", "bonito", "flipflop"])
def test_stream_equals_batch_many_waves(eng, kind):
    y1s, y2s = [], []
    for i in range(37):
        a, b = synth_pair(9100 + i, T=200 + 37 * (i % 9), flipflop=(kind == "flipflop"))
        y1s.append(a); y2s.append(b)
    want = eng.pair_decode_batch(y1s, y2s, kind, 5, "row_col")
    st = {}
    # 7 pairs per wave: 6 waves, each slot reused three times; the buffers grow on the way
    got = eng.pair_decode_stream(y1s, y2s, kind, 5, "row_col", return_envelope=True, wave_pairs=7, threads=3, stats=st)
    assert st["waves"] == 6
    for g, w in zip(got, want):
        assert g["status"] == w["status"]
        assert (g["seq1"], g["seq2"], g["consensus"]) == (w["seq1"], w["seq2"], w["consensus"])
        assert g["sequence_identity"] == w["sequence_identity"]
        if w["status"] == 0:
            assert np.array_equal(g["envelope"], w["envelope"])
    # a wave cut by rows instead of pairs, default threads, no envelope requested
    got2 = eng.pair_decode_stream(y1s, y2s, kind, 5, "row_col", wave_rows=3000)
    assert [g["consensus"] for g in got2] == [w["consensus"] for w in want]
    assert all(g["envelope"] is None for g in got2)


def test_stream_logits_f32_on_device(eng, oracle):
    """float32 logits in: log-softmax, reverse complement of read 2 on the device; vs the oracle fed with the
    reference's host arithmetic (float32 logsumexp): identical basecalls expected, tolerance 0.1 % edits"""
    rng = np.random.default_rng(5)
    l1s, l2s = [], []
    for i in range(9):
        y1, y2 = synth_pair(9300 + i, T=600)
        # logits whose log-softmax is y (any per-frame shift), stored reverse-complemented for read 2
        l1s.append((y1 + rng.normal(0, 3, (len(y1), 1))).astype(np.float32))
        l2s.append(np.ascontiguousarray((y2 + rng.normal(0, 3, (len(y2), 1)))[::-1][:, [3, 2, 1, 0, 4]]).astype(np.float32))
    got = eng.pair_decode_stream(l1s, l2s, "poreover", 5, "row_col", perm2=[3, 2, 1, 0, 4], reverse2=True, wave_pairs=4)
    edits = total = 0
    from test_gpu_batch_scale import levenshtein
    for g, a, b in zip(got, l1s, l2s):
        h1 = oracle.load_logits(a[None]).astype(np.float64)
        h2 = oracle.reverse_complement(oracle.load_logits(b[None]).astype(np.float64))
        w = oracle.pair_decode(h1, h2, "poreover", 5, "row_col")
        assert g["status"] == w["status"]
        edits += levenshtein(g["consensus"] or "", w["consensus"] or "") + levenshtein(g["seq1"], w["seq1"])
        total += len(w["consensus"] or "") + len(w["seq1"])
    assert edits <= 0.001 * total


def test_stream_uint8_traces(eng, oracle):
    """uint8 flip-flop traces in (decode.py:89-93 scaling on the device): float64 arithmetic on both sides"""
    tr1, tr2 = [], []
    for i in range(5):
        y1, y2 = synth_pair(9400 + i, T=500, flipflop=True)
        tr1.append(np.clip(np.rint(np.exp(y1) * 255), 0, 255).astype(np.uint8))
        tr2.append(np.clip(np.rint(np.exp(y2) * 255), 0, 255).astype(np.uint8))
    got = eng.pair_decode_stream(tr1, tr2, "flipflop", 5, "row_col")
    for g, a, b in zip(got, tr1, tr2):
        w = oracle.pair_decode(oracle.trace_to_log_prob(a), oracle.trace_to_log_prob(b), "flipflop", 5, "row_col")
        assert g["status"] == w["status"]
        assert (g["seq1"], g["seq2"], g["consensus"]) == (w["seq1"], w["seq2"], w["consensus"])


def test_stream_real_pair_logits(eng):
    """the reference's sample pair as the file holds it (float32 logits, read 2 to be reverse-complemented)"""
    with open(os.path.join(GOLDEN_DIR, "real_golden.json")) as f:
        g = json.load(f)
    inp = dict(np.load(os.path.join(GOLDEN_DIR, "real_inputs.npz")))
    r = eng.pair_decode_stream([np.concatenate(inp["read1_logits"])], [np.concatenate(inp["read2_logits"])], "poreover", 5,
                               "row_col", perm2=[3, 2, 1, 0, 4], reverse2=True, return_envelope=True)[0]
    want = _cons(g["pair_revcomp"]["row_col_w5"]["fasta_2d"])
    from test_gpu_batch_scale import levenshtein
    assert r["status"] == 0
    assert levenshtein(r["seq1"], g["viterbi1"]) <= 0.001 * len(g["viterbi1"])
    if r["consensus"] != want:
        assert abs(len(r["consensus"]) - len(want)) < 20


def test_driver_logits_files_revcomp(eng, tmp_path, oracle):
    """pair-decode driver on .npy files of float32 logits with --reverse_complement: deferred ingest (device) gives
    the same records as the host ingest of the same files"""
    from poreover_amd.decoding import pair_decode, decode
    rng = np.random.default_rng(11)
    lines = []
    for i in range(6):
        y1, y2 = synth_pair(9500 + i, T=800)
        n1, n2 = len(y1) // 100 * 100, len(y2) // 100 * 100
        l1 = (y1[:n1] + rng.normal(0, 2, (n1, 1))).astype(np.float32).reshape(-1, 100, 5)
        l2 = np.ascontiguousarray((y2[:n2] + rng.normal(0, 2, (n2, 1)))[::-1][:, [3, 2, 1, 0, 4]]).astype(np.float32).reshape(-1, 100, 5)
        np.save(tmp_path / ("a%d.npy" % i), l1)
        np.save(tmp_path / ("b%d.npy" % i), l2)
        lines.append("a%d.npy\tb%d.npy" % (i, i))
    (tmp_path / "pairs.txt").write_text("\n".join(lines) + "\n")
    a = argparse.Namespace(dir=str(tmp_path), basecaller="poreover", reverse_complement=True, out=str(tmp_path / "o"),
                           threads=1, method="envelope", single="viterbi", logging="info", debug=False, algorithm="beam",
                           alignment="banded", beam_width=5, debug_envelope=False, diagonal_envelope=False,
                           diagonal_width=50, padding=5, skip_matches=False, skip_threshold=10,
                           beam_search_method="row_col", window=200)
    setattr(a, "in", [str(tmp_path / "pairs.txt")])
    m = decode.model_from_trace(str(tmp_path / "b0.npy"), "poreover")
    assert m.engine_input()[1] == 0          # deferred: float32 logits go to the device as they are
    pair_decode.pair_decode(a)
    got = open(str(tmp_path / "o") + ".2d.fasta").read().split(">")[1:]
    assert len(got) == 6
    from test_gpu_batch_scale import levenshtein
    edits = total = 0
    for i, rec in enumerate(got):
        h1 = oracle.load_logits(np.load(tmp_path / ("a%d.npy" % i))).astype(np.float64)
        h2 = oracle.reverse_complement(oracle.load_logits(np.load(tmp_path / ("b%d.npy" % i))).astype(np.float64))
        w = oracle.pair_decode(h1, h2, "poreover", 5, "row_col")
        assert rec.split("\n")[0] == "consensus;a%d;b%d" % (i, i)
        edits += levenshtein(_cons(">" + rec), w["consensus"])
        total += len(w["consensus"])
    assert edits <= 0.001 * total


@pytest.mark.parametrize("kind", ["bonito", "poreover"])
def test_alternating_wave_geometries_on_one_pipeline(eng, kind):
    """the pair beam kernels keep {magic, epoch} words in the workspace instead of clearing their value
    store per launch; a pipeline reuses ONE workspace buffer for waves of different geometry, which moves the
    sub-workspace.  Alternating two geometries for many calls (the steady state a single pass never reaches) must
    give, call after call, the results of the first call (the host now tracks the layout last used on a buffer)."""
    from poreover_amd import _lib
    batches = []
    for base, n, T in ((9300, 41, 330), (9400, 17, 520)):
        y1s, y2s = [], []
        for i in range(n):
            a, b = synth_pair(base + i, T=T + 29 * (i % 7))
            y1s.append(a); y2s.append(b)
        batches.append((y1s, y2s))
    for route in ("auto", "legacy"):
        _lib.set_pair_route(route)
        try:
            first = [None, None]
            for rep in range(8):
                j = rep & 1
                got = [r["consensus"] for r in eng.pair_decode_stream(batches[j][0], batches[j][1], kind, 5, "row_col", wave_pairs=16)]
                if first[j] is None:
                    first[j] = got
                assert got == first[j], (route, rep)
        finally:
            _lib.set_pair_route("auto")
    want = [r["consensus"] for r in eng.pair_decode_batch(batches[0][0], batches[0][1], kind, 5, "row_col")]
    assert first[0] == want


def test_stream_large_job_records_built_while_decoding(eng, oracle):
    """More than 4 096 pairs: several waves on three slots, and the Python records of finished waves are built while the later
    waves decode (the engine writes a pair's status last; batch._PENDING marks "not yet").  4 700 pairs — 14 distinct ones,
    one of them with a read too short to align (skipped, consensus None) — every record in its place, against the oracle;
    strict=True raises for an engine error only after the call has come back."""
    rng = np.random.default_rng(23)
    base = []
    for i in range(14):
        a, b = synth_pair(7300 + i, T=120 + 40 * (i % 5))
        if i == 5:
            b = b[:3]                      # length skip / identity skip upstream: no consensus
        base.append((a, b, oracle.pair_decode(a, b, "poreover", 5, "row_col")))
    idx = rng.integers(len(base), size=4700)
    st = {}
    got = eng.pair_decode_stream([base[i][0] for i in idx], [base[i][1] for i in idx], "poreover", 5, "row_col", strict=False, stats=st)
    assert st["waves"] >= 2 and len(got) == len(idx)   # (round 4: a short first wave, then full ones)
    for k, i in enumerate(idx):
        w = base[i][2]
        assert got[k]["seq1"] == w["seq1"] and got[k]["seq2"] == w["seq2"], k
        assert (got[k]["consensus"] or None) == (w["consensus"] or None), k
        assert (got[k]["status"] == 0) == (w["status"] == 0), k
