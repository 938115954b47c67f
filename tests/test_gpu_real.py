"""GPU parity on REAL signal: the reference's sample reads (62 000 / 75 600 frames of float32 logits), through
the device ingest kernel, the 1-D kernels and the whole pair-decode chain with --reverse_complement, against the
outputs the reference produced for them (tests/golden/make_golden_real.py).  This is the input class the
synthetic vectors do not cover: a widest envelope band of 257 frames (beyond what the W <= 6 value store of
beam2d_kernel holds: the pair takes the retry pass), 10 frames per base, reads fifteen times the bench length."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, hexf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from poreover_amd import _lib, batch
    _lib.load()
    return batch


@pytest.fixture(scope="module")
def real(oracle):
    with open(os.path.join(GOLDEN_DIR, "real_golden.json")) as f:
        g = json.load(f)
    inp = dict(np.load(os.path.join(GOLDEN_DIR, "real_inputs.npz")))
    # the reference's float64 log-probabilities, bit for bit (digests checked by tests/test_oracle_real.py)
    y1 = oracle.load_logits(inp["read1_logits"]).astype(np.float64)
    y2 = oracle.load_logits(inp["read2_logits"]).astype(np.float64)
    return g, inp, y1, y2


def _cons(fasta):
    return "".join(fasta.split("\n")[1:])


def _edit_distance(a, b):
    """banded Levenshtein distance (band 64 around the diagonal; enough for near-identical strings)"""
    n, m = len(a), len(b)
    band = 64 + abs(n - m)
    prev = {j: j for j in range(0, min(m, band) + 1)}
    for i in range(1, n + 1):
        cur = {}
        for j in range(max(0, i - band), min(m, i + band) + 1):
            if j == 0:
                cur[j] = i
                continue
            best = prev.get(j - 1, 1 << 30) + (a[i - 1] != b[j - 1])
            best = min(best, prev.get(j, 1 << 30) + 1, cur.get(j - 1, 1 << 30) + 1)
            cur[j] = best
        prev = cur
    return prev[m]


def test_ingest_real(eng, real):
    g, inp, y1, y2 = real
    got1, got2 = eng.ingest_batch([np.concatenate(inp["read1_logits"]), np.concatenate(inp["read2_logits"])])
    # float32 arithmetic on both sides (device expf / log1pf vs numpy's): x - lse agrees to 2 float32 ulp of the
    # subtrahend lse (|lse| < 16 here: ulp 9.5e-7), most values bit for bit
    for got, want in ((got1, y1), (got2, y2)):
        assert got.shape == want.shape and got.dtype == np.float64
        assert np.abs(got - want).max() <= 2e-6
        assert (got == want).mean() > 0.5
        assert np.array_equal(got.astype(np.float32).astype(np.float64), got)      # float32 values, widened
    rc = eng.ingest_batch([np.concatenate(inp["read2_logits"])], perm=[3, 2, 1, 0, 4], reverse=True)[0]
    assert np.array_equal(rc, got2[::-1][:, [3, 2, 1, 0, 4]])


def test_viterbi_real(eng, real):
    g, inp, y1, y2 = real
    (s1, s2), paths = eng.viterbi_batch([y1, y2], return_path=True)
    assert s1 == g["viterbi1"] and s2 == g["viterbi2_forward"]
    assert hashlib.md5(s1.encode()).hexdigest()[:12] == "65ca2452895c"
    sha = [hashlib.sha256(np.ascontiguousarray(p.astype(np.int8)).tobytes()).hexdigest() for p in paths]
    assert sha == g["path_sha256"]
    lo, hi = g["segment"]
    assert eng.viterbi_batch([y1[lo:hi]])[0] == g["segment_viterbi"]
    assert eng.forward_batch([y1[lo:hi]], [g["segment_viterbi"]])[0] == pytest.approx(hexf(g["segment_forward_viterbi"]), rel=1e-12)
    assert eng.beam_search_batch([y1[lo:hi]], 10, model="ctc_merge_repeats")[0] == g["segment_beam_merge_w10"]


def test_beam1d_real(eng, real):
    g, inp, y1, y2 = real
    for W in (5, 10, 25):
        assert eng.beam_search_batch([y1], W)[0] == g["beam1d_read1"][str(W)], W


@pytest.mark.parametrize("method,route", [("row_col", "auto"), ("row_col", "legacy"), ("row", "auto")])
def test_pair_decode_real_revcomp(eng, real, method, route):
    """the reference's float64 log-probabilities in, every stage output compared (row_col on both one-pair-per-wave
    kernels: 62 000 x 75 600 frames, windows up to 257 wide — beyond the 254 frames the register-state kernel's packed walk records hold: it hands the pair to beam2d_kernel)"""
    from poreover_amd import _lib
    g, inp, y1, y2 = real
    y2rc = np.ascontiguousarray(y2[::-1][:, [3, 2, 1, 0, 4]])           # transducer.py:68-70
    want = g["pair_revcomp"][method + "_w5"]
    _lib.set_pair_route(route)
    try:
        res = eng.pair_decode_batch([y1], [y2rc], "poreover", 5, method)[0]
    finally:
        _lib.set_pair_route("auto")
    assert res["status"] == 0
    assert res["seq1"] == g["viterbi1"] and res["seq2"] == g["viterbi2_revcomp"]
    assert res["sequence_identity"] == hexf(want["summary"]["sequence_identity"])
    assert np.array_equal(res["envelope"], inp["envelope"])
    assert res["consensus"] == _cons(want["fasta_2d"])
    assert hashlib.md5(res["consensus"].encode()).hexdigest()[:12] == want["consensus_md5_12"]


def test_device_chain_real(eng, real):
    """float32 logits -> ingest kernel (log-softmax; read 2 reverse-complemented on the device) -> pair decode.
    The ingest kernel's float32 exp / log1p may differ from numpy's in the last float32 bit, so the beam search
    sees inputs that differ by <= 2e-7 relative: tolerance 0.1 % edit distance (BASELINE north_star), Viterbi
    strings identical unless such a difference flips an argmax."""
    g, inp, y1, y2 = real
    d1 = eng.ingest_batch([np.concatenate(inp["read1_logits"])])[0]
    d2 = eng.ingest_batch([np.concatenate(inp["read2_logits"])], perm=[3, 2, 1, 0, 4], reverse=True)[0]
    res = eng.pair_decode_batch([d1], [d2], "poreover", 5, "row_col")[0]
    assert res["status"] == 0
    want = _cons(g["pair_revcomp"]["row_col_w5"]["fasta_2d"])
    for got, ref in ((res["seq1"], g["viterbi1"]), (res["seq2"], g["viterbi2_revcomp"]), (res["consensus"], want)):
        if got != ref:
            assert _edit_distance(got, ref) <= 0.001 * len(ref)


def test_real_segments_batch(eng, oracle, real):
    """a batch of T = 4000 stretches of the real pair (read 2 reverse-complemented), cut along the reference's
    envelope so that both stretches cover the same bases: the batched path on real signal vs the oracle"""
    g, inp, y1, y2 = real
    y2rc = np.ascontiguousarray(y2[::-1][:, [3, 2, 1, 0, 4]])
    env = inp["envelope"]
    a, b = [], []
    for u0 in range(2000, 58000, 4000):
        v0, v1 = int(env[u0, 0]), int(env[u0 + 3999, 1])
        a.append(y1[u0:u0 + 4000]); b.append(y2rc[v0:v1])
    got = eng.pair_decode_batch(a, b, "poreover", 5, "row_col")
    for i, (p, q) in enumerate(zip(a, b)):
        want = oracle.pair_decode(p, q, "poreover", 5, "row_col")
        assert got[i]["status"] == want["status"], i
        assert (got[i]["seq1"], got[i]["seq2"]) == (want["seq1"], want["seq2"]), i
        if want["status"] == 0:
            assert np.array_equal(got[i]["envelope"], want["envelope"]), i
            assert got[i]["consensus"] == want["consensus"], i


def test_decode_driver_real_logits_file(eng, real, tmp_path):
    """`decode` on the reference's sample read as its .npy holds it (float32 logits): uploaded as float32, log-softmax
    on the device, one engine call — the reference's Viterbi basecall (0 edits expected) and beam search (<= 0.1 %)"""
    import argparse
    from poreover_amd.decoding import decode
    g, inp, y1, y2 = real
    np.save(tmp_path / "read1.npy", inp["read1_logits"])
    for algo, want in (("viterbi", g["viterbi1"]), ("beam", g["beam1d_read1"]["25"])):
        a = argparse.Namespace(out=str(tmp_path / algo), basecaller="poreover", algorithm=algo, window=400, beam_width=25, threads=1)
        setattr(a, "in", [str(tmp_path / "read1.npy")])
        decode.decode(a)
        got = "".join(open(str(tmp_path / algo) + ".fasta").read().split("\n")[1:])
        if got != want:
            assert _edit_distance(got, want) <= 0.001 * len(want), algo
