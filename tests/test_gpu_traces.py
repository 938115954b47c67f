"""GPU: flip-flop TRACE files (BASELINE config 5's named inputs): the reference's Flappie .hdf5 and Guppy .fast5
samples (data fixtures under tests/golden/), read by the engine's own HDF5 reader, decoded through the C-ABI and
compared with what the reference decodes from them (tests/golden/make_golden_trace.py) and with the oracle.
uint8 traces are quantised, so EXACT score ties between sibling nodes are common here: this is the input class on
which Beam::prune's tie order (libstdc++'s partial_sort on the pointer-sorted candidates) shows in the output."""
import argparse
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from poreover_amd import _lib, batch
    _lib.load()
    return batch


@pytest.fixture(scope="module")
def traces():
    from poreover_amd.decoding import decode
    with open(os.path.join(GOLDEN_DIR, "trace_golden.json")) as f:
        tg = json.load(f)
    ms = {n: decode.model_from_trace(os.path.join(GOLDEN_DIR, tg[n]["file"]), tg[n]["basecaller"]) for n in ("flappie", "guppy")}
    return tg, ms


def test_ingest_uint8_bits(eng, traces):
    tg, ms = traces
    for n, m in ms.items():
        raw = m.engine_input()[0]
        got = eng.ingest_batch([raw])[0]
        assert np.allclose(got, m.log_prob, rtol=1e-15, atol=0)


@pytest.mark.parametrize("name", ["flappie", "guppy"])
def test_viterbi_and_beam_on_real_traces(eng, oracle, traces, name):
    tg, ms = traces
    rec, y = tg[name], ms[name].log_prob
    seqs, paths = eng.viterbi_batch([y], "flipflop", return_path=True)
    assert seqs[0] == rec["viterbi"]                                   # the reference's own Viterbi basecall
    lo, hi = rec["segment"]
    assert eng.beam_search_batch([y[lo:hi]], 10, model="ctc_flipflop")[0] == rec["segment_beam_w10"]
    assert eng.beam_search_batch([y[lo:hi]], 5, model="ctc_flipflop")[0] == rec["segment_beam_w5"]
    got5 = eng.beam_search_batch([y], 5, model="ctc_flipflop")[0]
    assert got5 == rec["beam_w5"]                                      # 49 k frames, ties included: the reference's output
    for W in (10, 25):
        assert eng.beam_search_batch([y], W, model="ctc_flipflop")[0] == oracle.cpp_beam_search(y, W, model_="ctc_flipflop"), W


def test_decode_driver_on_trace_files(eng, traces, tmp_path):
    from poreover_amd.decoding import decode
    tg, ms = traces
    for name, basecaller in (("flappie", "flappie"), ("guppy", "guppy")):
        rec = tg[name]
        for algo, want in (("viterbi", rec["viterbi"]), ("beam", rec["beam_w5"])):
            a = argparse.Namespace(out=str(tmp_path / (name + algo)), basecaller=basecaller, algorithm=algo, window=400,
                                   beam_width=5, threads=1)
            setattr(a, "in", [os.path.join(GOLDEN_DIR, rec["file"])])
            decode.decode(a)
            txt = open(str(tmp_path / (name + algo)) + ".fasta").read()
            assert "".join(txt.split("\n")[1:]) == want


def test_pair_decode_of_the_two_traces(eng, oracle, traces):
    """the two files hold the same read basecalled by Flappie and by Guppy: a real flip-flop PAIR (uint8 traces in,
    scaling on the device, envelope band up to 156 frames) vs the oracle"""
    tg, ms = traces
    r1, r2 = ms["flappie"].engine_input()[0], ms["guppy"].engine_input()[0]
    got = eng.pair_decode_stream([r1], [r2], "flipflop", 5, "row_col", return_envelope=True)[0]
    want = oracle.pair_decode(ms["flappie"].log_prob, ms["guppy"].log_prob, "flipflop", 5, "row_col")
    assert got["status"] == want["status"] == 0
    assert (got["seq1"], got["seq2"]) == (want["seq1"], want["seq2"])
    assert np.array_equal(got["envelope"], want["envelope"])
    assert got["consensus"] == want["consensus"]
