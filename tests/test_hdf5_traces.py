"""The HDF5 trace route (decode.py:53-65,89-104): the engine's own reader on the reference's two flip-flop trace
files (data fixtures in tests/golden/), and the oracle pinned to what the reference decodes from them
(tests/golden/make_golden_trace.py).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR


@pytest.fixture(scope="module")
def tg():
    with open(os.path.join(GOLDEN_DIR, "trace_golden.json")) as f:
        return json.load(f)


def _load(name, tg):
    from poreover_amd.decoding import decode
    rec = tg[name]
    return decode.model_from_trace(os.path.join(GOLDEN_DIR, rec["file"]), rec["basecaller"]), rec


@pytest.mark.parametrize("name", ["flappie", "guppy"])
def test_reader_and_host_ingest(name, tg, oracle):
    m, rec = _load(name, tg)
    assert m.kind == "flipflop" and m.engine_input()[1] == 1          # uint8 trace, deferred to the device ingest
    raw = m.engine_input()[0]
    assert raw.dtype == np.uint8 and list(raw.shape) == rec["shape"]
    y = m.log_prob                                                     # the reference's host arithmetic
    assert hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest() == rec["log_prob_sha256"]
    assert np.array_equal(y, oracle.trace_to_log_prob(raw))


@pytest.mark.parametrize("name", ["flappie", "guppy"])
def test_oracle_decodes_traces_like_the_reference(name, tg, oracle):
    m, rec = _load(name, tg)
    y = m.log_prob
    seq, path = oracle.viterbi_decode(y, "flipflop")
    assert seq == rec["viterbi"]
    assert hashlib.sha256(path.astype(np.int8).tobytes()).hexdigest() == rec["path_sha256"]
    lo, hi = rec["segment"]
    assert oracle.cpp_beam_search(y[lo:hi], 10, model_="ctc_flipflop") == rec["segment_beam_w10"]
    assert oracle.cpp_beam_search(y[lo:hi], 5, model_="ctc_flipflop") == rec["segment_beam_w5"]
    assert oracle.cpp_beam_search(y, 5, model_="ctc_flipflop") == rec["beam_w5"]


def test_reader_other_datasets(tg):
    from poreover_amd.decoding import hdf5_lite
    f = hdf5_lite.File(os.path.join(GOLDEN_DIR, "guppy_flipflop.fast5"))
    assert f.keys() == ["Analyses", "Raw", "UniqueGlobalKey"]
    g = f["/Analyses/Basecall_1D_000/BaseCalled_template"]
    assert g.keys() == ["Fastq", "Move", "Trace"]
    fq = np.array(g["Fastq"]).tobytes().decode().split("\n")
    assert fq[0].startswith("@") and len(fq[1]) == tg["guppy"]["fastq_len"] and set(fq[1]) <= set("ACGT")
    mv = np.array(g["Move"])
    assert mv.ndim == 1 and mv.dtype == np.uint8 and len(mv) == tg["guppy"]["shape"][0] and mv.max() <= 1
    with pytest.raises(KeyError):
        f["/Analyses/Nope"]
    with pytest.raises(hdf5_lite.Hdf5Error):
        hdf5_lite.File(os.path.join(GOLDEN_DIR, "golden.json"))
