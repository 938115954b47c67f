"""GPU parity of cpp_forward and cpp_viterbi_acceptor vs the oracle and the golden values of the
reference's tests (tests/test_forward.py, tests/test_transducer.py)."""
import numpy as np
import pytest

from conftest import hexf
from poreover_amd.synth import synth_pair

pytestmark = pytest.mark.gpu

# device exp/log (ocml) vs host libm may differ in the last ulp; each forward value sums up to T * |label|
# logaddexp terms, so the comparison is relative 1e-12 (the reference's own test uses np.isclose, 1e-5)
RTOL = 1e-12


@pytest.fixture(scope="module")
def dec():
    from poreover_amd import _lib, decoding
    _lib.load()
    return decoding


def test_forward_golden_toys(dec, golden):
    toy, g = golden["toy_prob"], golden["toy"]
    t1f = np.log(np.array(toy["t1"], dtype=np.float32))
    ff = np.log(np.array(toy["ff"], dtype=np.float32))
    for lab, v in g["forward_ctc_t1f32"].items():
        assert np.isclose(dec.cpp_forward(t1f, lab, "AB"), hexf(v), rtol=RTOL, atol=0), lab
    for lab, v in g["forward_ff_f32"].items():
        got, want = dec.cpp_forward(ff, lab, "AB", model_="ctc_flipflop"), hexf(v)
        assert (np.isneginf(got) and np.isneginf(want)) or np.isclose(got, want, rtol=RTOL, atol=0), lab


def test_forward_csv_and_batch(dec, oracle, golden, golden_inputs):
    from poreover_amd import batch
    y = np.log(golden_inputs["poreover_csv_prob"])
    assert np.isclose(dec.cpp_forward(y, golden["csv"]["viterbi"]), hexf(golden["csv"]["forward_viterbi"]), rtol=RTOL)
    ys, labs, models = [], [], []
    for i, (m, ffm) in enumerate([("ctc", False), ("ctc_merge_repeats", False), ("ctc_flipflop", True)]):
        for j in range(3):
            y1 = synth_pair(9000 + 10 * i + j, T=300 + 200 * j, flipflop=ffm)[0]
            lab = oracle.cpp_beam_search(y1, 5, model_=m)
            want = oracle.cpp_forward(y1, lab, model_=m)
            got = dec.cpp_forward(y1, lab, model_=m)
            assert np.isclose(got, want, rtol=RTOL, atol=0), (m, j)
            if m == "ctc":
                ys.append(y1); labs.append(lab)
    ys.append(ys[0]); labs.append("")                       # empty label: root->last_probability()
    ys.append(ys[0][:700]); labs.append("ACGT" * 100)       # label longer than one 256-lane chunk
    got = batch.forward_batch(ys, labs)
    for g_, y_, l_ in zip(got, ys, labs):
        w = oracle.cpp_forward(y_, l_)
        assert (np.isneginf(g_) and np.isneginf(w)) or np.isclose(g_, w, rtol=RTOL, atol=0)


def test_acceptor_golden_and_oracle(dec, oracle, golden, golden_inputs):
    y = np.log(golden_inputs["poreover_csv_prob"])
    seq = golden["csv"]["viterbi"]
    assert dec.cpp_viterbi_acceptor(y, seq).tolist() == golden["csv"]["acceptor_cpp"]
    for i in range(4):
        y1 = synth_pair(9100 + i, T=600 + 500 * i)[0]
        s, _ = oracle.viterbi_decode(y1)
        for band in (1000, 40, 7):
            try:
                want = oracle.cpp_viterbi_acceptor(y1, s, band)
            except oracle.OracleError as e:        # a band too narrow to reach the last label: upstream hangs
                with pytest.raises(Exception):
                    dec.cpp_viterbi_acceptor(y1, s, band)
                continue
            assert dec.cpp_viterbi_acceptor(y1, s, band).tolist() == want.tolist(), (i, band)
    # a beam-search label (not the Viterbi one) and a long label (> 256 positions)
    y1 = synth_pair(9200, T=4000)[0]
    lab = oracle.cpp_beam_search(y1, 10)
    assert len(lab) > 256
    assert dec.cpp_viterbi_acceptor(y1, lab).tolist() == oracle.cpp_viterbi_acceptor(y1, lab).tolist()


def test_acceptor_cython_twin(oracle, golden, golden_inputs):
    """decoding_cy.viterbi_acceptor: the reference's path on its csv fixture, and the oracle on synthetic reads with
    several band sizes (the twin's band expression and '>' tie rule are reproduced as written)"""
    from poreover_amd import _lib, batch
    from poreover_amd.decoding import decoding_cy
    y = np.log(golden_inputs["poreover_csv_prob"])
    seq = golden["csv"]["viterbi"]
    assert decoding_cy.viterbi_acceptor(y, seq).tolist() == golden["csv"]["acceptor_cy"]
    ys, labs = [], []
    for i in range(6):
        y1 = synth_pair(9900 + i, T=120 + 90 * i)[0]
        ys.append(y1); labs.append(oracle.viterbi_decode(y1)[0][: 5 + 9 * i] or "A")
    for band in (0, 7, 40, 1000):
        got = batch.viterbi_acceptor_batch(ys, labs, band, flavor="cy")
        for g, yk, lb in zip(got, ys, labs):
            assert g.tolist() == oracle.viterbi_acceptor(yk, lb, "ACGT", band).tolist(), band
    with pytest.raises(_lib.EngineError):
        batch.viterbi_acceptor_batch([ys[0]], ["ACXT"], 0, flavor="cy")
