"""GPU parity of the 1-D prefix search (prefix_search_log_cy) and of `decode --algorithm prefix`
vs the reference's golden values and the oracle."""
import argparse
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, hexf
from poreover_amd.synth import synth_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ps():
    from poreover_amd import _lib
    from poreover_amd.decoding import prefix_search
    _lib.load()
    return prefix_search


def test_prefix_golden_toys(ps, golden):
    from collections import OrderedDict
    toy_alpha = OrderedDict([("A", 0), ("B", 1)])
    with np.errstate(divide="ignore"):
        for k, rec in golden["prefix_toy"].items():
            y = np.log(np.array(golden["prefix_prob"][k]))
            lab, lp = ps.prefix_search_log_cy(y, alphabet=toy_alpha)
            assert lab == rec["cy"][0]
            assert np.isclose(lp, hexf(rec["cy"][1]), rtol=1e-10, atol=0)


def test_prefix_golden_windows_and_decode(ps, tmp_path, golden, golden_inputs):
    y = golden_inputs["prefix_y"]
    for k, rec in golden["prefix_windows"].items():
        lo, hi = map(int, k.split("_"))
        lab, lp = ps.prefix_search_log_cy(y[lo:hi])
        assert lab == rec["cy"][0], k
        assert np.isclose(lp, hexf(rec["cy"][1]), rtol=1e-10, atol=0)
    yc = np.log(golden_inputs["poreover_csv_prob"])
    lab, lp = ps.prefix_search_log_cy(yc[:100])
    assert lab == golden["csv"]["prefix_first100_cy"][0]
    assert np.isclose(lp, hexf(golden["csv"]["prefix_first100_cy"][1]), rtol=1e-10)
    # decode --algorithm prefix --window 400 on the reference's csv fixture (decode.py:179-188)
    from poreover_amd.decoding import decode
    csv = tmp_path / "poreover.csv"
    with open(csv, "w") as f:
        f.write("A,C,G,T,\n")
        np.savetxt(f, golden_inputs["poreover_csv_prob"], delimiter=",", fmt="%.18e")
    a = argparse.Namespace(out=str(tmp_path / "pfx"), basecaller=None, algorithm="prefix", window=400, beam_width=25,
                           threads=1)
    setattr(a, "in", [str(csv)])
    decode.decode(a)
    assert open(str(tmp_path / "pfx") + ".fasta").read() == decode.fasta_format("poreover", golden["csv"]["decode_prefix_w400"]) + "\n"


def test_prefix_matches_oracle_batch(ps, oracle):
    from poreover_amd import batch
    y = synth_pair(9500, T=2000)[0]
    offs = [0, 400, 800, 1200, 1601, 1999, 2000]     # ragged windows incl. a single-frame one
    got = batch.prefix_search_batch(y, offs)
    for (lab, lp), lo, hi in zip(got, offs[:-1], offs[1:]):
        wl, wp = oracle.prefix_search_log(y[lo:hi], "cy")
        assert lab == wl, (lo, hi)
        assert np.isclose(lp, wp, rtol=1e-10, atol=0)
    assert ps.greedy_search(y[:300]) == oracle.viterbi_decode(y[:300])[0]


def test_pair_prefix_search(ps, oracle, golden, golden_inputs):
    """pair_prefix_search_log / _cy (prefix_search.py:247-385): the reference's toy pairs (test_prefix.py:106-162),
    a synthetic box, and a batch of random boxes against the oracle"""
    from collections import OrderedDict
    from poreover_amd import batch
    toy_alpha = OrderedDict([("A", 0), ("B", 1)])
    pm = golden["prefix_prob"]
    with np.errstate(divide="ignore"):
        for k, rec in golden["pair_prefix_toy"].items():
            a, b = k.split("_")
            ya, yb = np.log(np.array(pm[a])), np.log(np.array(pm[b]))
            for flavor, fn in (("py", ps.pair_prefix_search_log), ("cy", ps.pair_prefix_search_log_cy)):
                lab, lp = fn(ya, yb, alphabet=toy_alpha)
                assert lab == rec[flavor][0], (k, flavor)
                assert np.isclose(lp, hexf(rec[flavor][1]), rtol=1e-9)
    y = golden_inputs["prefix_y"]
    lab, lp = ps.pair_prefix_search_log_cy(y[:20], y[:20])
    assert lab == golden["pair_prefix_synth"][0]
    assert np.isclose(lp, hexf(golden["pair_prefix_synth"][1]), rtol=1e-9)
    b1, b2, want = [], [], {"py": [], "cy": []}
    for i in range(8):
        y1, y2 = synth_pair(9700 + i, T=60 + 25 * i)[:2]
        lo = 7 * i
        y1, y2 = y1[lo:lo + 30 + 12 * i], y2[lo:lo + 25 + 14 * i]     # ragged boxes
        b1.append(y1); b2.append(y2)
        for f in ("py", "cy"):
            want[f].append(oracle.pair_prefix_search_log(y1, y2, f))
    for f in ("py", "cy"):
        got = batch.pair_prefix_search_batch(b1, b2, flavor=f)
        assert [g[0] for g in got] == [w[0] for w in want[f]], f
        assert np.allclose([g[1] for g in got], [w[1] for w in want[f]], rtol=1e-9, atol=0)


def test_forward_vec_log(oracle, golden_inputs):
    """decoding_cy.forward_vec_log / prefix_search.forward_vec_log: the reference's rows (tests/golden) and the oracle"""
    from poreover_amd import batch
    from poreover_amd.decoding import decoding_cy
    y = golden_inputs["prefix_y"][:100]
    for flavor in ("cy", "py"):
        fw0 = batch.forward_vec_batch([y], -1, 0, None, flavor)[0]
        fw1 = batch.forward_vec_batch([y], 2, 1, [fw0], flavor)[0]
        fw2 = batch.forward_vec_batch([y], 1, 2, [fw1], flavor)[0]
        assert np.allclose(np.stack([fw0, fw1, fw2]), golden_inputs["fwvec_" + flavor], rtol=1e-13, atol=0)
    assert np.allclose(decoding_cy.forward_vec_log(2, 1, y, decoding_cy.forward_vec_log(-1, 0, y)),
                       golden_inputs["fwvec_cy"][1], rtol=1e-13, atol=0)
    ys = [synth_pair(9800 + k, T=50 + 37 * k)[0] for k in range(5)]          # ragged batch vs the oracle
    p0 = batch.forward_vec_batch(ys, -1, 0, None, "cy")
    p1 = batch.forward_vec_batch(ys, 3, 1, p0, "cy")
    for k, yk in enumerate(ys):
        w0 = oracle.forward_vec_log(-1, 0, yk, None, "cy")
        assert np.allclose(p0[k], w0, rtol=1e-13, atol=0)
        assert np.allclose(p1[k], oracle.forward_vec_log(3, 1, yk, w0, "cy"), rtol=1e-13, atol=0)


@pytest.fixture(scope="module")
def eng():
    from poreover_amd import _lib, batch
    _lib.load()
    return batch


@pytest.fixture(scope="module")
def oracle_(oracle):
    return oracle


# ---------------------------------------------------------------------------------------------------
# round 2: pair prefix search WITH an envelope (PairPrefixSearch.cpp:79-229 made to work), return_forward, and the
# remaining decoding_cy / align signatures
def _band(U, V, w):
    return np.array([(max(0, int(u * V / U) - w), min(V, int(u * V / U) + w)) for u in range(U + 1)])


@pytest.mark.parametrize("flavor", ["py", "cy"])
def test_pair_prefix_search_with_envelope_vs_oracle(eng, oracle, flavor):
    rng = np.random.default_rng(17)
    a1, a2, envs = [], [], []
    for i in range(10):
        y1, y2 = synth_pair(7100 + i, T=int(rng.integers(18, 70)))
        a1.append(y1); a2.append(y2); envs.append(_band(len(y1), len(y2), int(rng.integers(5, 14))))
    got = eng.pair_prefix_search_batch(a1, a2, "ACGT", flavor, envs)
    for i in range(len(a1)):
        lab, lp = oracle.pair_prefix_search_log(a1[i], a2[i], flavor, envs[i])
        assert got[i][0] == lab, i
        assert np.isclose(got[i][1], lp, rtol=1e-10, atol=1e-12), i
    # an envelope that covers every cell is the dense search
    full = [np.array([(0, len(b))] * (len(a) + 1)) for a, b in zip(a1, a2)]
    dense = eng.pair_prefix_search_batch(a1, a2, "ACGT", "py")
    for g, d in zip(eng.pair_prefix_search_batch(a1, a2, "ACGT", "py", full), dense):
        assert g[0] == d[0] and np.isclose(g[1], d[1], rtol=1e-10)


def test_pair_prefix_search_envelope_reference_toys(eng, golden):
    """the reference's toy pairs (tests/test_prefix.py:106-162) through the envelope entry points with the whole box
    as envelope: the labels and probabilities of its dense Python path"""
    from poreover_amd.decoding import decoding_cpp, prefix_search
    from collections import OrderedDict
    toy_alpha = OrderedDict([("A", 0), ("B", 1)])
    with np.errstate(divide="ignore"):
        for key, rec in golden["pair_prefix_toy"].items():
            a, b = key.split("_")
            ya, yb = np.log(np.array(golden["prefix_prob"][a])), np.log(np.array(golden["prefix_prob"][b]))
            full = np.array([(0, len(yb))] * (len(ya) + 1))
            lab, lp = prefix_search.pair_prefix_search_log(ya, yb, toy_alpha, envelope_ranges=full)
            assert lab == rec["py"][0] and np.isclose(lp, hexf(rec["py"][1]), rtol=1e-9)
            assert decoding_cpp.cpp_pair_prefix_search_log(ya, yb, full, "AB") == rec["py"][0]


def test_prefix_search_return_forward(eng):
    import json, os
    from conftest import GOLDEN_DIR
    from poreover_amd.decoding import prefix_search
    with open(os.path.join(GOLDEN_DIR, "extra_golden.json")) as f:
        ex = json.load(f)
    y1, _ = synth_pair(40, T=400)
    for key, rec in ex["prefix_return_forward"].items():
        lo, hi = (int(x) for x in key.split("_"))
        lab, mat = prefix_search.prefix_search_log_cy(y1[lo:hi], return_forward=True)
        want = np.array([[hexf(x) for x in row] for row in rec["matrix"]])
        assert lab == rec["label"] and list(mat.shape) == rec["shape"]
        assert np.allclose(mat, want, rtol=1e-12, atol=0)


def test_alignment_with_score_arguments(eng):
    import json, os
    from conftest import GOLDEN_DIR
    from poreover_amd.align import align
    with open(os.path.join(GOLDEN_DIR, "extra_golden.json")) as f:
        ex = json.load(f)
    for c in ex["align_scores"]:
        m, mm, g = c["scores"]
        f1, f2, _ = align.global_pair(c["s1"], c["s2"], m, mm, g)
        assert ["".join(f1), "".join(f2)] == c["full"], c["scores"]
        b1, b2 = align.global_pair_banded(c["s1"], c["s2"], 25, m, mm, g)
        assert ["".join(b1), "".join(b2)] == c["banded25"], c["scores"]


def test_decoding_cy_pair_gamma_log_envelope(eng):
    """decoding_cy.pyx:224-271 on small banded pairs vs the values the reference's own Cython build returned for the
    same inputs (tests/golden/make_golden_extra.py runs it; round 2 compared with a restatement written in this test)"""
    from poreover_amd.decoding import decoding_cy as cy
    with open(os.path.join(GOLDEN_DIR, "extra_golden.json")) as f:
        cases = json.load(f)["pair_gamma_cy_envelope"]
    assert len(cases) >= 3
    for c in cases:
        y1, y2 = synth_pair(c["seed"], T=40)
        y1, y2 = y1[:c["U"]], y2[:c["V"]]

        def fresh():
            m = cy.PySparseMatrix()
            for s, e in c["rows"]:
                m.push_row(s, e)
            return m
        got = cy.pair_gamma_log_envelope(y1, y2, fresh(), np.array(c["cells"], dtype=np.intp), fresh(), fresh())
        none = cy.pair_gamma_log_envelope(y1, y2, fresh(), None, fresh(), fresh())
        for (u, v), w in zip(c["cells"], c["gamma"]):
            a = got.get(u, v)
            w = hexf(w)
            assert (np.isneginf(a) and np.isneginf(w)) or np.isclose(a, w, rtol=1e-12), (c["seed"], u, v, a, w)
            b = none.get(u, v)
            assert (np.isneginf(a) and np.isneginf(b)) or a == b, (c["seed"], u, v)
