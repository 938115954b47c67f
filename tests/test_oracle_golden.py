"""Pin the CPU oracle (oracle/po_oracle.c) to golden vectors produced by RUNNING the
Python/Cython reference (tests/golden/make_golden.py).  Integer / string results must be
identical; floating-point results that go through numpy/scipy summation in the reference are
compared with np.isclose, as the reference's own tests do (tests/test_prefix.py)."""
import numpy as np
import pytest

from conftest import hexf

MODEL_OF_KIND = {"poreover": "ctc", "bonito": "ctc_merge_repeats", "flipflop": "ctc_flipflop"}


def test_toy_beam_and_forward(oracle, golden):
    toy, g = golden["toy_prob"], golden["toy"]
    t1, t2, t3 = (np.log(np.array(toy[k])) for k in ("t1", "t2", "t3"))
    ff = np.log(np.array(toy["ff"], dtype=np.float32))
    assert oracle.cpp_beam_search(t1, alphabet_="AB") == g["beam1d_t1"]
    assert oracle.cpp_beam_search(t2, alphabet_="AB") == g["beam1d_t2"]
    assert oracle.cpp_beam_search_2d(t1, t1, alphabet_="AB") == g["beam2d_same_t1"]
    assert oracle.cpp_beam_search_2d(t1, t3, alphabet_="AB") == g["beam2d_t1_t3"]
    # the reference's own test_flipflop_same FAILS upstream; parity target = its actual outputs
    assert oracle.cpp_beam_search(ff, alphabet_="AB", model_="ctc_flipflop") == g["ff_beam1d"]
    assert oracle.cpp_beam_search_2d(ff, ff, alphabet_="AB", method_="row", model_="ctc_flipflop") == g["ff_beam2d_row"]
    t1f = np.log(np.array(toy["t1"], dtype=np.float32))
    for lab, v in g["forward_ctc_t1f32"].items():
        assert oracle.cpp_forward(t1f, lab, "AB") == hexf(v)
    for lab, v in g["forward_ff_f32"].items():
        assert oracle.cpp_forward(ff, lab, "AB", model_="ctc_flipflop") == hexf(v)
    assert oracle.viterbi_decode(np.array(toy["t1"]), "poreover", "AB")[0] == g["viterbi_t1"]


def test_csv_fixture(oracle, golden, golden_inputs):
    g = golden["csv"]
    y = np.log(golden_inputs["poreover_csv_prob"])
    seq, path = oracle.viterbi_decode(y)
    assert seq == g["viterbi"] and path.tolist() == g["viterbi_path"]
    for W in (5, 10, 25):
        assert oracle.cpp_beam_search(y, W) == g["beam_w%d" % W]
    assert oracle.cpp_beam_search(y, 10, model_="ctc_merge_repeats") == g["beam_merge_w10"]
    assert oracle.cpp_forward(y, seq) == hexf(g["forward_viterbi"])
    T = len(y)
    env10 = np.array([(max(0, i - 10), min(i + 10, T)) for i in range(T)])
    assert oracle.cpp_beam_search_2d(y, y, env10, 10, method_="row") == g["self2d_row_w10"]
    assert oracle.cpp_beam_search_2d(y, y, env10, 10, method_="row_col") == g["self2d_row_col_w10"]
    assert oracle.cpp_beam_search_2d(y, y, env10, 5, method_="row_col") == g["self2d_row_col_w5"]
    diag = np.array([(i, i + 1) for i in range(T)])
    assert oracle.cpp_beam_search_2d(y, y, diag) == g["self2d_diag_w25"]
    assert oracle.cpp_viterbi_acceptor(y, seq).tolist() == g["acceptor_cpp"]
    assert oracle.viterbi_acceptor(y, seq).tolist() == g["acceptor_cy"]


def test_csv_prefix_search(oracle, golden, golden_inputs):
    g = golden["csv"]
    y = np.log(golden_inputs["poreover_csv_prob"])
    for flavor in ("cy", "py"):
        lab, lp = oracle.prefix_search_log(y[:100], flavor)
        assert lab == g["prefix_first100_%s" % flavor][0]
        assert np.isclose(lp, hexf(g["prefix_first100_%s" % flavor][1]), rtol=1e-12, atol=0)
    seq, i = "", 0
    while i + 400 < len(y):
        seq += oracle.prefix_search_log(y[i:i + 400], "cy")[0]
        i += 400
    seq += oracle.prefix_search_log(y[i:], "cy")[0]
    assert seq == g["decode_prefix_w400"]


def test_prefix_toys(oracle, golden):
    pm = golden["prefix_prob"]
    with np.errstate(divide="ignore"):
        for k, rec in golden["prefix_toy"].items():
            y = np.log(np.array(pm[k]))
            for flavor in ("py", "cy"):
                lab, lp = oracle.prefix_search_log(y, flavor)
                assert lab == rec[flavor][0]
                assert np.isclose(lp, hexf(rec[flavor][1]), rtol=1e-12)
        for k, rec in golden["pair_prefix_toy"].items():
            a, b = k.split("_")
            ya, yb = np.log(np.array(pm[a])), np.log(np.array(pm[b]))
            for flavor in ("py", "cy"):
                lab, lp = oracle.pair_prefix_search_log(ya, yb, flavor)
                assert lab == rec[flavor][0]
                assert np.isclose(lp, hexf(rec[flavor][1]), rtol=1e-9)
                assert np.isclose(oracle.pair_gamma_log(ya, yb, flavor)[0, 0], hexf(rec["gamma00_" + flavor]), rtol=1e-12)
            assert oracle.cpp_beam_search_2d(ya, yb, alphabet_="AB") == rec["beam2d"]


def test_prefix_windows_and_vectors(oracle, golden, golden_inputs):
    y = golden_inputs["prefix_y"]
    for k, rec in golden["prefix_windows"].items():
        lo, hi = map(int, k.split("_"))
        for flavor in ("cy", "py"):
            lab, lp = oracle.prefix_search_log(y[lo:hi], flavor)
            assert lab == rec[flavor][0]
            assert np.isclose(lp, hexf(rec[flavor][1]), rtol=1e-12)
    for flavor in ("cy", "py"):
        fw0 = oracle.forward_vec_log(-1, 0, y[:100], None, flavor)
        fw1 = oracle.forward_vec_log(2, 1, y[:100], fw0, flavor)
        fw2 = oracle.forward_vec_log(1, 2, y[:100], fw1, flavor)
        want = golden_inputs["fwvec_" + flavor]
        if flavor == "cy":  # same libm calls in the same order: bit-exact
            assert np.array_equal(np.stack([fw0, fw1, fw2]), want)
        else:
            assert np.allclose(np.stack([fw0, fw1, fw2]), want, rtol=1e-13, atol=0)
    ya, yb = y[:30], y[40:65]
    assert np.array_equal(oracle.pair_gamma_log(ya, yb, "cy"), golden_inputs["gamma_dense_cy"])
    assert np.allclose(oracle.pair_gamma_log(ya, yb, "py"), golden_inputs["gamma_dense_py"], rtol=1e-12, atol=0)
    lab, lp = oracle.pair_prefix_search_log(ya[:20], ya[:20], "cy")
    assert lab == golden["pair_prefix_synth"][0]
    assert np.isclose(lp, hexf(golden["pair_prefix_synth"][1]), rtol=1e-9)


def test_alignment(oracle, golden):
    for rec in golden["nw"]:
        a1, a2 = oracle.global_pair(rec["s1"], rec["s2"])
        assert ["".join(a1), "".join(a2)] == rec["full"]
        b1, b2 = oracle.global_pair_banded(rec["s1"], rec["s2"])
        assert ["".join(b1), "".join(b2)] == rec["banded"]
        if "banded30" in rec:
            c1, c2 = oracle.global_pair_banded(rec["s1"], rec["s2"], 30)
            assert ["".join(c1), "".join(c2)] == rec["banded30"]


def _fasta(name, seq, width=60):
    out = ">" + name + "\n"
    w = 0
    while w + width < len(seq):
        out += seq[w:w + width] + "\n"
        w += width
    return out + seq[w:] + "\n"


def _fasta_seq(txt):
    return "".join(txt.split("\n")[1:])


def test_pairs_end_to_end(oracle, golden, golden_inputs):
    for rec in golden["pairs"]:
        idx, kind = rec["index"], rec["kind"]
        y1, y2 = golden_inputs["pair%d_y1" % idx], golden_inputs["pair%d_y2" % idx]
        s1, p1 = oracle.viterbi_decode(y1, kind)
        s2, p2 = oracle.viterbi_decode(y2, kind)
        assert (s1, s2) == (rec["viterbi1"], rec["viterbi2"])
        assert p1.tolist() == rec["path1"] and p2.tolist() == rec["path2"]
        assert oracle.get_sequence_mapping(p1, kind).tolist() == rec["map1"]
        assert oracle.get_sequence_mapping(p2, kind).tolist() == rec["map2"]
        for W in (5, 10):
            assert oracle.cpp_beam_search(y1, W, model_=MODEL_OF_KIND[kind]) == rec["beam1d_w%d" % W]
        for name, run in rec["runs"].items():
            if name == "diag30":
                env = oracle.diagonal_envelope(len(y1), len(y2), 30)
                cons = oracle.cpp_beam_search_2d(y1, y2, env, 5, method_="row_col")
                assert cons == _fasta_seq(run["fasta_2d"])
                continue
            if name.startswith("skip") or name.startswith("split"):   # --skip_matches runs are host orchestration over these same stages
                continue
            method, W, aln = name.rsplit("_", 2)
            out = oracle.pair_decode(y1, y2, kind, int(W[1:]), method, 5, aln)
            sm = run["summary"]
            assert out["length1"] == sm["length1"] and out["length2"] == sm["length2"]
            assert out["skipped"] == sm["skipped"]
            if run["n_out"] != 3:
                continue
            assert out["sequence_identity"] == hexf(sm["sequence_identity"])
            assert out["envelope"].tolist() == run["envelope"]
            assert out["consensus"] == _fasta_seq(run["fasta_2d"]), (idx, name)
            assert run["fasta_1d"] == _fasta("read_a.npy", s1) + _fasta("read_b.npy", s2)


def test_revcomp_fixture_shape(golden_inputs):
    a, b = golden_inputs["revcomp_poreover_in"], golden_inputs["revcomp_poreover_out"]
    assert np.array_equal(a[::-1][:, [3, 2, 1, 0, 4]], b)


def test_grid_method_golden(oracle, golden, golden_grid, golden_inputs):
    """the hidden `grid` pair method (BeamSearch2.h:33-184) against outputs of the reference"""
    from conftest import grid_cases
    for name, y1, y2, env, W, model, alphabet, want in grid_cases(golden, golden_grid, golden_inputs):
        got = oracle.cpp_beam_search_2d(y1, y2, env, W, alphabet_=alphabet, model_=model, method_="grid")
        assert got == want, name
