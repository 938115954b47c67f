"""CPU checks of the Python mirror of the reference's host layer (no GPU needed): FASTA formatting,
trace ingest, reverse complement, CLI flag surface."""
import numpy as np
import pytest


def test_fasta_format_golden(golden):
    from poreover_amd.decoding.decode import fasta_format
    g = golden["fasta_format"]
    assert fasta_format("r1", "ACGT" * 10) == g["short"]
    assert fasta_format("r2", "A" * 60) == g["exact60"]
    assert fasta_format("r3", "ACGTT" * 31) == g["long"]
    assert fasta_format("r4", "") == g["empty"]


def test_load_logits_golden(tmp_path, golden_inputs):
    """decode.load_logits: 3-D float32 logits -> log-softmax; 2-D probabilities -> log"""
    from poreover_amd.decoding.decode import load_logits
    np.save(tmp_path / "l.npy", golden_inputs["ingest_logits"])
    got = load_logits(str(tmp_path / "l.npy"), flatten=True)
    want = golden_inputs["ingest_logits_out"]
    assert got.shape == want.shape and got.dtype == want.dtype == np.float32   # float32 in, float32 out, as upstream
    # same scipy.special.logsumexp call as the reference; tolerance = 2 float32 ulp in case the scipy
    # build on the test box differs from the one that produced the fixture
    assert np.allclose(got, want, rtol=0, atol=5e-7)
    np.save(tmp_path / "p.npy", golden_inputs["ingest_prob"])
    got = load_logits(str(tmp_path / "p.npy"), flatten=True)
    assert np.array_equal(got, golden_inputs["ingest_prob_out"])   # np.log of the same array: identical


def test_model_from_trace_kinds(tmp_path, golden_inputs):
    from poreover_amd.decoding.decode import model_from_trace
    np.save(tmp_path / "l.npy", golden_inputs["ingest_logits"])
    m = model_from_trace(str(tmp_path / "l.npy"), "poreover")
    assert m.kind == "poreover" and m.log_prob.shape == (120, 5) and m.t_max == 120
    b = model_from_trace(str(tmp_path / "l.npy"), "bonito")          # blank first -> blank last
    assert b.kind == "bonito" and np.array_equal(b.log_prob, m.log_prob[:, [1, 2, 3, 4, 0]])
    prob = golden_inputs["poreover_csv_prob"]
    with open(tmp_path / "t.csv", "w") as f:
        f.write("A,C,G,T,\n")
        np.savetxt(f, prob, delimiter=",", fmt="%.18e")
    c = model_from_trace(str(tmp_path / "t.csv"))
    assert c.kind == "poreover" and np.array_equal(c.log_prob, np.log(prob))
    ff = np.hstack([prob[:, :4], prob[:, :4]]) / 2
    with open(tmp_path / "f.csv", "w") as f:
        f.write("A,C,G,T,a,c,g,t\n")
        np.savetxt(f, ff, delimiter=",", fmt="%.18e")
    assert model_from_trace(str(tmp_path / "f.csv")).kind == "flipflop"
    with pytest.raises(SystemExit):
        model_from_trace(str(tmp_path / "x.bin"))


def test_reverse_complement_golden(golden_inputs):
    from poreover_amd.decoding import transducer
    m = transducer.poreover(golden_inputs["revcomp_poreover_in"])
    m.reverse_complement()
    assert np.array_equal(m.log_prob, golden_inputs["revcomp_poreover_out"])
    y8 = np.arange(24, dtype=np.float64).reshape(3, 8)
    f = transducer.flipflop(y8)
    f.reverse_complement()
    assert np.array_equal(f.log_prob, y8[::-1, [3, 2, 1, 0, 7, 6, 5, 4]])
    assert transducer.remove_repeated("AAaACC") == "AaAC"


def test_cli_flag_surface():
    """the decode / pair-decode flags and defaults of the reference (__main__.py:52-91)"""
    from poreover_amd.__main__ import build_parser
    p = build_parser()
    d = vars(p.parse_args(["decode", "a.npy", "b.npy", "--basecaller", "bonito"]))
    assert d["in"] == ["a.npy", "b.npy"] and d["out"] == "out" and d["algorithm"] == "viterbi"
    assert d["beam_width"] == 25 and d["window"] == 400 and d["threads"] == 1
    q = vars(p.parse_args(["pair-decode", "pairs.txt"]))
    want = dict(dir=".", basecaller=None, reverse_complement=False, out="out", threads=1, method="envelope",
                single="viterbi", logging="info", debug=False, algorithm="beam", alignment="banded", beam_width=5,
                debug_envelope=False, diagonal_envelope=False, diagonal_width=50, padding=5, skip_matches=False,
                skip_threshold=10, beam_search_method="row_col", window=200)
    for k, v in want.items():
        assert q[k] == v, k
    with pytest.raises(SystemExit):
        p.parse_args(["pair-decode", "x", "--beam_search_method", "diagonal"])


def test_get_anchors_host_logic():
    """pair_decode.get_anchors (reference pair_decode.py:53-89) restated on the host: hand-checked cases"""
    from poreover_amd.decoding import pair_decode
    ga = pair_decode.get_anchors
    assert ga(("AAAAA-CC", "AAAAATCC"), matches=3, indels=1) == ([(0, 5), (5, 6)], ["mat", "ins"])
    assert ga(("ACGT", "ACGT"), matches=2, indels=100) == ([], [])            # the run still open at the end is dropped
    assert ga(("ACGTA", "ACGTC"), matches=4, indels=100) == ([(0, 4)], ["mat"])  # a mismatch closes the run
    assert ga(("AC--GT", "ACTTGT"), matches=2, indels=2) == ([(0, 2), (2, 4)], ["mat", "ins"])
    assert ga(("ACTTGT", "AC--GT"), matches=5, indels=2) == ([(2, 4)], ["del"])
    assert ga(("ACAC", "AGAG"), matches=1, indels=1) == ([(0, 1), (2, 3)], ["mat", "mat"])  # mismatches never count
