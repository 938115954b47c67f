"""GPU: the decode / pair-decode DRIVERS end to end (files in, FASTA / log files out), compared with
what the reference's drivers produced for the same inputs (golden fixtures)."""
import argparse
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default", "reg", "legacy"])
def kernel_route(request, monkeypatch):
    """every test runs four times: with the engine's own choice of pair beam kernel (by model, width and batch size), with
    the two-pairs-per-wave kernel forced wherever it can run, with the LDS-ring kernel wherever it can run, and with
    beam2d_kernel always (_lib.set_pair_route)"""
    from poreover_amd import _lib
    _lib.set_pair_route({"reg": "reg", "legacy": "legacy"}.get(request.param, "auto"))
    yield request.param
    _lib.set_pair_route("auto")


@pytest.fixture(scope="module")
def eng():
    from poreover_amd import _lib
    _lib.load()
    return _lib


def _pair_args(**kw):
    d = dict(dir=".", basecaller="poreover", reverse_complement=False, out="out", threads=1, method="envelope",
             single="viterbi", logging="info", debug=False, algorithm="beam", alignment="banded", beam_width=5,
             debug_envelope=False, diagonal_envelope=False, diagonal_width=50, padding=5, skip_matches=False,
             skip_threshold=10, beam_search_method="row_col", window=200)
    d.update(kw)
    return argparse.Namespace(**d)


def test_decode_driver_files(eng, tmp_path, golden, golden_inputs):
    from poreover_amd.decoding import decode
    prob = golden_inputs["poreover_csv_prob"]
    csv = tmp_path / "poreover.csv"
    with open(csv, "w") as f:
        f.write("A,C,G,T,\n")
        np.savetxt(f, prob, delimiter=",", fmt="%.18e")
    g = golden["csv"]
    for algo, want in (("viterbi", g["viterbi"]), ("beam", g["beam_w25"])):
        a = argparse.Namespace(out=str(tmp_path / algo), basecaller=None, algorithm=algo, window=400, beam_width=25,
                               threads=1)
        setattr(a, "in", [str(csv)])
        decode.decode(a)
        txt = open(str(tmp_path / algo) + ".fasta").read()
        assert txt == decode.fasta_format("poreover", want) + "\n"      # print() adds the blank line
    # several files -> one batched call, records in input order
    for i in range(3):
        np.save(tmp_path / ("r%d.npy" % i), np.exp(golden_inputs["pair%d_y1" % i]))
    a = argparse.Namespace(out=str(tmp_path / "multi"), basecaller="poreover", algorithm="viterbi", window=400,
                           beam_width=25, threads=4)
    setattr(a, "in", [str(tmp_path / ("r%d.npy" % i)) for i in range(3)])
    decode.decode(a)
    recs = open(str(tmp_path / "multi") + ".fasta").read().split(">")[1:]
    assert [r.split("\n")[0] for r in recs] == ["r0", "r1", "r2"]
    a.algorithm, a.out = "prefix", str(tmp_path / "multi_prefix")      # windows of 400 frames per read
    decode.decode(a)
    assert len(open(a.out + ".fasta").read().split(">")[1:]) == 3


def test_pair_decode_driver_matches_reference_outputs(eng, tmp_path, monkeypatch, golden, golden_inputs):
    """pairs file -> {out}.1d.fasta / .2d.fasta / .log, byte-for-byte what the reference's callback writes
    (the golden run injected the same float64 matrices the loader is patched to return here)"""
    from poreover_amd.decoding import decode, pair_decode, transducer
    recs = [r for r in golden["pairs"] if r["kind"] == "poreover"][:6]
    mats = {}
    lines = []
    for r in recs:
        a, b = "p%d_a.npy" % r["index"], "p%d_b.npy" % r["index"]
        mats[a], mats[b] = golden_inputs["pair%d_y1" % r["index"]], golden_inputs["pair%d_y2" % r["index"]]
        lines.append(a + " " + b)
    pairs_file = tmp_path / "pairs.txt"
    pairs_file.write_text("\n".join(lines) + "\n")
    monkeypatch.setattr(decode, "model_from_trace",
                        lambda f, basecaller="": transducer.poreover(mats[os.path.basename(str(f))]))
    args = _pair_args(out=str(tmp_path / "run"))
    setattr(args, "in", [str(pairs_file)])
    pair_decode.pair_decode(args)
    want_1d, want_2d, want_log = "", "", []
    for r in recs:
        run = r["runs"]["row_col_w5_banded"]
        a, b = "p%d_a.npy" % r["index"], "p%d_b.npy" % r["index"]
        want_1d += run["fasta_1d"].replace("read_a.npy", a).replace("read_b.npy", b) + "\n"
        want_2d += run["fasta_2d"].replace("consensus;read_a;read_b", "consensus;%s;%s" % (a[:-4], b[:-4])) + "\n"
        sm = run["summary"]
        want_log.append("\t".join(map(str, [a, b, sm["length1"], sm["length2"],
                                            float.fromhex(sm["sequence_identity"]), sm["skipped"]])))
    assert open(str(tmp_path / "run") + ".1d.fasta").read() == want_1d
    assert open(str(tmp_path / "run") + ".2d.fasta").read() == want_2d
    log = open(str(tmp_path / "run") + ".log").read().splitlines()
    assert log[0] == "# PoreOver pair-decode" and log[2].split("\t")[-1] == "skipped"
    assert log[3:] == want_log
    # two positionals -> one pair -> {out}.fasta, same consensus
    a0 = lines[0].split()
    args2 = _pair_args(out=str(tmp_path / "single"))
    setattr(args2, "in", a0)
    res = pair_decode.pair_decode_helper(args2)
    assert len(res) == 3 and res[1].split("\n", 1)[1] == recs[0]["runs"]["row_col_w5_banded"]["fasta_2d"].split("\n", 1)[1]
    pair_decode.pair_decode(args2)
    assert open(str(tmp_path / "single") + ".fasta").read() == res[1] + "\n"
    # --diagonal_envelope: 2-tuple with the reference's header quirk (pair_decode.py:527)
    args3 = _pair_args(diagonal_envelope=True, diagonal_width=30)
    setattr(args3, "in", a0)
    out = pair_decode.pair_decode_helper(args3)
    assert len(out) == 2 and out[0].startswith(">consensus;envelope;p%d_a\n" % recs[0]["index"])
    assert out[0].split("\n", 1)[1] == recs[0]["runs"]["diag30"]["fasta_2d"].split("\n", 1)[1]
    # unsupported routes are refused loudly
    for kw in (dict(method="split"), dict(method="align"), dict(algorithm="prefix"),
               dict(skip_matches=True, diagonal_envelope=True)):
        bad = _pair_args(**kw)
        setattr(bad, "in", a0)
        with pytest.raises(eng.EngineError):
            pair_decode.pair_decode_helper(bad)


def test_transducer_classes(eng, golden, golden_inputs):
    from poreover_amd.decoding import transducer
    m = transducer.poreover(np.log(golden_inputs["poreover_csv_prob"]))
    seq, path = m.viterbi_decode(return_path=True)
    assert seq == golden["csv"]["viterbi"] and path.tolist() == golden["csv"]["viterbi_path"]
    assert m.argmax_decode() == seq
    assert transducer.poreover(np.array(golden["toy_prob"]["t1"]), "AB").viterbi_decode() == golden["toy"]["viterbi_t1"]
    for rec in golden["pairs"]:
        if rec["kind"] == "poreover":
            continue
        cls = {"bonito": transducer.bonito, "flipflop": transducer.flipflop}[rec["kind"]]
        s, p = cls(golden_inputs["pair%d_y1" % rec["index"]]).viterbi_decode(return_path=True)
        assert s == rec["viterbi1"] and p.tolist() == rec["path1"]


def test_pair_decode_single_beam(eng, monkeypatch, golden_inputs):
    """--single beam (pair_decode.py:363-370): 1-D basecalls by beam search W = 25, frame maps by the Viterbi
    acceptor (band 1000); checked stage by stage against the oracle"""
    from oracle import po_oracle as O
    from poreover_amd import batch
    y1, y2 = golden_inputs["pair3_y1"], golden_inputs["pair3_y2"]
    got = batch.pair_decode_batch([y1], [y2], single="beam")[0]
    b1, b2 = O.cpp_beam_search(y1, 25), O.cpp_beam_search(y2, 25)
    assert (got["seq1"], got["seq2"]) == (b1, b2)
    m1 = np.nonzero(O.cpp_viterbi_acceptor(y1, b1, 1000) < 4)[0]
    m2 = np.nonzero(O.cpp_viterbi_acceptor(y2, b2, 1000) < 4)[0]
    a1, a2 = O.global_pair_banded(b1, b2)
    env = O.build_envelope(len(y1), len(y2), a1, a2, m1, m2, 5)
    assert np.array_equal(got["envelope"], env)
    assert got["sequence_identity"] == sum(x == y for x, y in zip(a1, a2)) / len(a1)
    assert got["consensus"] == O.cpp_beam_search_2d(y1, y2, env, 5, method_="row_col")


def test_pair_decode_skip_matches(eng, tmp_path, monkeypatch, golden, golden_inputs):
    """--skip_matches (pair_decode.py:412-467,512-522): anchors copied, boxes decoded inside their envelope slice,
    pieces joined in signal order — the consensus records the reference's helper produced, for 8 pairs x 2
    thresholds, all boxes of all pairs in one batched GPU call"""
    from poreover_amd.decoding import decode, pair_decode, transducer
    recs = [r for r in golden["pairs"] if "skip10" in r["runs"]]
    assert len(recs) >= 8
    mats, in_paths = {}, []
    for r in recs:
        a, b = "p%d_a.npy" % r["index"], "p%d_b.npy" % r["index"]
        mats[a], mats[b] = golden_inputs["pair%d_y1" % r["index"]], golden_inputs["pair%d_y2" % r["index"]]
        in_paths.append([a, b])
    monkeypatch.setattr(decode, "model_from_trace",
                        lambda f, basecaller="": transducer.poreover(mats[os.path.basename(str(f))]))
    for thr in (10, 6):
        res = pair_decode.decode_pairs(in_paths, _pair_args(skip_matches=True, skip_threshold=thr))
        for r, got in zip(recs, res):
            want = r["runs"]["skip%d" % thr]
            assert want["n_out"] == 3 and len(got) == 3
            assert got[1].split("\n", 1)[1] == want["fasta_2d"].split("\n", 1)[1], (r["index"], thr)
    # get_anchors restated: the reference's own small cases
    assert pair_decode.get_anchors(("AAAAA-CC", "AAAAATCC"), matches=3, indels=1) == ([(0, 5), (5, 6)], ["mat", "ins"])
    assert pair_decode.get_anchors(("ACGT", "ACGT"), matches=2, indels=100) == ([], [])   # the open run at the end is not reported


def test_pair_decode_split_method(eng, monkeypatch, golden, golden_inputs):
    """--method split --diagonal_envelope (the only combination that returns upstream): boxes of --window frames
    along the diagonal, pair beam search without an envelope in each; the reference's consensus records"""
    from poreover_amd.decoding import decode, pair_decode, transducer
    recs = [r for r in golden["pairs"] if "split_beam_100" in r["runs"]]
    assert len(recs) >= 4
    mats, in_paths = {}, []
    for r in recs:
        a, b = "p%d_a.npy" % r["index"], "p%d_b.npy" % r["index"]
        mats[a], mats[b] = golden_inputs["pair%d_y1" % r["index"]], golden_inputs["pair%d_y2" % r["index"]]
        in_paths.append([a, b])
    monkeypatch.setattr(decode, "model_from_trace",
                        lambda f, basecaller="": transducer.poreover(mats[os.path.basename(str(f))]))
    res = pair_decode.decode_pairs(in_paths, _pair_args(method="split", window=100, diagonal_envelope=True))
    for r, got in zip(recs, res):
        want = r["runs"]["split_beam_100"]
        assert want["n_out"] == 2 and len(got) == 2
        assert got[0].startswith(">consensus;split;p%d_a\n" % r["index"])
        assert got[0].split("\n", 1)[1] == want["fasta_2d"].split("\n", 1)[1], r["index"]


def test_debug_routes(eng, tmp_path, golden, golden_inputs, capsys, monkeypatch):
    """--debug_envelope (pair_decode.py:503-507: band statistics line, pair reported as skipped) and --debug
    (pair_decode.py:482-490: debug.p) against what the reference's helper computed for the same pair"""
    import pickle
    from poreover_amd.decoding import pair_decode
    rec = [r for r in golden["pairs"] if r["kind"] == "poreover"][0]
    run = rec["runs"]["row_col_w5_banded"]
    for k in ("y1", "y2"):
        np.save(tmp_path / ("d_%s.npy" % k), np.exp(golden_inputs["pair%d_%s" % (rec["index"], k)]))
    monkeypatch.chdir(tmp_path)
    a = _pair_args(dir=str(tmp_path), out=str(tmp_path / "dbg"), debug_envelope=True)
    setattr(a, "in", ["d_y1.npy", "d_y2.npy"])
    out = pair_decode.pair_decode_helper(a)
    assert out == [{"skipped": 1}]
    fields = capsys.readouterr().out.strip().split()
    env = np.array(run["envelope"])
    size = env[:, 1] - env[:, 0]
    assert fields[:2] == ["d_y1", "d_y2"] and [int(x) for x in fields[2:5]] == [len(rec["viterbi1"]), len(rec["viterbi2"]), len(env)]
    assert np.allclose([float(x) for x in fields[6:]], [np.mean(size), np.std(size), np.median(size), np.min(size), np.max(size)])
    a = _pair_args(dir=str(tmp_path), out=str(tmp_path / "dbg2"), debug=True)
    setattr(a, "in", ["d_y1.npy", "d_y2.npy"])
    res = pair_decode.pair_decode_helper(a)
    assert len(res) == 3
    with open(tmp_path / "debug.p", "rb") as f:
        d = pickle.load(f)
    assert ["".join(d["alignment"][0]), "".join(d["alignment"][1])] == run["alignment"]
    assert d["sequence_to_signal1"] == rec["map1"] and d["sequence_to_signal2"] == rec["map2"]
    assert d["alignment_to_sequence"][0, -1] == len(rec["viterbi1"]) and d["alignment_to_sequence"][1, -1] == len(rec["viterbi2"])
