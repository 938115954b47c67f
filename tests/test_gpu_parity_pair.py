"""GPU parity, whole pair-decode stage chain (pair_decode.pair_decode_helper default route) run on
the device through po_pair_decode_batch: 1-D basecalls, banded / full alignment, identity and length
skips, envelope, consensus — vs the golden outputs of the reference and vs the CPU oracle."""
import numpy as np
import pytest

from conftest import hexf
from poreover_amd.synth import synth_pair

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
    from poreover_amd import _lib, batch
    _lib.load()
    return batch


def _fasta_seq(txt):
    return "".join(txt.split("\n")[1:])


def test_pipeline_golden(eng, golden, golden_inputs):
    for rec in golden["pairs"]:
        y1 = golden_inputs["pair%d_y1" % rec["index"]]
        y2 = golden_inputs["pair%d_y2" % rec["index"]]
        for name, run in rec["runs"].items():
            if name == "diag30":
                out = eng.pair_decode_batch([y1], [y2], rec["kind"], 5, "row_col", diagonal_envelope=True,
                                            diagonal_width=30)[0]
                assert out["consensus"] == _fasta_seq(run["fasta_2d"])
                continue
            if name.startswith("skip") or name.startswith("split"):   # --skip_matches runs: tests/test_gpu_drivers.py
                continue
            method, W, aln = name.rsplit("_", 2)
            if method != "row_col":
                continue
            out = eng.pair_decode_batch([y1], [y2], rec["kind"], int(W[1:]), method, 5, aln)[0]
            sm = run["summary"]
            assert (out["seq1"], out["seq2"]) == (rec["viterbi1"], rec["viterbi2"])
            assert out["length1"] == sm["length1"] and out["length2"] == sm["length2"]
            assert out["skipped"] == sm["skipped"]
            if run["n_out"] != 3:
                continue
            assert out["sequence_identity"] == hexf(sm["sequence_identity"])
            assert out["envelope"].tolist() == run["envelope"], (rec["index"], name)
            assert out["consensus"] == _fasta_seq(run["fasta_2d"]), (rec["index"], name)


@pytest.mark.parametrize("kind", ["poreover", "bonito", "flipflop"])
def test_pipeline_matches_oracle_batch(eng, oracle, kind):
    y1s, y2s = [], []
    for i in range(10):
        y1, y2 = synth_pair(6000 + i, T=250 + 45 * i, flipflop=(kind == "flipflop"))
        y1s.append(y1); y2s.append(y2)
    got = eng.pair_decode_batch(y1s, y2s, kind, 5, "row_col")
    for i, (y1, y2) in enumerate(zip(y1s, y2s)):
        try:
            want = oracle.pair_decode(y1, y2, kind, 5, "row_col")
        except oracle.OracleError as e:      # e.g. the bonito frame-map assertion of the reference
            assert got[i]["status"] == e.code
            continue
        assert got[i]["status"] == want["status"], i
        assert (got[i]["seq1"], got[i]["seq2"]) == (want["seq1"], want["seq2"])
        if want["status"] == 0:
            assert np.array_equal(got[i]["envelope"], want["envelope"]), i
            assert got[i]["sequence_identity"] == want["sequence_identity"]
            assert got[i]["consensus"] == want["consensus"], i


def test_pipeline_full_alignment_and_skips(eng, oracle):
    y1, y2 = synth_pair(6100, T=500)
    got = eng.pair_decode_batch([y1], [y2], "poreover", 5, "row_col", alignment="full")[0]
    want = oracle.pair_decode(y1, y2, "poreover", 5, "row_col", alignment="full")
    assert got["consensus"] == want["consensus"] and np.array_equal(got["envelope"], want["envelope"])
    # reads with nothing in common -> identity < 0.5 -> skipped (pair_decode.py:395-398)
    from poreover_amd.synth import _render
    rng = np.random.default_rng(5)
    a = _render(rng, np.zeros(40, dtype=np.int64), 400, False)        # AAAA...
    b = _render(rng, np.ones(44, dtype=np.int64), 420, False)         # CCCC...
    got = eng.pair_decode_batch([a], [b])[0]
    want = oracle.pair_decode(a, b)
    assert want["skipped"] == 1 and want["status"] == oracle.SKIP_IDENTITY
    assert got["skipped"] == 1 and got["status"] == want["status"]
    assert got["sequence_identity"] == want["sequence_identity"] and got["consensus"] is None
    # length mismatch > 1000 bases -> skipped before alignment (pair_decode.py:372-375)
    long_, _ = synth_pair(6103, T=12000)
    got = eng.pair_decode_batch([long_], [a])[0]
    want = oracle.pair_decode(long_, a)
    assert want["status"] == oracle.SKIP_LENGTH and got["status"] == want["status"]
    assert (got["length1"], got["length2"]) == (want["length1"], want["length2"])
    # an empty basecall (a read of blanks only, in either position): the reference aligns it to gaps, identity 0.0,
    # and skips the pair like any other of low identity (found by scripts/fuzz_parity.py --pipeline)
    e1, e2 = synth_pair(947200141, T=58)
    e2 = e2[: max(2, len(e2) // 2)]
    assert oracle.viterbi_decode(e2)[0] == "" and oracle.viterbi_decode(e1)[0] != ""
    for p, q in ((e1, e2), (e2, e1)):
        got = eng.pair_decode_batch([p], [q])[0]
        want = oracle.pair_decode(p, q)
        assert want["status"] == oracle.SKIP_IDENTITY and got["status"] == want["status"]
        assert got["sequence_identity"] == 0.0 and (got["length1"], got["length2"]) == (want["length1"], want["length2"])


def test_pipeline_full_size(eng, oracle):
    """BASELINE config 3/4 shape: T ~ 4000 pairs, CLI defaults (W = 5, row_col, banded, padding 5)"""
    y1s, y2s = zip(*[synth_pair(7000 + i, T=4000) for i in range(8)])
    got = eng.pair_decode_batch(list(y1s), list(y2s))
    for i in range(8):
        want = oracle.pair_decode(y1s[i], y2s[i])
        assert got[i]["status"] == 0 and want["status"] >= 0
        assert np.array_equal(got[i]["envelope"], want["envelope"]), i
        assert got[i]["sequence_identity"] == want["sequence_identity"]
        assert got[i]["consensus"] == want["consensus"], i


def test_pipeline_long_reads(eng, oracle):
    """'maximum sizes': one pair of T ~ 60 000 frame reads (the size of the reference's real read1/2.npy,
    6 000+ bases each), so banded NW really is banded (|band| 500 < length), the label walk is long and the
    value-store ring has to grow with the wider look-ahead windows."""
    y1, y2 = synth_pair(9900, T=60000)
    got = eng.pair_decode_batch([y1], [y2])[0]
    want = oracle.pair_decode(y1, y2)
    assert want["status"] == 0 and got["status"] == 0
    assert got["length1"] > 6000 and (got["seq1"], got["seq2"]) == (want["seq1"], want["seq2"])
    assert got["sequence_identity"] == want["sequence_identity"]
    assert np.array_equal(got["envelope"], want["envelope"])
    assert got["consensus"] == want["consensus"]


def test_pipeline_grid_method(eng, oracle):
    """--beam_search_method grid (hidden upstream option) through the whole stage chain, small and full size"""
    y1s, y2s = zip(*([synth_pair(6100 + i, T=300 + 50 * i) for i in range(4)] + [synth_pair(7000, T=4000)]))
    got = eng.pair_decode_batch(list(y1s), list(y2s), "poreover", 5, "grid")
    for i in range(len(y1s)):
        want = oracle.pair_decode(y1s[i], y2s[i], "poreover", 5, "grid")
        assert got[i]["status"] == want["status"] == 0, i
        assert np.array_equal(got[i]["envelope"], want["envelope"]), i
        assert got[i]["consensus"] == want["consensus"], i


def _dense_read(rng, seq, T):
    """a read with a base every ~2.3 frames (Bonito's stride / fast flip-flop models): (T, 5) log-probabilities"""
    from poreover_amd.synth import log_softmax
    pos = np.sort(rng.choice(T, size=len(seq), replace=False))
    lab = np.full(T, 4, dtype=np.int64)
    lab[pos] = seq
    logits = rng.normal(0, 1, (T, 5)).astype(np.float32)
    logits[np.arange(T), lab] += 6.0
    return log_softmax(logits)


@pytest.mark.parametrize("kind", ["poreover", "bonito"])
def test_dense_basecalls_no_capacity_error(eng, oracle, kind):
    """0.35 - 0.5 bases per frame: more than the rows / 4 the first alignment pass budgets for — the second pass
    (slices for one base per frame) must decode them, in the same batch as ordinary pairs (PO_E_CAP)"""
    rng = np.random.default_rng(77)
    y1s, y2s = [], []
    for i, dens in enumerate((0.35, 0.45, 0.5, 0.1, 0.42)):
        T = 900 + 60 * i
        ref = rng.integers(4, size=int(T * dens))
        if kind == "bonito":       # no repeated bases next to each other: bonito collapses them
            ref = ref[np.insert(np.diff(ref) != 0, 0, True)]
        mut = ref.copy()
        flip = rng.random(len(mut)) < 0.05
        mut[flip] = (mut[flip] + 1) % 4
        if kind == "bonito":
            mut = mut[np.insert(np.diff(mut) != 0, 0, True)]
        y1s.append(_dense_read(rng, ref, T)); y2s.append(_dense_read(rng, mut, T + 37))
    got = eng.pair_decode_batch(y1s, y2s, kind, 5, "row_col")
    assert max(g["length1"] / len(y) for g, y in zip(got, y1s)) > 0.3
    for i, (a, b) in enumerate(zip(y1s, y2s)):
        try:
            want = oracle.pair_decode(a, b, kind, 5, "row_col")
        except oracle.OracleError as e:
            assert got[i]["status"] == e.code
            continue
        assert got[i]["status"] == want["status"], i
        assert (got[i]["seq1"], got[i]["seq2"]) == (want["seq1"], want["seq2"]), i
        if want["status"] == 0:
            assert np.array_equal(got[i]["envelope"], want["envelope"]), i
            assert got[i]["consensus"] == want["consensus"], i
