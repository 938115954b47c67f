"""GPU parity, pair path: cpp_beam_search_2d (method row_col) through the C-ABI vs the CPU oracle
and the golden vectors produced by the reference."""
import numpy as np
import pytest

from poreover_amd.synth import synth_pair

pytestmark = pytest.mark.gpu

MODEL_OF_KIND = {"poreover": "ctc", "bonito": "ctc_merge_repeats", "flipflop": "ctc_flipflop"}


@pytest.fixture(scope="module")
def eng():
    from poreover_amd import _lib, batch
    _lib.load()
    return batch


def _fasta_seq(txt):
    return "".join(txt.split("\n")[1:])


def test_rowcol_golden_pairs(eng, golden, golden_inputs):
    """envelopes and consensus strings captured from the reference's pair_decode_helper"""
    for rec in golden["pairs"]:
        y1 = golden_inputs["pair%d_y1" % rec["index"]]
        y2 = golden_inputs["pair%d_y2" % rec["index"]]
        for name, run in rec["runs"].items():
            if not name.startswith("row_col") or run["n_out"] != 3:
                continue
            W = int(name.split("_")[2][1:])
            got = eng.beam_search_2d_batch([y1], [y2], [np.array(run["envelope"])], W,
                                           model=MODEL_OF_KIND[rec["kind"]], method="row_col")[0]
            assert got == _fasta_seq(run["fasta_2d"]), (rec["index"], name)


def test_rowcol_csv_self_pair(eng, golden, golden_inputs):
    from poreover_amd.decoding import cpp_beam_search_2d
    y = np.log(golden_inputs["poreover_csv_prob"])
    T = len(y)
    env10 = np.array([(max(0, i - 10), min(i + 10, T)) for i in range(T)])
    assert cpp_beam_search_2d(y, y, env10.tolist(), beam_width_=10, method_="row_col") == golden["csv"]["self2d_row_col_w10"]
    assert cpp_beam_search_2d(y, y, env10.tolist(), beam_width_=5, method_="row_col") == golden["csv"]["self2d_row_col_w5"]


@pytest.mark.parametrize("model,ff", [("ctc", False), ("ctc_merge_repeats", False), ("ctc_flipflop", True)])
@pytest.mark.parametrize("W", [1, 2, 5, 10])
def test_rowcol_matches_oracle_batch(eng, oracle, model, ff, W):
    kind = {"ctc": "poreover", "ctc_merge_repeats": "bonito", "ctc_flipflop": "flipflop"}[model]
    y1s, y2s, envs, want = [], [], [], []
    for i in range(6):
        y1, y2 = synth_pair(4000 + i, T=300 + 40 * i, flipflop=ff)
        if i % 2 == 0:
            env = oracle.pair_decode(y1, y2, kind, 5, "row_col")["envelope"]   # pipeline envelope
        else:
            env = oracle.diagonal_envelope(len(y1), len(y2), 12 + i)
        y1s.append(y1); y2s.append(y2); envs.append(env)
        want.append(oracle.cpp_beam_search_2d(y1, y2, env, W, model_=model, method_="row_col"))
    got = eng.beam_search_2d_batch(y1s, y2s, envs, W, model=model, method="row_col")
    assert got == want


def test_rowcol_full_size(eng, oracle):
    """BASELINE config 3: pairs of T ~ 4000 reads, W = 5 (CLI default) and W = 10"""
    y1s, y2s, envs = [], [], []
    for i in range(6):
        y1, y2 = synth_pair(5000 + i, T=4000)
        y1s.append(y1); y2s.append(y2)
        envs.append(oracle.pair_decode(y1, y2, "poreover", 5, "row_col")["envelope"])
    for W in (5, 10):
        got = eng.beam_search_2d_batch(y1s, y2s, envs, W, method="row_col")
        want = [oracle.cpp_beam_search_2d(a, b, e, W, method_="row_col") for a, b, e in zip(y1s, y2s, envs)]
        assert got == want, W


def test_rowcol_errors(eng, oracle):
    from poreover_amd import _lib
    y1, y2 = synth_pair(1, T=120)
    U, V = len(y1), len(y2)
    bad = oracle.diagonal_envelope(U, V, 10)
    bad[:, 1] += V                      # columns beyond V: the reference overflows envelope_ranges_t
    seqs, st = eng.beam_search_2d_batch([y1], [y2], [bad], 5, method="row_col", return_status=True)
    assert st[0] == _lib.E_ENVELOPE
    # row 3 ends before the diagonal reaches it while column 3 is still covered: the reference runs
    # its main step with uninitialised row bounds (BeamSearch.h:309); oracle and engine both refuse
    gap = np.tile([0, V], (U, 1))
    gap[3] = (0, 1)
    with pytest.raises(oracle.OracleError) as ei:
        oracle.cpp_beam_search_2d(y1, y2, gap, 5, method_="row_col")
    assert ei.value.code == oracle.E_ENVELOPE
    seqs, st = eng.beam_search_2d_batch([y1], [y2], [gap], 5, method="row_col", return_status=True)
    assert st[0] == _lib.E_ENVELOPE
    # one bad pair does not disturb its neighbours in the batch
    good = oracle.diagonal_envelope(U, V, 10)
    seqs, st = eng.beam_search_2d_batch([y1, y1, y1], [y2, y2, y2], [good, gap, good], 5, method="row_col",
                                        return_status=True)
    assert st.tolist() == [0, _lib.E_ENVELOPE, 0]
    assert seqs[0] == seqs[2] == oracle.cpp_beam_search_2d(y1, y2, good, 5, method_="row_col")
