"""GPU parity, pair path: cpp_beam_search_2d (method row_col) through the C-ABI vs the CPU oracle
and the golden vectors produced by the reference."""
import numpy as np
import pytest

from poreover_amd.synth import synth_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default", "reg", "legacy"])
def kernel_route(request, monkeypatch):
    """every test runs four times: with the engine's own choice of pair beam kernel (by model, width and batch size), with
    the two-pairs-per-wave kernel forced wherever it can run, with the LDS-ring kernel wherever it can run, and with
    beam2d_kernel always (_lib.set_pair_route)"""
    from poreover_amd import _lib
    _lib.set_pair_route({"reg": "reg", "legacy": "legacy"}.get(request.param, "auto"))
    _ROUTE[0] = request.param
    yield request.param
    _lib.set_pair_route("auto")
    _ROUTE[0] = "default"


_ROUTE = ["default"]


def kernel_route_name():
    return _ROUTE[0]

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


def test_rowcol_two_kernel_paths_in_one_batch(eng, oracle, monkeypatch):
    """row_col with an envelope runs beam2d_reg_kernel; pairs that kernel hands on are decoded by beam2d_kernel in the same
    call.  set_pair_route(defer_odd=True) sends every odd pair down the second path: one batch, both kernels, every string
    equal to the oracle's."""
    from poreover_amd import _lib
    route = {"reg": "reg", "legacy": "legacy"}.get(kernel_route_name(), "auto")
    _lib.set_pair_route(route, defer_odd=True)
    y1s, y2s, envs, want = [], [], [], []
    for i in range(9):
        y1, y2 = synth_pair(4200 + i, T=250 + 60 * i)
        env = oracle.pair_decode(y1, y2, "poreover", 5, "row_col")["envelope"]
        y1s.append(y1); y2s.append(y2); envs.append(env)
        want.append(oracle.cpp_beam_search_2d(y1, y2, env, 5, method_="row_col"))
    assert eng.beam_search_2d_batch(y1s, y2s, envs, 5, method="row_col") == want
    _lib.set_pair_route(route, defer_odd=False)
    assert eng.beam_search_2d_batch(y1s, y2s, envs, 5, method="row_col") == want
    # wide windows: fewer row groups fit the store; a pair whose groups run out there is deferred too
    y1, y2 = synth_pair(4300, T=700)
    env = oracle.diagonal_envelope(len(y1), len(y2), 100)
    assert eng.beam_search_2d_batch([y1], [y2], [env], 3, method="row_col") == \
        [oracle.cpp_beam_search_2d(y1, y2, env, 3, method_="row_col")]


@pytest.mark.parametrize("model,ff", [("ctc", False), ("ctc_merge_repeats", False), ("ctc_flipflop", True)])
def test_rowcol_wide_beam_one_pair_per_wave(eng, oracle, monkeypatch, model, ff):
    """7 <= W <= 12 runs beam2d_reg_kernel's 64-slot layout (lane = element slot, the reads one after the other; route
    "legacy" forces beam2d_kernel): same strings as the oracle from both"""
    kind = {"ctc": "poreover", "ctc_merge_repeats": "bonito", "ctc_flipflop": "flipflop"}[model]
    y1s, y2s, envs = [], [], []
    for i in range(5):
        y1, y2 = synth_pair(4400 + i, T=280 + 70 * i, flipflop=ff)
        y1s.append(y1); y2s.append(y2)
        envs.append(oracle.pair_decode(y1, y2, kind, 5, "row_col")["envelope"])
    for W in (7, 10, 12):
        want = [oracle.cpp_beam_search_2d(a, b, e, W, model_=model, method_="row_col") for a, b, e in zip(y1s, y2s, envs)]
        assert eng.beam_search_2d_batch(y1s, y2s, envs, W, model=model, method="row_col") == want
        from poreover_amd import _lib
        _lib.set_pair_route("legacy")
        try:
            assert eng.beam_search_2d_batch(y1s, y2s, envs, W, model=model, method="row_col") == want, W
        finally:
            _lib.set_pair_route({"reg": "reg", "legacy": "legacy"}.get(kernel_route_name(), "auto"))


def test_rowcol_full_size(eng, oracle):
    """BASELINE config 3: pairs of T ~ 4000 reads, W = 5 (CLI default) and W = 10"""
    y1s, y2s, envs = [], [], []
    for i in range(6):
        y1, y2 = synth_pair(5000 + i, T=4000)
        y1s.append(y1); y2s.append(y2)
        envs.append(oracle.pair_decode(y1, y2, "poreover", 5, "row_col")["envelope"])
    for W, npair in ((5, 6), (10, 6), (25, 2)):   # W = 25: up to 25 new row groups per step
        got = eng.beam_search_2d_batch(y1s[:npair], y2s[:npair], envs[:npair], W, method="row_col")
        want = [oracle.cpp_beam_search_2d(a, b, e, W, method_="row_col")
                for a, b, e in zip(y1s[:npair], y2s[:npair], envs[:npair])]
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


def _jagged_envelope(rng, U, V, kind):
    """irregular but usable envelopes: the diagonal walk then meets long catch-up runs, windows that jump past the
    previous step's window end, single-cell rows, and rows much wider than their neighbours"""
    env = np.zeros((U, 2), dtype=np.int64)
    c = np.linspace(0, V - 1, U)
    for u in range(U):
        if kind == "stairs":        # flat runs then jumps: many catch-ups on one read, then on the other
            lo = int(c[(u // 7) * 7]) - 2
            hi = lo + 7 + 2 * V // U
        elif kind == "wobble":      # width changes from row to row, starts move back and forth
            w = int(rng.integers(2, 14))
            lo = int(c[u]) - int(rng.integers(0, w))
            hi = lo + w + int(rng.integers(1, 6))
        else:                       # "bursts": mostly narrow, every now and then a very wide row
            w = 40 if rng.random() < 0.08 else int(rng.integers(3, 7))
            lo = int(c[u]) - w // 2
            hi = lo + w
        lo, hi = max(0, min(lo, V - 1)), max(1, min(hi, V))
        if u > 0 and kind != "wobble":   # connected staircase: starts and ends never move back, rows overlap
            lo = min(max(lo, env[u - 1, 0]), env[u - 1, 1] - 1)
            hi = max(hi, env[u - 1, 1])
        if lo >= hi:
            lo = hi - 1
        env[u] = (lo, hi)
    if kind != "wobble":
        env[-1, 1] = V
    return env


@pytest.mark.parametrize("model,ff", [("ctc", False), ("ctc_merge_repeats", False), ("ctc_flipflop", True)])
def test_rowcol_irregular_envelopes(eng, oracle, model, ff):
    """schedule kernel, skipped catch-ups and skipped stores against envelopes that are nothing like the
    pipeline's: strings — or the refusal, when the reference would read uninitialised bounds — equal the oracle's"""
    from poreover_amd import _lib
    rng = np.random.default_rng(321)
    y1s, y2s, envs = [], [], []
    for i in range(18):
        T = int(rng.integers(3, 260))
        y1, y2 = synth_pair(4600 + i, T=T, flipflop=ff)
        if i % 5 == 0:
            y2 = y2[: max(2, len(y2) // 2)]          # very unequal lengths
        y1s.append(y1); y2s.append(y2)
        envs.append(_jagged_envelope(rng, len(y1), len(y2), ("stairs", "wobble", "bursts")[i % 3]))
    for W in (1, 3, 5, 6, 9):
        want, wst = [], []
        for a, b, e in zip(y1s, y2s, envs):
            try:
                want.append(oracle.cpp_beam_search_2d(a, b, e, W, model_=model, method_="row_col")); wst.append(0)
            except oracle.OracleError as ex:
                want.append(""); wst.append(ex.code)
        got, st = eng.beam_search_2d_batch(y1s, y2s, envs, W, model=model, method="row_col", return_status=True)
        assert st.tolist() == wst, (W, st.tolist(), wst)
        assert [g if c == 0 else "" for g, c in zip(got, wst)] == want, W
    assert any(c == 0 for c in wst) and len(set(wst)) >= 1


@pytest.mark.parametrize("seed", [11, 12])
def test_rowcol_irregular_envelopes_long(eng, oracle, seed):
    """the same at lengths where the incremental / steady step machinery has room to run (hundreds of main steps
    per pair, windows that shrink, jump and stall)"""
    rng = np.random.default_rng(seed)
    y1s, y2s, envs = [], [], []
    for i in range(24):
        T = int(rng.integers(300, 1200))
        y1, y2 = synth_pair(5200 + 100 * seed + i, T=T)
        if i % 6 == 0:
            y2 = y2[: max(2, (2 * len(y2)) // 3)]
        y1s.append(y1); y2s.append(y2)
        envs.append(_jagged_envelope(rng, len(y1), len(y2), ("stairs", "wobble", "bursts")[i % 3]))
    for W in (4, 5):
        want, wst = [], []
        for a, b, e in zip(y1s, y2s, envs):
            try:
                want.append(oracle.cpp_beam_search_2d(a, b, e, W, method_="row_col")); wst.append(0)
            except oracle.OracleError as ex:
                want.append(""); wst.append(ex.code)
        got, st = eng.beam_search_2d_batch(y1s, y2s, envs, W, method="row_col", return_status=True)
        assert st.tolist() == wst, (W, st.tolist(), wst)
        assert [g if c == 0 else "" for g, c in zip(got, wst)] == want, W
    assert sum(c == 0 for c in wst) >= 4


# ------------------------------------------------------------------------------------------------
# method "row" (the API default) and the no-envelope overload
def test_row_golden_toys_and_csv(eng, golden, golden_inputs):
    """the reference's own 2-D tests (tests/test_beam.py:25-105), with its actual outputs"""
    from poreover_amd.decoding import cpp_beam_search_2d
    toy, g = golden["toy_prob"], golden["toy"]
    t1, t3 = np.log(np.array(toy["t1"])), np.log(np.array(toy["t3"]))
    ff = np.log(np.array(toy["ff"], dtype=np.float32))
    assert cpp_beam_search_2d(t1, t1, alphabet_="AB") == g["beam2d_same_t1"]          # no envelope, W = 25
    assert cpp_beam_search_2d(t1, t3, alphabet_="AB") == g["beam2d_t1_t3"]
    assert cpp_beam_search_2d(ff, ff, alphabet_="AB", method_="row", model_="ctc_flipflop") == g["ff_beam2d_row"]
    pm = golden["prefix_prob"]
    with np.errstate(divide="ignore"):
        for k, rec in golden["pair_prefix_toy"].items():
            a, b = k.split("_")
            assert cpp_beam_search_2d(np.log(np.array(pm[a])), np.log(np.array(pm[b])), alphabet_="AB") == rec["beam2d"]
    y = np.log(golden_inputs["poreover_csv_prob"])
    T = len(y)
    env10 = np.array([(max(0, i - 10), min(i + 10, T)) for i in range(T)])
    assert cpp_beam_search_2d(y, y, env10.tolist(), beam_width_=10, method_="row") == golden["csv"]["self2d_row_w10"]
    diag = np.array([(i, i + 1) for i in range(T)])
    assert cpp_beam_search_2d(y, y, diag.tolist()) == golden["csv"]["self2d_diag_w25"]  # W = 25, 1-cell band


def test_row_golden_pairs(eng, golden, golden_inputs):
    for rec in golden["pairs"]:
        y1 = golden_inputs["pair%d_y1" % rec["index"]]
        y2 = golden_inputs["pair%d_y2" % rec["index"]]
        run = rec["runs"].get("row_w5_banded")
        if not run or run["n_out"] != 3:
            continue
        got = eng.beam_search_2d_batch([y1], [y2], [np.array(run["envelope"])], 5,
                                       model=MODEL_OF_KIND[rec["kind"]], method="row")[0]
        assert got == _fasta_seq(run["fasta_2d"]), rec["index"]


@pytest.mark.parametrize("model,ff", [("ctc", False), ("ctc_merge_repeats", False), ("ctc_flipflop", True)])
@pytest.mark.parametrize("W", [1, 3, 5, 10, 25])
def test_row_matches_oracle_batch(eng, oracle, model, ff, W):
    kind = {"ctc": "poreover", "ctc_merge_repeats": "bonito", "ctc_flipflop": "flipflop"}[model]
    y1s, y2s, envs, want = [], [], [], []
    for i in range(4):
        y1, y2 = synth_pair(8000 + i, T=260 + 50 * i, flipflop=ff)
        env = (oracle.pair_decode(y1, y2, kind, 5, "row")["envelope"] if i % 2 == 0
               else oracle.diagonal_envelope(len(y1), len(y2), 10 + i))
        y1s.append(y1); y2s.append(y2); envs.append(env)
        want.append(oracle.cpp_beam_search_2d(y1, y2, env, W, model_=model, method_="row"))
    assert eng.beam_search_2d_batch(y1s, y2s, envs, W, model=model, method="row") == want


@pytest.mark.parametrize("model,ff", [("ctc", False), ("ctc_merge_repeats", False), ("ctc_flipflop", True)])
@pytest.mark.parametrize("W", [2, 5, 25])
def test_row_no_envelope_matches_oracle(eng, oracle, model, ff, W):
    y1s, y2s, want = [], [], []
    for i in range(3):
        y1, y2 = synth_pair(8100 + i, T=70 + 25 * i, flipflop=ff)
        y1, y2 = y1[:60 + 20 * i], y2[:50 + 15 * i]
        y1s.append(y1); y2s.append(y2)
        want.append(oracle.cpp_beam_search_2d(y1, y2, None, W, model_=model, method_="row"))
    assert eng.beam_search_2d_batch(y1s, y2s, None, W, model=model, method="row") == want


def test_row_full_size_and_limits(eng, oracle):
    from poreover_amd import _lib
    y1, y2 = synth_pair(8200, T=4000)
    env = oracle.pair_decode(y1, y2, "poreover", 5, "row")["envelope"]
    assert eng.beam_search_2d_batch([y1], [y2], [env], 5, method="row")[0] == \
        oracle.cpp_beam_search_2d(y1, y2, env, 5, method_="row")
    # row without an envelope keeps V+2 times per node: beyond the first pass's store the pair is decoded again by the
    # retry pass with the larger one (T = 1500), and refused when that is exceeded too — never a different string
    a1, a2 = synth_pair(8201, T=1500)
    assert eng.beam_search_2d_batch([a1], [a2], None, 5, method="row")[0] == oracle.cpp_beam_search_2d(a1, a2, None, 5, method_="row")
    seqs, st = eng.beam_search_2d_batch([y1], [y2], None, 5, method="row", return_status=True)
    assert st[0] in (0, _lib.E_NOMEM)
    if st[0] == 0:
        assert seqs[0] == oracle.cpp_beam_search_2d(y1, y2, None, 5, method_="row")


# ---- method grid (hidden upstream option): one beam per cell
def test_grid_golden(eng, golden, golden_grid, golden_inputs):
    from conftest import grid_cases
    cases = grid_cases(golden, golden_grid, golden_inputs)
    for name, y1, y2, env, W, model, alphabet, want in cases:
        got = eng.beam_search_2d_batch([y1], [y2], None if env is None else [env], W, alphabet=alphabet, model=model,
                                       method="grid")[0]
        assert got == want, name
    # without an envelope every method but "row" runs grid (BeamSearch.h:441-458)
    name, y1, y2, env, W, model, alphabet, want = [c for c in cases if c[0] == "synth_9004"][0]
    assert eng.beam_search_2d_batch([y1], [y2], None, W, model=model, method="row_col")[0] == want


@pytest.mark.parametrize("W", [1, 3, 5, 8, 25])
def test_grid_envelope_matches_oracle_batch(eng, oracle, W):
    """ctc with an envelope (for the other models the reference's order in a narrow band is heap-address order)"""
    y1s, y2s, envs, want = [], [], [], []
    for i in range(6):
        y1, y2 = synth_pair(7100 + i, T=150 + 30 * i)
        if i % 2 == 0:
            env = oracle.pair_decode(y1, y2, "poreover", 5, "row_col")["envelope"]   # pipeline envelope
        else:
            env = oracle.diagonal_envelope(len(y1), len(y2), 5 + i)
        y1s.append(y1); y2s.append(y2); envs.append(env)
        want.append(oracle.cpp_beam_search_2d(y1, y2, env, W, method_="grid"))
    assert eng.beam_search_2d_batch(y1s, y2s, envs, W, method="grid") == want


@pytest.mark.parametrize("model,ff", [("ctc", False), ("ctc_merge_repeats", False), ("ctc_flipflop", True)])
@pytest.mark.parametrize("W", [2, 4, 7])
def test_grid_no_envelope_matches_oracle(eng, oracle, model, ff, W):
    y1s, y2s, want = [], [], []
    for i in range(4):
        y1, y2 = synth_pair(7200 + i, T=40 + 10 * i, flipflop=ff)
        y1s.append(y1); y2s.append(y2)
        want.append(oracle.cpp_beam_search_2d(y1, y2, None, W, model_=model, method_="grid"))
    assert eng.beam_search_2d_batch(y1s, y2s, None, W, model=model, method="grid") == want


def test_grid_limits(eng, oracle):
    from poreover_amd import _lib
    y1, y2 = synth_pair(7300, T=300)
    U, V = len(y1), len(y2)
    env = np.array(oracle.diagonal_envelope(U, V, 8))
    # row starts that move backwards: the ring store would have dropped what a later row reads
    bad = env.copy(); bad[100] = (max(0, bad[99][0] - 3), bad[100][1])
    seqs, st = eng.beam_search_2d_batch([y1, y1], [y2, y2], [env, bad], 5, method="grid", return_status=True)
    assert st.tolist() == [0, _lib.E_UNSUPPORTED]
    assert seqs[0] == oracle.cpp_beam_search_2d(y1, y2, env, 5, method_="grid")
    # out-of-range bounds, where the reference reads past y2
    oob = env.copy(); oob[50] = (oob[50][0], V + 2)
    _, st = eng.beam_search_2d_batch([y1], [y2], [oob], 5, method="grid", return_status=True)
    assert st[0] == _lib.E_ENVELOPE
    # a row with an empty band: its successors fall back to the seed beam
    gap = env.copy(); gap[120] = (gap[120][0], gap[120][0])
    assert eng.beam_search_2d_batch([y1], [y2], [gap], 5, method="grid")[0] == \
        oracle.cpp_beam_search_2d(y1, y2, gap, 5, method_="grid")


def test_grid_exact_ties(eng, oracle):
    """Beams wider than the finite candidates of a narrow band fill up with -inf scores: exact ties, resolved as
    libstdc++ leaves the creation-ordered candidates (the oracle's rule; two pairs the open-ended fuzz run found, on
    which score-then-creation-order gives other strings — 'ATTTT' and 'AAACACCATTT...')."""
    y1, y2 = synth_pair(773929922, T=178)
    env = np.asarray(oracle.pair_decode(y1, y2, "poreover", 5, "row_col")["envelope"])
    want = oracle.cpp_beam_search_2d(y1, y2, env, 9, method_="grid")
    assert want == "AAAAATTTT"
    assert eng.beam_search_2d_batch([y1], [y2], [env], 9, method="grid") == [want]
    y1, y2 = synth_pair(962012875, T=254)
    env = np.asarray(oracle.diagonal_envelope(len(y1), len(y2), 11))
    want = oracle.cpp_beam_search_2d(y1, y2, env, 13, method_="grid")
    assert want.startswith("CACATATTTACG")
    assert eng.beam_search_2d_batch([y1], [y2], [env], 13, method="grid") == [want]


def _quantised(y):
    """log-probabilities as a uint8 trace would give them (decode.py:92): exact score ties become common"""
    return np.log((np.clip(np.rint(np.exp(y) * 255), 0, 255) + 1e-7) / (255 + 1e-7))


def test_exact_ties_follow_the_reference(eng, oracle):
    """Beam::prune with exact score ties = libstdc++'s partial_sort on the creation-ordered candidates.  Seed 20127
    (bonito model, row, W = 8) is a pair on which score-then-creation-order gives a different string than the
    reference's own C++ does; plus a sweep of quantised pairs, every one against the oracle (which replays libstdc++
    and is pinned to the compiled reference on this input by tests/test_oracle_vs_ref.py — in a fresh process: the
    reference sorts node POINTERS, which is creation order only while malloc hands out ascending addresses, and in a
    long-lived test process with a fragmented heap its answer on this very pair flips between runs)."""
    rng = np.random.default_rng(5)
    Ts = [int(rng.integers(40, 200)) for _ in range(128)]      # the draw sequence that found the pair
    y1, y2 = synth_pair(20127, T=Ts[127])
    y1, y2 = _quantised(y1), _quantised(y2)
    U, V = len(y1), len(y2)
    env = np.array([(max(0, int(u * V / U) - 8), min(V, int(u * V / U) + 9)) for u in range(U)])
    want = oracle.cpp_beam_search_2d(y1, y2, env, 8, model_="ctc_merge_repeats", method_="row")
    assert eng.beam_search_2d_batch([y1], [y2], [env], 8, model="ctc_merge_repeats", method="row") == [want]
    for kind, model in (("poreover", "ctc"), ("bonito", "ctc_merge_repeats"), ("flipflop", "ctc_flipflop")):
        a, b, envs = [], [], []
        for i in range(14):
            p, q = synth_pair(21000 + i, T=60 + 9 * i, flipflop=(kind == "flipflop"))
            p, q = _quantised(p), _quantised(q)
            a.append(p); b.append(q)
            envs.append(np.array([(max(0, int(u * len(q) / len(p)) - 6), min(len(q), int(u * len(q) / len(p)) + 7)) for u in range(len(p))]))
        for method, W in (("row_col", 3), ("row_col", 5), ("row", 5), ("row_col", 16), ("row", 8)):
            got = eng.beam_search_2d_batch(a, b, envs, W, model=model, method=method)
            for i in range(len(a)):
                assert got[i] == oracle.cpp_beam_search_2d(a[i], b[i], envs[i], W, model_=model, method_=method), (kind, method, W, i)



# ---- every reason the register-state kernel hands a pair to beam2d_kernel, forced one by one: the
# pair must come back with the oracle's string, and the hand-over must really have happened (po_debug_deferred_pairs)
def _wide_env(U, V, half):
    return np.array([(max(0, int(u * V / U) - half), min(V, int(u * V / U) + half + 1)) for u in range(U)], dtype=np.int64)


@pytest.mark.parametrize("reason", ["window_beyond_walk_records", "window_beyond_store_geometry", "non_monotone_envelope",
                                    "row_groups_exhausted", "arena_exhausted", "odd_pairs"])
@pytest.mark.parametrize("model,ff", [("ctc", False), ("ctc_merge_repeats", False), ("ctc_flipflop", True)])
def test_reg_kernel_hands_over_and_result_is_the_oracles(eng, oracle, reason, model, ff):
    from poreover_amd import _lib
    if kernel_route_name() == "legacy":
        pytest.skip("beam2d_kernel alone: nothing is handed over")
    kind = {"ctc": "poreover", "ctc_merge_repeats": "bonito", "ctc_flipflop": "flipflop"}[model]
    route = {"reg": "reg"}.get(kernel_route_name(), "auto")
    W, starve, odd = 5, 0, False
    y1s, y2s, envs = [], [], []
    for i in range(4):
        y1, y2 = synth_pair(4700 + i, T=420 + 50 * i, flipflop=ff)
        env = np.asarray(oracle.pair_decode(y1, y2, kind, 5, "row_col")["envelope"], dtype=np.int64)
        if reason == "window_beyond_walk_records":       # 255 <= window < 512 at W <= 4: the store holds it (32 row groups of
            W = 3                                        # R = 512), the packed walk records (8-bit lengths) do not
            env = _wide_env(len(y1), len(y2), 140)
        elif reason == "window_beyond_store_geometry":   # ... at W = 5 the 32 row groups are too few: the pre-pass hands on
            env = _wide_env(len(y1), len(y2), 140)
        elif reason == "non_monotone_envelope":          # a row that starts before its predecessor (a caller's own array)
            env = env.copy()
            for u in range(40, len(env) - 1, 37):
                env[u, 0] = max(0, env[u, 0] - 6)
        y1s.append(y1); y2s.append(y2); envs.append(env)
    if reason == "row_groups_exhausted":
        starve = 1
    elif reason == "arena_exhausted":
        starve = 2
    elif reason == "odd_pairs":
        odd = True
    want = []
    for a, b, e in zip(y1s, y2s, envs):
        try:
            want.append(oracle.cpp_beam_search_2d(a, b, e, W, model_=model, method_="row_col"))
        except oracle.OracleError:
            want.append(None)
    assert all(w is not None for w in want)
    _lib.deferred_pairs(reset=True)
    _lib.set_pair_route(route, defer_odd=odd, starve=starve)
    try:
        got = eng.beam_search_2d_batch(y1s, y2s, envs, W, model=model, method="row_col")
        handed = _lib.deferred_pairs(reset=True)
    finally:
        _lib.set_pair_route(route)
    assert got == want, reason
    assert handed >= (2 if reason == "odd_pairs" else 1), (reason, handed)
    # ... and the same pairs without the hook / with ordinary envelopes are decoded by the kernel itself
    if reason in ("row_groups_exhausted", "arena_exhausted", "odd_pairs"):
        assert eng.beam_search_2d_batch(y1s, y2s, envs, W, model=model, method="row_col") == want
        assert _lib.deferred_pairs(reset=True) == 0
