"""GPU parity of the pair gamma DP (Gamma.h envelope version, decoding_cy dense version) vs the oracle, the
reference's compiled C++ (when built) and its golden values."""
import numpy as np
import pytest

from conftest import hexf
from poreover_amd.synth import synth_pair

pytestmark = pytest.mark.gpu
RTOL = 1e-12   # device exp/log vs libm, accumulated over U + V logaddexp steps


@pytest.fixture(scope="module")
def eng():
    from poreover_amd import _lib, batch
    _lib.load()
    return batch


def _inclusive_band(U, V, w):
    e = np.array([(max(0, int(u / U * V) - w), min(V, int(u / U * V) + w)) for u in range(U + 1)])
    e[U] = (max(0, V - w), V)
    return e


def test_gamma_envelope_matches_oracle(eng, oracle, capsys):
    y1s, y2s, envs, want = [], [], [], []
    for i in range(4):
        y1, y2 = synth_pair(9700 + i, T=200 + 150 * i)
        e = _inclusive_band(len(y1), len(y2), 8 + 3 * i)
        y1s.append(y1); y2s.append(y2); envs.append(e)
        want.append(oracle.pair_gamma_log_envelope(y1, y2, e))
    got = eng.pair_gamma_batch(y1s, y2s, envs)
    assert np.allclose(got, want, rtol=RTOL, atol=0)
    if oracle.have_ref():
        assert np.isclose(got[0], oracle.ref_pair_gamma_log_envelope(y1s[0], y2s[0], envs[0]), rtol=RTOL)
    from poreover_amd.decoding import decoding_cpp
    assert decoding_cpp.cpp_pair_gamma_log_envelope(y1s[0], y2s[0], envs[0]) is None     # prints, like upstream
    assert np.isclose(float(capsys.readouterr().out.strip()), want[0], rtol=RTOL)


def test_gamma_dense_golden(eng, golden, golden_inputs):
    from poreover_amd.decoding import decoding_cy
    y = golden_inputs["prefix_y"]
    ya, yb = y[:30], y[40:65]
    g = decoding_cy.pair_gamma_log(ya, yb)
    want = golden_inputs["gamma_dense_cy"]
    assert g.shape == want.shape
    finite = want > -9000
    assert np.allclose(g[finite], want[finite], rtol=1e-11, atol=0)
    assert np.all(g[~finite] < -9000)
    pm = golden["prefix_prob"]
    with np.errstate(divide="ignore"):
        for k, rec in golden["pair_prefix_toy"].items():
            a, b = k.split("_")
            g00 = decoding_cy.pair_gamma_log(np.log(np.array(pm[a])), np.log(np.array(pm[b])))[0, 0]
            assert np.isclose(g00, hexf(rec["gamma00_cy"]), rtol=1e-12)
    assert decoding_cy.diagonal_band_envelope(10, 20, 3)[1].tolist()[:3] == [[0, 3], [0, 5], [1, 7]]
