"""CPU: round-2 API corners — the oracle pinned to the reference's outputs (tests/golden/make_golden_extra.py) and the
host-side mirrors of decoding_cy's containers / scalar helpers."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, hexf


@pytest.fixture(scope="module")
def extra():
    with open(os.path.join(GOLDEN_DIR, "extra_golden.json")) as f:
        return json.load(f)


def test_oracle_alignment_with_scores(oracle, extra):
    for c in extra["align_scores"]:
        m, mm, g = c["scores"]
        a1, a2 = oracle.global_pair(c["s1"], c["s2"], m, mm, g)
        assert ["".join(a1), "".join(a2)] == c["full"], c["scores"]
        b1, b2 = oracle.global_pair_banded(c["s1"], c["s2"], 25, m, mm, g)
        assert ["".join(b1), "".join(b2)] == c["banded25"], c["scores"]
    # the default scores are back afterwards
    a1, a2 = oracle.global_pair("ACGTACGTTT", "ACGTCGTTTA")
    assert ("".join(a1), "".join(a2)) == ("ACGTACGTTT-", "ACGT-CGTTTA")


def test_oracle_pair_prefix_search_full_envelope_is_dense(oracle, golden):
    with np.errstate(divide="ignore"):
        for key, rec in golden["pair_prefix_toy"].items():
            a, b = key.split("_")
            ya, yb = np.log(np.array(golden["prefix_prob"][a])), np.log(np.array(golden["prefix_prob"][b]))
            # the toys use the alphabet "AB": the oracle's search is over the first C - 1 symbols named "ACGT"
            full = np.array([(0, len(yb))] * (len(ya) + 1))
            lab_d, lp_d = oracle.pair_prefix_search_log(ya, yb, "py")
            lab_e, lp_e = oracle.pair_prefix_search_log(ya, yb, "py", full)
            assert lab_d == lab_e and np.isclose(lp_d, lp_e, rtol=1e-12)
            assert lab_d.replace("C", "B") == rec["py"][0] and np.isclose(lp_d, hexf(rec["py"][1]), rtol=1e-9)


def test_host_mirrors_of_decoding_cy_helpers(extra):
    from poreover_amd.decoding import decoding_cy as cy
    m = cy.PySparseMatrix()
    m.push_row(2, 5); m.push_row(0, 1)
    assert m.get(0, 3) == -np.inf and m.get(5, 0) == -np.inf and m.get(0, 9) == -np.inf
    m.set(0, 3, 1.5); m.set(0, 9, 7.0); m.set(7, 0, 1.0)      # the last two are dropped silently
    assert m.get(0, 3) == 1.5 and m.get(0, 9) == -np.inf and m.get(1, 1) == -np.inf
    assert np.isclose(cy.logsumexp(np.log(np.array([0.25, 0.25, 0.5]))), 0.0, atol=1e-15)
    assert cy.logsumexp(np.array([-np.inf])) == -np.inf
    pp = extra["pair_prefix_prob"]
    a1 = np.array([hexf(x) for x in pp["alpha1"]]); a2 = np.array([hexf(x) for x in pp["alpha2"]])
    env, ranges, idx = cy.diagonal_band_envelope(6, 9, 2)
    assert ranges.tolist()[:2] == [[0, 2], [0, 4]] and env.get(0, 1) == 1 and tuple(idx[0]) == (0, 0)
    assert len(idx) == sum(e - s + 1 for s, e in ranges)


def test_device_logaddexp_formulation(tmp_path):
    """The kernels' logaddexp (po_device.h PoLaeFast) replayed on the CPU from the same tables: the integer-step
    formulation gives the bits of the rint / ldexp one, and both stay within 1.6e-16 of a long-double log1p(exp(d))."""
    import os
    import re
    import subprocess
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "check_lae.c")
    exe = str(tmp_path / "check_lae")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"])
    out = subprocess.run([exe, "2000000"], capture_output=True, text=True, check=True).stdout
    m = re.search(r"max abs err: old (\S+) new (\S+) glibc-double \S+; old!=new in (\d+) of", out)
    assert m, out
    assert float(m.group(1)) < 1.7e-16 and float(m.group(2)) < 1.7e-16 and int(m.group(3)) == 0, out
    assert "lae(-inf,-inf) old -inf new -inf" in out
