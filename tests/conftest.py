import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_grid():
    """outputs of the reference's hidden `grid` pair method (tests/golden/make_golden_grid.py)"""
    with open(os.path.join(GOLDEN_DIR, "golden_grid.json")) as f:
        return json.load(f)


def grid_cases(golden, golden_grid, golden_inputs):
    """(name, y1, y2, envelope or None, W, model, alphabet, expected) for every grid golden vector"""
    from poreover_amd.synth import synth_pair
    out = []
    y = np.log(golden_inputs["poreover_csv_prob"])
    T = len(y)
    env10 = np.array([(max(0, i - 10), min(i + 10, T)) for i in range(T)])
    for k, v in golden_grid["csv_self_grid"].items():
        out.append(("csv_" + k, y, y, env10, int(k[1:]), "ctc", "ACGT", v))
    with np.errstate(divide="ignore"):
        for k, v in golden_grid["toy_noenv_grid"].items():
            a, b = k.split("_")
            out.append(("toy_" + k, np.log(np.array(golden["prefix_prob"][a])), np.log(np.array(golden["prefix_prob"][b])),
                        None, 5, "ctc", "AB", v))
    for c in golden_grid["synthetic"]:
        y1, y2 = synth_pair(c["seed"], T=c["T"], flipflop=c["flipflop"])
        U, V = len(y1), len(y2)
        env = None
        if c["band"] is not None:
            env = np.array([(max(0, int(u * V / U) - c["band"]), min(V, int(u * V / U) + c["band"])) for u in range(U)])
        out.append(("synth_%d" % c["seed"], y1, y2, env, c["W"], c["model"], "ACGT", c["out"]))
    return out


@pytest.fixture(scope="session")
def golden_inputs():
    return dict(np.load(os.path.join(GOLDEN_DIR, "inputs.npz")))


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure).  Built on demand with gcc."""
    from oracle import po_oracle
    po_oracle.build()
    return po_oracle


def hexf(s):
    return float.fromhex(s)
