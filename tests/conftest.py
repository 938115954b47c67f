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
