import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def hmm_golden():
    return np.load(os.path.join(GOLDEN, "hmm_golden.npz"))


@pytest.fixture(scope="session")
def posteriors_golden():
    return np.load(os.path.join(GOLDEN, "posteriors_golden.npz"))


@pytest.fixture(scope="session")
def endpoint_golden():
    return np.load(os.path.join(GOLDEN, "endpoint_golden.npz"))


@pytest.fixture(scope="session")
def mfcc_golden():
    return np.load(os.path.join(GOLDEN, "mfcc_golden.npz"))


@pytest.fixture(scope="session")
def folds_golden():
    return np.load(os.path.join(GOLDEN, "folds_golden.npz"))


def golden_params(g, case):
    from oracle.hmm_oracle import Params
    p = case + "_"
    return Params(g[p + "pi"], g[p + "A"], g[p + "mu"], g[p + "var"])
