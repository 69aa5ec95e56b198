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


def viterbi_lattice(log_pi, log_A, log_B):
    """delta[t, j] of utils.viterbi (my_hmmlearn/utils.py:243-256) in float64."""
    T, S = log_B.shape
    d = np.empty((T, S))
    d[0] = log_pi + log_B[0]
    for t in range(1, T):
        d[t] = np.max(d[t - 1][:, None] + log_A, axis=0) + log_B[t]
    return d


def path_score(log_pi, log_A, log_B, path):
    """Log-probability of one state path under the float64 model."""
    path = np.asarray(path, dtype=np.int64)
    s = log_pi[path[0]] + log_B[np.arange(len(path)), path].sum()
    if len(path) > 1:
        s += log_A[path[:-1], path[1:]].sum()
    return float(s)


def assert_ties_only(pi, A, log_B, lengths, got_path, ref_path, ref_logprob, eps=1e-5):
    """"Bit-exact, ties excepted" for a float32 Viterbi path against the float64 one.

    Per utterance: (1) the float32 path, scored under the float64 model, reaches the optimum
    within eps * |optimum|; (2) every maximal run of frames where the two paths differ ends
    -- the backtrace walks from the last frame to the first, so that is where the run begins
    for it -- at a frame where the two candidates, given the same successor state, are a
    near-tie in the float64 lattice: gap below eps * |optimum|.  Returns (frames that
    differ, largest gap seen, largest score shortfall)."""
    with np.errstate(divide="ignore"):
        log_pi, log_A = np.log(np.asarray(pi, dtype=np.float64)), np.log(np.asarray(A, dtype=np.float64))
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    n_diff, worst_gap, worst_short = 0, 0.0, 0.0
    ref_logprob = np.atleast_1d(ref_logprob)
    for u in range(len(lengths)):
        a, b = off[u], off[u + 1]
        if b == a:
            continue
        g, r, lb = np.asarray(got_path[a:b]), np.asarray(ref_path[a:b]), np.asarray(log_B[a:b], dtype=np.float64)
        if np.array_equal(g, r):
            continue
        tol = eps * abs(float(ref_logprob[u]))
        short = float(ref_logprob[u]) - path_score(log_pi, log_A, lb, g)
        assert short <= tol, f"utterance {u}: float32 path is {short} below the optimum (tol {tol})"
        worst_short = max(worst_short, short)
        d = viterbi_lattice(log_pi, log_A, lb)
        diff = g != r
        n_diff += int(diff.sum())
        T = len(g)
        for t in np.nonzero(diff)[0]:
            if t + 1 < T and diff[t + 1]:
                continue                              # inside a run: decided by the run's last frame
            if t + 1 == T:
                gap = d[t, r[t]] - d[t, g[t]]
            else:
                nxt = r[t + 1]                        # same successor in both paths
                gap = (d[t, r[t]] + log_A[r[t], nxt]) - (d[t, g[t]] + log_A[g[t], nxt])
            assert 0 <= gap + 1e-9 and gap <= tol, \
                f"utterance {u}, frame {t}: paths part at a gap of {gap}, not a tie (tol {tol})"
            worst_gap = max(worst_gap, float(gap))
    return n_diff, worst_gap, worst_short
