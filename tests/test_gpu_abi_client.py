"""The C ABI from a program that is not Python: tests/abi/abi_client.cpp includes only
include/srx.h and the HIP runtime, links libsrx.so, allocates its own device buffers and
scores a golden case in float64.  The binary is built by __graft_entry__.build()."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c_client_scores_golden_case(tmp_path, hmm_golden):
    exe = os.path.join(ROOT, "tests", "abi", "abi_client")
    assert os.path.exists(exe), "tests/abi/abi_client is missing: run __graft_entry__.build()"
    g = hmm_golden
    cases = ["s3d26", "s8d39"]
    for case in cases:
        p = case + "_"
        pi, A, mu, var = g[p + "pi"], g[p + "A"], g[p + "mu"], g[p + "var"]
        X, lengths = g[p + "X"], g[p + "lengths"]
        S, D = mu.shape
        off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        path = tmp_path / f"{case}.bin"
        with open(path, "wb") as f:
            f.write(struct.pack("<5q", 1, S, D, len(lengths), len(X)))
            for a in (pi, A, mu, var, X):
                f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
            f.write(off.tobytes())
            f.write(np.ascontiguousarray(g[p + "utt_logprob"], dtype=np.float64).tobytes())
        r = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (r.stdout, r.stderr)
        assert "max_rel_err" in r.stdout and "refused_null -1" in r.stdout
