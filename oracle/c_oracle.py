"""ctypes wrapper of oracle/libsrx_oracle.so (compiled C oracle; test
infrastructure only -- built by `make -C oracle`)."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsrx_oracle.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_PATH)
        _lib.orc_cepstra.restype = C.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _prep(p, X, lengths):
    X = np.ascontiguousarray(X, dtype=np.float64)
    L = np.ascontiguousarray(lengths if lengths is not None else [len(X)], dtype=np.int64)
    if L.sum() != len(X):
        raise ValueError(f"lengths array {L} doesn't sum to {len(X)} samples")
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (p.pi, p.A, p.means, p.covars)]
    return X, L, arrs


def score(p, X, lengths=None):
    X, L, (pi, A, mu, var) = _prep(p, X, lengths)
    out = np.zeros(len(L))
    rc = _load().orc_score(_p(X), _p(L), C.c_int64(len(L)), C.c_int(X.shape[1]), _p(pi), _p(A),
                           _p(mu), _p(var), C.c_int(p.S), _p(out))
    assert rc == 0
    return out


def estep(p, X, lengths=None):
    """Packed record [nobs, logprob, start, trans, post, obs, obs2]."""
    X, L, (pi, A, mu, var) = _prep(p, X, lengths)
    S, D = p.S, X.shape[1]
    out = np.zeros(2 + S + S * S + S + 2 * S * D)
    rc = _load().orc_estep(_p(X), _p(L), C.c_int64(len(L)), C.c_int(D), _p(pi), _p(A), _p(mu),
                           _p(var), C.c_int(S), _p(out))
    assert rc == 0
    return out


def viterbi(p, X, lengths=None):
    X, L, (pi, A, mu, var) = _prep(p, X, lengths)
    path = np.zeros(len(X), dtype=np.int32)
    lp = np.zeros(len(L))
    with np.errstate(divide="ignore"):
        rc = _load().orc_viterbi(_p(X), _p(L), C.c_int64(len(L)), C.c_int(X.shape[1]), _p(pi),
                                 _p(A), _p(mu), _p(var), C.c_int(p.S), _p(path), _p(lp))
    assert rc == 0
    return lp, path


def cepstra(signal, cfg, fbank):
    sig = np.ascontiguousarray(signal, dtype=np.float64)
    T = cfg.n_frames(len(sig))
    out = np.zeros((max(T, 0), cfg.n_ceps))
    fb = np.ascontiguousarray(fbank, dtype=np.float64)
    n = _load().orc_cepstra(_p(sig), C.c_int64(len(sig)), C.c_int(cfg.frame_len), C.c_int(cfg.hop),
                            C.c_int(cfg.nfft), C.c_int(cfg.n_mel), _p(fb), C.c_int(cfg.n_ceps),
                            C.c_double(cfg.log_mult), _p(out))
    assert n == T
    return out
