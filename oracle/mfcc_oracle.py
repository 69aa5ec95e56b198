"""CPU oracle (test infrastructure) for the MFCC(+delta) front-end.

fp64 numpy restatement of the reference front-end.  Each function names the
reference lines it follows.  Checked against the imported reference through
``tests/golden/mfcc_*.npz`` (parity pinned).

The reference hard-codes NFFT=1024, 26 filters designed for 16 kHz, 13
cepstra and one delta order; every constant is a parameter here so the
synthetic 16 kHz / 512-point / 40-mel / delta+delta-delta shape can be
expressed with the same code.
"""
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class MfccConfig:
    sample_rate: int = 16000
    frame_size: float = 0.025      # seconds, mfcc.py:15
    frame_stride: float = 0.01     # seconds, mfcc.py:15
    nfft: int = 1024               # mfcc.py:25
    n_mel: int = 26                # mfcc.py:47
    mel_sample_rate: int = 16000   # mfcc.py:40 never forwards the real rate
    n_ceps: int = 13               # mfcc.py:78
    delta_order: int = 1           # reference: one delta block (mfcc.py:165-167)
    delta_n: int = 2               # mfcc.py:88
    log_mult: float = 20.0         # 20*log10, mfcc.py:44
    zscore: bool = True            # HMM_final.py:22

    @property
    def frame_len(self):
        # mfcc.py:18-19: the float product is truncated where it is used.
        return int(self.frame_size * self.sample_rate)

    @property
    def hop(self):
        return int(self.frame_stride * self.sample_rate)

    @property
    def n_feat(self):
        return self.n_ceps * (1 + self.delta_order)

    def n_frames(self, n_samples):
        """Raw frame count, mfcc.py:19 (tail dropped, no padding)."""
        if n_samples < self.frame_len:
            return 0
        return (n_samples - self.frame_len) // self.hop + 1

    def n_out_frames(self, n_samples):
        """Frames kept after the delta trim, mfcc.py:96 / :167."""
        return max(self.n_frames(n_samples) - 2 * self.delta_n, 0) \
            if self.delta_order > 0 else self.n_frames(n_samples)


def hamming(n):
    """Symmetric Hamming window, np.hamming as used at mfcc.py:21."""
    if n == 1:
        return np.ones(1)
    k = np.arange(n)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * k / (n - 1))


def mel_filterbank(nfft, mel_sample_rate=16000, n_mel=26):
    """Triangular HTK-mel filter matrix (n_mel, nfft//2+1), mfcc.py:47-61."""
    top = 2595.0 * np.log10(1.0 + mel_sample_rate / 2 / 700)
    mel_pts = np.linspace(0, top, n_mel + 2)
    hz = 700 * (10 ** (mel_pts / 2595) - 1)
    edges = np.floor((nfft + 1) * hz / mel_sample_rate)
    fb = np.zeros((n_mel, nfft // 2 + 1))
    for m in range(n_mel):
        lo, mid, hi = edges[m], edges[m + 1], edges[m + 2]
        for k in range(int(lo), int(mid)):
            fb[m, k] = (k - lo) / (mid - lo)
        for k in range(int(mid), int(hi)):
            fb[m, k] = (hi - k) / (hi - mid)
    return fb


def dct_matrix(n_in, n_out):
    """Orthonormal DCT-II rows 0..n_out-1 (scipy dct type=2 norm='ortho',
    mfcc.py:79)."""
    n = np.arange(n_in)
    k = np.arange(n_out)[:, None]
    c = np.sqrt(2.0 / n_in) * np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_in))
    c[0] *= 1.0 / np.sqrt(2.0)
    return c


def frames_windowed(signal, cfg):
    """mfcc.py:15-23."""
    sig = np.asarray(signal, dtype=np.float64)
    L, hop = cfg.frame_len, cfg.hop
    T = cfg.n_frames(len(sig))
    out = np.empty((T, L))
    win = hamming(L)
    for t in range(T):
        out[t] = sig[t * hop: t * hop + L] * win
    return out


def log_mel(frames, cfg):
    """mfcc.py:35-45: |rfft|^2/NFFT -> mel -> eps floor at exact zero -> 20 log10."""
    spec = np.fft.rfft(frames, cfg.nfft)
    power = (spec.real ** 2 + spec.imag ** 2) / cfg.nfft
    fb = mel_filterbank(cfg.nfft, cfg.mel_sample_rate, cfg.n_mel)
    e = power @ fb.T
    e = np.where(e == 0, np.finfo(float).eps, e)
    return cfg.log_mult * np.log10(e)


def cepstra(signal, cfg):
    """compute_mfcc, mfcc.py:82-86 -> (T, n_ceps)."""
    lm = log_mel(frames_windowed(signal, cfg), cfg)
    return lm @ dct_matrix(cfg.n_mel, cfg.n_ceps).T


def delta_full(c, n=2):
    """Regression delta over edge-padded frames BEFORE the trim, mfcc.py:88-95.
    c: (T, K) -> (T, K)."""
    T = c.shape[0]
    denom = 2 * sum(i * i for i in range(1, n + 1))
    out = np.zeros_like(c)
    for t in range(T):
        acc = np.zeros(c.shape[1])
        for k in range(-n, n + 1):
            acc += k * c[min(max(t + k, 0), T - 1)]
        out[t] = acc / denom
    return out


def features(signal, cfg):
    """calculate_mfcc_for_onerecording body, mfcc.py:141-169, transposed to
    (T-2n, n_feat).  delta_order=2 appends the delta of the untrimmed delta
    (an extension the reference does not have)."""
    c = cepstra(signal, cfg)
    n = cfg.delta_n
    blocks = [c]
    for _ in range(cfg.delta_order):
        blocks.append(delta_full(blocks[-1], n))
    if cfg.delta_order == 0:
        return c
    if c.shape[0] <= 2 * n:
        return np.zeros((0, cfg.n_feat))
    return np.concatenate([b[n:-n] for b in blocks], axis=1)


def zscore(f):
    """StandardScaler().fit_transform per utterance, HMM_final.py:22: population
    variance; a constant column keeps scale 1."""
    f = np.asarray(f, dtype=np.float64)
    if f.shape[0] == 0:
        return f.copy()
    n = f.shape[0]
    mean = f.mean(axis=0)
    var = ((f - mean) ** 2).mean(axis=0)
    eps = np.finfo(np.float64).eps
    constant = var <= n * eps * var + (n * mean * eps) ** 2
    scale = np.sqrt(var)
    scale[constant] = 1.0
    scale[scale == 0] = 1.0
    return (f - mean) / scale


def extract(signal, cfg):
    """extract_mfcc_calc, HMM_final.py:18-23 (features, then optional z-score)."""
    f = features(signal, cfg)
    return zscore(f) if cfg.zscore else f
