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


# ---------------------------------------------------------------------------
# The OTHER front-end of the reference: extract_mfcc (HMM_lib.py:11-15,
# HMM_final.py:11-15) = librosa.feature.mfcc(y, sr, n_mfcc=13) with librosa's
# defaults, then the same StandardScaler.  librosa is a third-party package that
# the reference neither vendors nor pins (requirements.txt:4 names it without a
# version) and that is absent from this image, and the reference holds no output
# of that call: PARITY UNPINNED.  What follows restates the published algorithm of
# librosa 0.10.x (feature.mfcc -> feature.melspectrogram -> core.stft,
# filters.mel, core.power_to_db; keyword-only `y=` as the call sites use it came
# with 0.10, whose stft pads with zeros -- pad_mode="constant"; 0.9 and earlier
# reflected, offered here as pad_mode="reflect").  librosa keeps the audio, the
# transform and the filter matrix in float32; this restatement is float64.
# ---------------------------------------------------------------------------
@dataclass(frozen=True)
class LibrosaConfig:
    sample_rate: int = 22050
    n_mfcc: int = 13               # HMM_lib.py:11
    n_fft: int = 2048              # melspectrogram default
    hop_length: int = 512          # melspectrogram default
    n_mels: int = 128              # filters.mel default
    pad_mode: str = "constant"     # stft default since 0.10
    amin: float = 1e-10            # power_to_db default
    top_db: float = 80.0           # power_to_db default
    zscore: bool = True            # HMM_lib.py:14

    def n_frames(self, n_samples):
        """stft(center=True): one frame per hop, the first centred on sample 0."""
        return 1 + n_samples // self.hop_length if n_samples > 0 else 0


def slaney_hz_to_mel(f):
    """librosa.hz_to_mel(htk=False): linear below 1 kHz, logarithmic above."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    logstep = np.log(6.4) / 27.0
    lin = f / f_sp
    with np.errstate(divide="ignore"):
        log = min_log_hz / f_sp + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log, lin)


def slaney_mel_to_hz(m):
    """librosa.mel_to_hz(htk=False)."""
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_filterbank(sr, n_fft, n_mels=128):
    """librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=0, fmax=sr/2, htk=False,
    norm='slaney') -> (n_mels, 1 + n_fft//2): triangles over the bins' centre
    frequencies, each scaled to unit area (2 / width in Hz)."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = slaney_mel_to_hz(np.linspace(slaney_hz_to_mel(0.0), slaney_hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    fb = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        fb[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return fb * enorm[:, None]


def librosa_melspectrogram(y, cfg):
    """feature.melspectrogram(power=2.0) -> (T, n_mels): centred frames of n_fft
    samples (padding per cfg.pad_mode), periodic Hann, |rfft|^2, mel filters."""
    y = np.asarray(y, dtype=np.float64)
    N, hop = cfg.n_fft, cfg.hop_length
    T = cfg.n_frames(len(y))
    if T == 0:
        return np.zeros((0, cfg.n_mels))
    yp = np.pad(y, N // 2, mode=cfg.pad_mode)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(N) / N)      # get_window('hann', N, fftbins=True)
    frames = np.stack([yp[t * hop: t * hop + N] for t in range(T)]) * win
    spec = np.fft.rfft(frames, N)
    power = spec.real ** 2 + spec.imag ** 2
    return power @ slaney_filterbank(cfg.sample_rate, N, cfg.n_mels).T


def librosa_mfcc(y, cfg):
    """feature.mfcc(y=, sr=, n_mfcc=) transposed to (T, n_mfcc): power_to_db of the
    mel spectrogram (ref 1, amin, top_db below the utterance's maximum), DCT-II
    'ortho' over the mel axis, first n_mfcc rows."""
    S = librosa_melspectrogram(y, cfg)
    if S.shape[0] == 0:
        return np.zeros((0, cfg.n_mfcc))
    db = 10.0 * np.log10(np.maximum(cfg.amin, S))
    if cfg.top_db:
        db = np.maximum(db, db.max() - cfg.top_db)
    return db @ dct_matrix(cfg.n_mels, cfg.n_mfcc).T


def librosa_extract(y, cfg):
    """extract_mfcc, HMM_lib.py:11-15, after the file has been read."""
    f = librosa_mfcc(y, cfg)
    return zscore(f) if cfg.zscore else f
