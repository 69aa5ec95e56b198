"""Front-end with the reference's call surface (mfcc.py), computed on the GPU.

Same function names and argument meaning as the reference's ``mfcc.py``:
``compute_mfcc`` (mfcc.py:82), ``calculate_mfcc_for_onerecording`` (mfcc.py:134),
``mel_filterbank`` (mfcc.py:47).  The arithmetic (framing, Hamming, FFT, mel,
log, DCT, delta, z-score) is libsrx's kernel chain; there is no numpy path.
``extract_batch`` is the additive batched entry point.
"""
import numpy as np

from . import SpeechUtil as su
from .runtime import Context, MfccPlan, filterbank_host

_plans = {}


def _plan(ctx, sample_rate, **kw):
    key = (id(ctx), int(sample_rate)) + tuple(sorted(kw.items()))
    if key not in _plans:
        _plans[key] = MfccPlan(ctx, sample_rate, **kw)
    return _plans[key]


def drop_plans(ctx):
    """Destroy the cached plans of a context (called by Context.close: a plan is bound to
    the context it was made with)."""
    for key in [k for k, p in _plans.items() if p.ctx is ctx]:
        _plans.pop(key).close()


def mel_filterbank(NFFT, sample_rate=16000, num_filters=26):
    """Triangular HTK-mel filter matrix (num_filters, NFFT//2+1), mfcc.py:47-61."""
    return filterbank_host(NFFT, sample_rate, num_filters)


def _pack(signals, dtype):
    lens = np.array([len(s) for s in signals], dtype=np.int64)
    so = np.zeros(len(signals) + 1, dtype=np.int64)
    np.cumsum(lens, out=so[1:])
    npdt = np.float64 if dtype == "float64" else np.float32
    if len(signals) == 1:
        packed = np.ascontiguousarray(signals[0], dtype=npdt).reshape(-1)
    else:
        packed = np.concatenate([np.asarray(s, dtype=npdt).reshape(-1) for s in signals]) \
            if len(signals) else np.zeros(0, dtype=npdt)
    return packed, so


def extract_ragged(signals, sample_rate, *, nfft=1024, n_mel=26, mel_sample_rate=16000, n_ceps=13,
                   delta_order=1, zscore=True, dtype="float32", ctx=None):
    """MFCC(+delta) features of many utterances in one launch, left on the device: a
    ``Ragged`` batch (packed (sum T_u, n_feat) features + frame offsets) that scoring,
    decoding and training take as it is."""
    ctx = ctx or Context.default()
    plan = _plan(ctx, sample_rate, nfft=nfft, n_mel=n_mel, mel_sample_rate=mel_sample_rate,
                 n_ceps=n_ceps, delta_order=delta_order, zscore=zscore)
    packed, so = _pack(signals, dtype)
    return plan.run_ragged(packed, so, dtype=dtype)


def extract_batch(signals, sample_rate, *, nfft=1024, n_mel=26, mel_sample_rate=16000, n_ceps=13,
                  delta_order=1, zscore=True, dtype="float32", ctx=None, to_host=True):
    """MFCC(+delta) features of many utterances in one launch.

    signals: list of 1-D arrays.  Returns (features, lengths): features is the
    packed (sum T_u, n_feat) matrix (a numpy array, or the device tensor with
    to_host=False), lengths the frames per utterance."""
    data = extract_ragged(signals, sample_rate, nfft=nfft, n_mel=n_mel, mel_sample_rate=mel_sample_rate,
                          n_ceps=n_ceps, delta_order=delta_order, zscore=zscore, dtype=dtype, ctx=ctx)
    feats = data.X.cpu().numpy() if to_host else data.X
    return feats, np.diff(data.offsets_host)


def compute_mfcc(signal, sample_rate, dtype="float64"):
    """(T, 13) cepstra of one signal, mfcc.py:82-86."""
    f, _ = extract_batch([signal], sample_rate, delta_order=0, zscore=False, dtype=dtype)
    return f.astype(np.float64)


def calculate_mfcc_for_onerecording(check, audio_file_path, sample_rate=16000, dtype="float64"):
    """(26, T-4) MFCC + delta of one WAV file, mfcc.py:134-169.  ``check`` (a
    comparison with librosa the reference never enables) is accepted and
    ignored."""
    audio_file_path = str(audio_file_path).replace(".wav", "")
    signal = su.read_recording(audio_file_path, sample_rate)
    f, _ = extract_batch([signal], sample_rate, delta_order=1, zscore=False, dtype=dtype)
    return f.astype(np.float64).T


def extract_features(signal, sample_rate, dtype="float64"):
    """(T-4, 26) z-scored features of one signal: the body of
    extract_mfcc_calc (HMM_final.py:18-23) without the file read."""
    f, _ = extract_batch([signal], sample_rate, delta_order=1, zscore=True, dtype=dtype)
    return f.astype(np.float64)


def librosa_mfcc_ragged(signals, sample_rate, n_mfcc=13, *, zscore=False, dtype="float32", ctx=None, **kw):
    """``librosa_mfcc_batch`` with the features left on the device as a ``Ragged`` batch."""
    ctx = ctx or Context.default()
    key = ("librosa", id(ctx), int(sample_rate), int(n_mfcc), bool(zscore)) + tuple(sorted(kw.items()))
    if key not in _plans:
        _plans[key] = MfccPlan.librosa(ctx, sample_rate, n_mfcc=n_mfcc, zscore=zscore, **kw)
    packed, so = _pack(signals, dtype)
    return _plans[key].run_ragged(packed, so, dtype=dtype)


def librosa_mfcc_batch(signals, sample_rate, n_mfcc=13, *, zscore=False, dtype="float32", ctx=None,
                       to_host=True, **kw):
    """``librosa.feature.mfcc(y=, sr=, n_mfcc=)`` conventions (the front-end of
    extract_mfcc, HMM_lib.py:11-15) for many utterances in one launch; keyword
    arguments as ``MfccPlan.librosa``.  Returns (packed (sum T, n_mfcc), lengths).
    librosa itself is not used (it is a third-party package the reference does not
    pin); this follows the published algorithm of its 0.10 line."""
    data = librosa_mfcc_ragged(signals, sample_rate, n_mfcc, zscore=zscore, dtype=dtype, ctx=ctx, **kw)
    feats = data.X.cpu().numpy() if to_host else data.X
    return feats, np.diff(data.offsets_host)
