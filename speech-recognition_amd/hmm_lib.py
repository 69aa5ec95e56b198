"""The reference's second harness, ``HMM_lib.py``: stock-hmmlearn models on
librosa MFCCs, with its function names and argument order.

``extract_mfcc`` (HMM_lib.py:11), ``train_hmm`` (:26), ``recognize_lib`` (:50),
``load_models`` (:67), ``train_using_all_files`` (:55), ``test_trained_model``
(:130).  The features follow librosa 0.10's published ``feature.mfcc`` algorithm
on the GPU front-end and the model is this package's ``GaussianHMM``; neither
third-party package is used, and no reference output pins them (the reference
holds no result of these calls): results are close to, not guaranteed identical
with, an installation of librosa + hmmlearn.
"""
import os
import pickle

import numpy as np
from scipy.io import wavfile

from . import mfcc as fe
from . import recognizer as _rec
from .hmm import GaussianHMM
from .recognizer import WORDS, _word_of, classify, extract_mfcc_calc, score_matrix, word_files   # noqa: F401


def _resample(y, sr, sample_rate):
    """Band-limited resampling of a float32 signal from ``sr`` to ``sample_rate`` Hz by a
    polyphase FIR filter (scipy.signal.resample_poly: Kaiser window, ratio reduced by the greatest
    common divisor), ceil(n * sample_rate / sr) samples out -- the length librosa.load returns.
    librosa's own default is the soxr library's "soxr_hq" filter, which is not in the image:
    the two agree to the filters' stop-band ripple, not bit for bit (parity unpinned)."""
    from math import gcd
    from scipy.signal import resample_poly
    g = gcd(int(sr), int(sample_rate))
    return resample_poly(y.astype(np.float64), int(sample_rate) // g, int(sr) // g).astype(np.float32)


def _load(audio_path, sample_rate):
    """librosa.load(path, sr=sample_rate): float32 mono in [-1, 1), resampled to ``sample_rate``
    when the file is stored at another rate (HMM_lib.py:12)."""
    path = str(audio_path)
    if not path.endswith(".wav"):
        path += ".wav"
    sr, y = wavfile.read(path)
    if np.issubdtype(y.dtype, np.integer):
        y = y.astype(np.float32) / float(np.iinfo(y.dtype).max + 1)
    y = np.asarray(y, dtype=np.float32)
    y = y.mean(axis=1) if y.ndim == 2 else y
    if sample_rate is not None and int(sr) != int(sample_rate):
        y = _resample(y, sr, sample_rate)
    return y


def features_of(audio_paths, sample_rate, n_mfcc=13, dtype="float32"):
    """extract_mfcc of many files in one launch, left on the device (a ``Ragged`` batch)."""
    sigs = [_load(p, sample_rate) for p in audio_paths]
    return fe.librosa_mfcc_ragged(sigs, sample_rate, n_mfcc, zscore=True, dtype=dtype)


def extract_many(audio_paths, sample_rate, n_mfcc=13, dtype="float32", to_host=True):
    """extract_mfcc of many files in one launch -> (packed features, lengths)."""
    data = features_of(audio_paths, sample_rate, n_mfcc, dtype=dtype)
    return (data.X.cpu().numpy() if to_host else data.X), np.diff(data.offsets_host)


def extract_mfcc(audio_path, sample_rate, n_mfcc=13):
    """(T, n_mfcc) z-scored librosa-convention MFCCs of one WAV file,
    HMM_lib.py:11-15."""
    return extract_many([audio_path], sample_rate, n_mfcc)[0].astype(np.float64)


def train_hmm(models, model_name, audio_files, labels, sample_rate, i=0, *, model_dir="SavedModels",
              n_components=3, tol=1e-4, n_iter=2000, save=True, verbose=True, **hmm_kw):
    """One GaussianHMM per distinct label stored in ``models[label]`` and pickled as
    ``<label><model_name>``, HMM_lib.py:26-42.  The features of all files come from one
    front-end launch and stay on the device for the fits (the models compute in float64, the
    class default, on the float32 features widened there)."""
    print("Training started ...")
    data = features_of(audio_files, sample_rate)
    _rec._fit_per_label(GaussianHMM, data, labels, lambda label: f"{label}{model_name}", models,
                        model_dir, save,
                        dict(n_components=n_components, tol=tol, n_iter=n_iter, implementation="log",
                             verbose=verbose, **hmm_kw))


def recognize_many(models, audio_paths, sample_rate):
    """(labels, scores) of many files on the librosa-convention features: front-end,
    scoring and arg-max on the device, only labels and scores come back."""
    data = features_of(audio_paths, sample_rate)
    labels, scores = classify(models, data)
    names = list(models.keys())
    return [names[int(k)] for k in labels.cpu().numpy()], scores.cpu().numpy()


def recognize_lib(models, audio_path, sample_rate):
    """HMM_lib.py:50-53."""
    return recognize_many(models, [audio_path], sample_rate)[0][0]


def recognize(models, audio_path, sample_rate):
    """HMM_lib.py:45-48: the same models on the reference's own front-end."""
    return _rec.recognize(models, audio_path, sample_rate)


def load_models(model_name, model_dir="models"):
    """Every ``*<model_name>`` file of a directory, HMM_lib.py:67-81.  Only for files
    written by ``train_hmm`` of this package."""
    models = {}
    for file in os.listdir(model_dir):
        if file.endswith(model_name):
            label = file.replace(model_name, "")
            path = os.path.join(model_dir, file)
            with open(path, "rb") as f:
                models[label] = pickle.load(f)
            print(f"Model for {label} loaded from {path}")
    return models


def train_using_all_files(models, model_name, directory="cleaned_audios_3005", sample_rate=22050,
                          n_files=25, **kw):
    """HMM_lib.py:55-65 with the module-level ``models`` made an argument."""
    audio_files, labels = word_files(directory, n_files)
    train_hmm(models, model_name, audio_files, labels, sample_rate, **kw)


def test_trained_model(models, directory, sample_rate, folds_dir="Folds", lists=None):
    """HMM_lib.py:130-161 (recognize_lib on every validation sample): the loop and the
    printed lines of the other harness with this module's recogniser."""
    return _rec.test_trained_model(models, directory, sample_rate, folds_dir=folds_dir, lists=lists,
                                   recognizer=recognize_many)


test_trained_model.__test__ = False   # not a pytest test
