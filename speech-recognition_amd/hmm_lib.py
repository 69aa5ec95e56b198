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
from .hmm import GaussianHMM
from .recognizer import WORDS, _word_of, extract_mfcc_calc, score_matrix   # noqa: F401


def _load(audio_path, sample_rate):
    """librosa.load(path, sr=sample_rate): float32 mono in [-1, 1).  A file stored at
    another rate would be resampled by librosa (soxr); that is not provided here."""
    path = str(audio_path)
    if not path.endswith(".wav"):
        path += ".wav"
    sr, y = wavfile.read(path)
    if sample_rate is not None and int(sr) != int(sample_rate):
        raise ValueError(f"{path} is stored at {sr} Hz, not {sample_rate} Hz: resampling is "
                         "not provided")
    if np.issubdtype(y.dtype, np.integer):
        y = y.astype(np.float32) / float(np.iinfo(y.dtype).max + 1)
    y = np.asarray(y, dtype=np.float32)
    return y.mean(axis=1) if y.ndim == 2 else y


def extract_many(audio_paths, sample_rate, n_mfcc=13, dtype="float32"):
    """extract_mfcc of many files in one launch -> (packed features, lengths)."""
    sigs = [_load(p, sample_rate) for p in audio_paths]
    return fe.librosa_mfcc_batch(sigs, sample_rate, n_mfcc, zscore=True, dtype=dtype)


def extract_mfcc(audio_path, sample_rate, n_mfcc=13):
    """(T, n_mfcc) z-scored librosa-convention MFCCs of one WAV file,
    HMM_lib.py:11-15."""
    return extract_many([audio_path], sample_rate, n_mfcc)[0].astype(np.float64)


def train_hmm(models, model_name, audio_files, labels, sample_rate, i=0, *, model_dir="SavedModels",
              n_components=3, tol=1e-4, n_iter=2000, save=True, verbose=True, **hmm_kw):
    """One GaussianHMM per distinct label stored in ``models[label]`` and pickled as
    ``<label><model_name>``, HMM_lib.py:26-42."""
    print("Training started ...")
    labels = np.asarray(labels)
    feats, lengths = extract_many(audio_files, sample_rate)
    off = np.concatenate([[0], np.cumsum(lengths)])
    for label in np.unique(labels):
        idx = [k for k in range(len(labels)) if labels[k] == label]
        X = np.vstack([feats[off[k]:off[k + 1]] for k in idx]).astype(np.float64)
        model = GaussianHMM(n_components=n_components, tol=tol, n_iter=n_iter,
                            implementation="log", verbose=verbose, **hmm_kw)
        model.fit(X, [int(lengths[k]) for k in idx])
        models[label] = model
        if save:
            os.makedirs(model_dir, exist_ok=True)
            path = os.path.join(model_dir, f"{label}{model_name}")
            with open(path, "wb") as f:
                pickle.dump(model, f)
            print(f"Model for {label} saved at {path}")


def recognize_many(models, audio_paths, sample_rate):
    feats, lengths = extract_many(audio_paths, sample_rate)
    scores = score_matrix(models, feats, lengths, dtype="float32")
    names = list(models.keys())
    return [names[int(np.argmax(row))] for row in scores], scores


def recognize_lib(models, audio_path, sample_rate):
    """HMM_lib.py:50-53."""
    return recognize_many(models, [audio_path], sample_rate)[0][0]


def recognize(models, audio_path, sample_rate):
    """HMM_lib.py:45-48: the same models on the reference's own front-end."""
    from .recognizer import recognize as _recognize
    return _recognize(models, audio_path, sample_rate)


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
    audio_files, labels = [], []
    for word in WORDS:
        for k in range(n_files):
            audio_files.append(f"{directory}/{word}_{k + 1}.wav")
            labels.append(word)
    train_hmm(models, model_name, audio_files, labels, sample_rate, **kw)


def test_trained_model(models, directory, sample_rate, folds_dir="Folds", lists=None):
    """HMM_lib.py:130-161 (recognize_lib on every validation sample)."""
    lists = lists or ["validation 1.txt", "validation 1.txt", "validation 2.txt",
                      "validation 3.txt", "validation 4.txt", "validation 5.txt"]
    results = []
    for name in lists:
        with open(os.path.join(folds_dir, name)) as f:
            samples = [l.strip() for l in f if l.strip()]
        print(name)
        pred, _ = recognize_many(models, [f"{directory}/{s}" for s in samples], sample_rate)
        right = 0
        for s, p in zip(samples, pred):
            print(_word_of(s))
            print(p)
            right += int(p == _word_of(s))
        print(f"right: {right}; wrong: {len(samples) - right}")
        results.append((right, len(samples) - right))
    return results


test_trained_model.__test__ = False   # not a pytest test
