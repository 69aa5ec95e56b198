"""Train / score harness with the reference's function names (HMM_final.py).

``extract_mfcc_calc`` (HMM_final.py:18), ``train_hmm`` (:26), ``recognize``
(:49), ``load_models`` (:71), ``test_trained_model`` (:134) keep their
arguments and results.  Under them everything is batched: all utterances of a
call go through one front-end launch and one scoring launch against all word
models at once (the reference loops over files and over the six models).
"""
import os
import pickle

import numpy as np

from . import SpeechUtil as su
from . import mfcc as fe
from .hmm import MyGaussianHMM
from .runtime import Context, HMMBank, Ragged, argmax_scores

WORDS = ["play_music", "stop_music", "turn_off_the_lights", "turn_on_the_lights",
         "what_time_is_it", "odessa"]          # HMM_final.py:64


def extract_mfcc_calc(audio_path, sample_rate, n_mfcc=13, dtype="float64"):
    """(T-4, 26) z-scored MFCC+delta of one WAV file, HMM_final.py:18-23."""
    signal = su.read_recording(str(audio_path).replace(".wav", ""), sample_rate)
    return fe.extract_features(signal, sample_rate, dtype=dtype)


def features_of(audio_paths, sample_rate, dtype="float64"):
    """Features of many files in one launch, left on the device: a ``Ragged`` batch."""
    sigs = [su.read_recording(str(p).replace(".wav", ""), sample_rate) for p in audio_paths]
    return fe.extract_ragged(sigs, sample_rate, dtype=dtype)


def extract_many(audio_paths, sample_rate, dtype="float64", to_host=True):
    """Features of many files in one launch -> (packed features, lengths); the features as
    a numpy array, or with to_host=False as the device tensor the kernels wrote."""
    data = features_of(audio_paths, sample_rate, dtype=dtype)
    return (data.X.cpu().numpy() if to_host else data.X), np.diff(data.offsets_host)


def _gather_utterances(data, idx):
    """The utterances ``idx`` of a device batch as one packed device tensor + their lengths
    (device-to-device copies; the features never visit the host)."""
    import torch
    off = data.offsets_host
    parts = [data.X[int(off[k]):int(off[k + 1])] for k in idx]
    X = torch.cat(parts) if len(parts) != 1 else parts[0]
    return X, [int(off[k + 1] - off[k]) for k in idx]


def _fit_per_label(model_cls, data, labels, make_name, models, model_dir, save, ctor_kw):
    """The label loop shared by the two harnesses' train_hmm (HMM_final.py:28-46,
    HMM_lib.py:28-42): a model per distinct label, fitted on that label's utterances (taken
    from the device batch), stored in ``models`` and pickled."""
    labels = np.asarray(labels)
    for label in np.unique(labels):
        idx = [k for k in range(len(labels)) if labels[k] == label]
        X, L = _gather_utterances(data, idx)
        model = model_cls(**ctor_kw)
        model.fit(X, L)
        models[label] = model
        if save:
            os.makedirs(model_dir, exist_ok=True)
            path = os.path.join(model_dir, make_name(label))
            with open(path, "wb") as f:
                pickle.dump(model, f)
            print(f"Model for {label} saved at {path}")


def train_hmm(models, audio_files, labels, sample_rate, i=-1, *, model_dir="SavedModels",
              n_components=3, tol=1e-4, n_iter=2000, save=True, verbose=True, **hmm_kw):
    """One model per distinct label, trained on that label's files and stored in
    ``models[label]`` (and pickled), HMM_final.py:26-46.  Like the reference's, the
    models are ``MyGaussianHMM(n_components=3, tol=1e-4, n_iter=2000,
    implementation="scaling")`` and ``fit`` re-estimates start and transition
    probabilities only; ``mstep="full"`` (and any other constructor argument, e.g.
    ``random_state``, ``dtype``) goes through ``hmm_kw``.  The features of all files come
    from one front-end launch and stay on the device for the fits."""
    print("Training started ...")
    data = features_of(audio_files, sample_rate, dtype=hmm_kw.get("dtype", "float64"))
    _fit_per_label(MyGaussianHMM, data, labels,
                   lambda label: f"{label}_hmm_model.pkl" if i == -1 else f"{label}_{i}_hmm_model.pkl",
                   models, model_dir, save,
                   dict(n_components=n_components, tol=tol, n_iter=n_iter, implementation="scaling",
                        verbose=verbose, **hmm_kw))


_banks = {}          # (context, parameter bytes of the models) -> HMMBank; a handful of entries


def bank_of(models, ctx=None):
    """The word models of a dict as one device bank, in insertion order.  Cached on the
    models' parameter values, so that recognising file after file with the same models
    uploads them once."""
    ctx = ctx or Context.default()
    ms = list(models.values()) if isinstance(models, dict) else list(models)
    key = (id(ctx),) + tuple(m._param_key() for m in ms)
    bank = _banks.get(key)
    if bank is None:
        if len(_banks) >= 8:
            _banks.pop(next(iter(_banks)))
        bank = _banks[key] = HMMBank.from_models(ctx, ms)
    return bank


def score_matrix(models, feats, lengths=None, dtype="float64", ctx=None, to_host=True):
    """(U, n_models) log-likelihoods in ``models`` insertion order: the dict
    comprehension of HMM_final.py:51 for every utterance at once.  feats: packed features
    (host array or device tensor) with ``lengths``, or a ``Ragged`` batch."""
    ctx = ctx or Context.default()
    ms = list(models.values())
    shapes = {(m.n_components, m.means_.shape[1]) for m in ms}
    data = feats if isinstance(feats, Ragged) else Ragged(ctx, feats, lengths, dtype=dtype)
    if len(shapes) == 1:
        scores = bank_of(models, ctx).score(data)
    else:
        import torch
        scores = torch.stack([bank_of([m], ctx).score(data)[:, 0] for m in ms], dim=1).contiguous()
    return scores.cpu().numpy() if to_host else scores


def classify(models, data, ctx=None):
    """(labels (U,) int32, scores (U, n_models) float64) on the device for a ``Ragged``
    batch: scores against every model, then max(scores, key=scores.get) of
    HMM_final.py:52 as an arg-max kernel (first best model in insertion order)."""
    ctx = ctx or data.ctx
    scores = score_matrix(models, data, ctx=ctx, to_host=False)
    return argmax_scores(ctx, scores), scores


def recognize_device(models, audio, sample_offsets, sample_rate, dtype="float32", ctx=None, **front_end):
    """Audio that is already on the device -> (labels, scores, features), all on the device:
    front-end, scoring against every model and the arg-max as one chain of launches with no
    host copy in between.  audio: packed 1-D device tensor, sample_offsets: host int64
    [U+1]; front_end: keyword arguments of ``MfccPlan`` (default: the reference's mfcc.py)."""
    ctx = ctx or Context.default()
    plan = fe._plan(ctx, sample_rate, **front_end)
    data = plan.run_ragged(audio, sample_offsets, dtype=dtype)
    labels, scores = classify(models, data, ctx)
    return labels, scores, data


def recognize_many(models, audio_paths, sample_rate, dtype="float64"):
    """(labels, (U, n_models) scores) of many files: one front-end launch, one scoring
    launch, one arg-max launch; only the labels and the score matrix come back."""
    data = features_of(audio_paths, sample_rate, dtype=dtype)
    labels, scores = classify(models, data)
    names = list(models.keys())
    return [names[int(k)] for k in labels.cpu().numpy()], scores.cpu().numpy()


def recognize(models, audio_path, sample_rate, dtype="float64"):
    """Label whose model gives the utterance the highest log-likelihood,
    HMM_final.py:49-52."""
    return recognize_many(models, [audio_path], sample_rate, dtype=dtype)[0][0]


def load_models(model_dir="models"):
    """Unpickle every ``*_hmm_model.pkl`` of a directory, HMM_final.py:71-85.
    Only for files written by ``train_hmm`` of this package."""
    models = {}
    for file in os.listdir(model_dir):
        if file.endswith("_hmm_model.pkl"):
            label = file.replace("_hmm_model.pkl", "")
            path = os.path.join(model_dir, file)
            with open(path, "rb") as f:
                models[label] = pickle.load(f)
            print(f"Model for {label} loaded from {path}")
    return models


def train_using_all_files(models, directory="cleaned_audios_3005", sample_rate=22050, n_files=25,
                          **kw):
    """HMM_final.py:59-69 with the module-level ``models`` made an argument."""
    audio_files, labels = word_files(directory, n_files)
    train_hmm(models, audio_files, labels, sample_rate, **kw)


def _word_of(sample):
    stem = sample.replace(".wav", "")
    return stem[: stem.rfind("_")]


def word_files(directory, n_files=25):
    """(files, labels) of the training set both harnesses use, HMM_final.py:59-69."""
    audio_files, labels = [], []
    for word in WORDS:
        for k in range(n_files):
            audio_files.append(f"{directory}/{word}_{k + 1}.wav")
            labels.append(word)
    return audio_files, labels


def test_trained_model(models, directory, sample_rate, folds_dir="Folds", lists=None,
                       dtype="float64", recognizer=None):
    """Score the validation lists and print right / wrong per list,
    HMM_final.py:134-165 (fold 1 is listed twice there).  Returns the
    (right, wrong) pairs.  recognizer(models, paths, sample_rate) -> (labels, scores):
    default ``recognize_many`` with ``dtype``."""
    if recognizer is None:
        def recognizer(models, paths, sample_rate):
            return recognize_many(models, paths, sample_rate, dtype=dtype)
    lists = lists or ["validation 1.txt", "validation 1.txt", "validation 2.txt",
                      "validation 3.txt", "validation 4.txt", "validation 5.txt"]
    results = []
    for name in lists:
        with open(os.path.join(folds_dir, name)) as f:
            samples = [l.strip() for l in f if l.strip()]
        print(name)
        pred, _ = recognizer(models, [f"{directory}/{s}" for s in samples], sample_rate)
        right = 0
        for s, p in zip(samples, pred):
            print(_word_of(s))
            print(p)
            right += int(p == _word_of(s))
        print(f"right: {right}; wrong: {len(samples) - right}")
        results.append((right, len(samples) - right))
    return results


test_trained_model.__test__ = False   # not a pytest test


def cross_validate(feats, lengths, names, folds, words=None, *, n_components=3, n_iter=8,
                   init_fn=None, mstep="full", dtype="float64", ctx=None, random_state=0):
    """The k-fold experiment of HMM_final.py:93-114 / :134-165 as ONE batched job:
    a word model per (fold, word) -- 30 for the reference's 5 folds x 6 words --
    all trained together (one E-step launch per EM iteration over every fold's
    training utterances), then every fold's validation utterances scored against
    that fold's models in one launch.

    feats (sum T, D) packed features of all recordings, lengths (n,), names (n,)
    recording names ``<word>_<k>``; folds: list of (train_names, validation_names).
    init_fn(X_word, n_components) -> (startprob, transmat, means, covars) gives
    each model's start (default: the class's ``init``, seeded per model).
    Returns a dict with per-fold ``pred``, ``right``, ``scores`` (n_val, n_words)
    and ``hist`` (n_folds, n_words, n_iter) log-likelihood histories, ``bank``."""
    from .parallel import BaumWelchTrainer
    ctx = ctx or Context.default()
    names = [str(n) for n in names]
    lengths = np.asarray(lengths, dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lengths)])
    index = {n: i for i, n in enumerate(names)}
    words = list(words) if words is not None else sorted({_word_of(n) for n in names})
    W, F = len(words), len(folds)
    feats = np.asarray(feats)

    def utt(n):
        i = index[str(n).replace(".wav", "")]
        return feats[off[i]:off[i + 1]]

    inits, xs, um = [], [], []
    for f, (train, _) in enumerate(folds):
        for w_i, w in enumerate(words):
            sel = [n for n in train if _word_of(str(n)) == w]
            if not sel:
                raise ValueError(f"fold {f + 1} has no training recording of {w!r}")
            Xw = np.vstack([utt(n) for n in sel]).astype(np.float64)
            if init_fn is not None:
                inits.append(tuple(np.asarray(a, dtype=np.float64) for a in init_fn(Xw, n_components)))
            else:
                m = MyGaussianHMM(n_components=n_components, verbose=False,
                                  random_state=random_state + f * W + w_i)
                m.init(Xw, [len(utt(n)) for n in sel])
                inits.append((m.startprob_, m.transmat_, m.means_, m._covars_))
            xs += [utt(n) for n in sel]
            um += [f * W + w_i] * len(sel)
    bank = HMMBank(ctx, np.stack([i[0] for i in inits]), np.stack([i[1] for i in inits]),
                   np.stack([i[2] for i in inits]), np.stack([i[3] for i in inits]))
    data = Ragged(ctx, np.vstack(xs), [len(x) for x in xs], dtype=dtype)
    trainer = BaumWelchTrainer(bank, data, np.asarray(um, dtype=np.int32), mstep=mstep)
    hist = np.stack([trainer.step().cpu().numpy() for _ in range(n_iter)], axis=1)
    vx, vfold = [], []
    for f, (_, val) in enumerate(folds):
        vx += [utt(n) for n in val]
        vfold += [f] * len(val)
    vdata = Ragged(ctx, np.vstack(vx), [len(x) for x in vx], dtype=dtype)
    all_scores = bank.score(vdata).cpu().numpy()                    # (n_val_total, F*W)
    out = {"hist": hist.reshape(F, W, n_iter), "bank": bank, "words": words, "folds": []}
    pos = 0
    for f, (_, val) in enumerate(folds):
        sc = all_scores[pos:pos + len(val), f * W:(f + 1) * W]
        pos += len(val)
        pred = [words[int(np.argmax(r))] for r in sc]
        right = sum(p == _word_of(str(n)) for p, n in zip(pred, val))
        out["folds"].append({"scores": sc, "pred": pred, "right": right, "wrong": len(val) - right})
    return out
