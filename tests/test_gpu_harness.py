"""The HMM_final.py call surface end to end on the GPU: WAV files -> read_recording ->
GPU MFCC -> GPU HMM -> label, through extract_mfcc_calc / calculate_mfcc_for_onerecording /
recognize / test_trained_model / train_hmm / load_models, against what the reference's own
functions returned and printed for the same files (tests/golden/harness_golden.npz, made by
tests/golden/make_golden.py harness: the 24 fold-1 validation recordings as float32 WAVs).

Tolerances: float64 is the validation arithmetic (agreement to rounding); float32 must keep
un-normalised cepstra within 1e-4 of each coefficient row's RMS, log-likelihoods within 1e-4
relative, and every predicted label identical."""
import contextlib
import io
import os
import pickle

import numpy as np
import pytest
from scipy.io import wavfile

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def sra():
    import speech_recognition_amd as s
    return s


@pytest.fixture(scope="module")
def hg():
    return np.load(os.path.join(GOLDEN, "harness_golden.npz"))


@pytest.fixture(scope="module")
def site(hg, tmp_path_factory):
    """A working directory like the reference's: wav/<name>.wav, Folds/validation k.txt."""
    root = tmp_path_factory.mktemp("harness")
    (root / "wav").mkdir()
    (root / "Folds").mkdir()
    so = np.concatenate([[0], np.cumsum(hg["sig_lengths"])])
    for i, n in enumerate(hg["names"]):
        wavfile.write(str(root / "wav" / f"{n}.wav"), int(hg["sr"]), hg["signals"][so[i]:so[i + 1]])
    for k in range(1, 6):
        (root / "Folds" / f"validation {k}.txt").write_text("".join(f"{n}.wav\n" for n in hg[f"list{k}"]))
    return root


def _fold1_models(sra, dtype="float64"):
    g = np.load(os.path.join(GOLDEN, "folds_golden.npz"))
    models = {}
    for w in g["words"]:
        m = sra.MyGaussianHMM(n_components=3, verbose=False, dtype=dtype)
        p = f"f1_{w}_"
        m.startprob_, m.transmat_ = g[p + "pi"].copy(), g[p + "A"].copy()
        m.means_, m._covars_ = g[p + "mu"].copy(), g[p + "var"].copy()
        m.n_features = m.means_.shape[1]
        models[str(w)] = m
    return models


def test_read_recording_and_features_of_files(sra, hg, site):
    """a1 + a8: SpeechUtil.read_recording, mfcc.calculate_mfcc_for_onerecording (26, T-4) and
    HMM_final.extract_mfcc_calc (T-4, 26) per file."""
    sr = int(hg["sr"])
    so = np.concatenate([[0], np.cumsum(hg["sig_lengths"])])
    fo = np.concatenate([[0], np.cumsum(hg["lengths"])])
    worst = {"feat64": 0.0, "z64": 0.0, "feat32_row_rms": 0.0, "z32": 0.0}
    for i, n in enumerate(hg["names"]):
        path = str(site / "wav" / f"{n}.wav")
        s = sra.SpeechUtil.read_recording(path[:-4], sr)
        assert s.dtype == np.float32 and np.array_equal(s, hg["signals"][so[i]:so[i + 1]])
        ref_feat = hg["feat"][fo[i]:fo[i + 1]].T           # (26, T-4)
        ref_z = hg["z"][fo[i]:fo[i + 1]]
        feat = sra.mfcc.calculate_mfcc_for_onerecording(False, path, sample_rate=sr)
        z = sra.recognizer.extract_mfcc_calc(path, sr)
        assert feat.shape == ref_feat.shape and z.shape == ref_z.shape
        worst["feat64"] = max(worst["feat64"], np.max(np.abs(feat - ref_feat)) / np.max(np.abs(ref_feat)))
        worst["z64"] = max(worst["z64"], np.max(np.abs(z - ref_z)))
        feat32 = sra.mfcc.calculate_mfcc_for_onerecording(False, path, sample_rate=sr, dtype="float32")
        rms = np.sqrt(np.mean(ref_feat ** 2, axis=1, keepdims=True))
        worst["feat32_row_rms"] = max(worst["feat32_row_rms"], np.max(np.abs(feat32 - ref_feat) / rms))
        z32 = sra.recognizer.extract_mfcc_calc(path, sr, dtype="float32")
        worst["z32"] = max(worst["z32"], np.max(np.abs(z32 - ref_z)))
    print("harness feature errors:", worst)
    assert worst["feat64"] < 1e-10 and worst["z64"] < 1e-8
    assert worst["feat32_row_rms"] < 1e-4
    # z-scored values are O(1): the cepstral error divided by the column's standard deviation
    # (measured on MI355X: 8.9e-6 of a row's RMS before, 1.0e-5 absolute after the z-score)
    assert worst["z32"] < 1e-4


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_recognize_and_test_trained_model(sra, hg, site, dtype):
    """a22 + the validation loop: labels, scores, the printed text and the right / wrong
    counts of the reference's recognize / test_trained_model, from WAV files."""
    sr = int(hg["sr"])
    models = _fold1_models(sra, dtype)
    assert list(models) == [str(w) for w in hg["words"]]
    paths = [str(site / "wav" / f"{n}.wav") for n in hg["names"]]
    labels = [sra.recognizer.recognize(models, p, sr, dtype=dtype) for p in paths]
    assert labels == [str(l) for l in hg["labels"]]
    pred, scores = sra.recognizer.recognize_many(models, paths, sr, dtype=dtype)
    assert pred == labels
    rel = np.max(np.abs(scores - hg["scores"]) / np.abs(hg["scores"]))
    print(f"harness scores ({dtype}): max relative error {rel:.2e}")
    assert rel < (1e-9 if dtype == "float64" else 1e-4)
    cwd = os.getcwd()
    os.chdir(site)
    try:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = sra.recognizer.test_trained_model(models, "wav", sr, dtype=dtype)
    finally:
        os.chdir(cwd)
    assert buf.getvalue() == str(hg["test_trained_model_stdout"])
    want = [tuple(int(v) for v in l.replace("right:", "").replace("wrong:", "").split(";"))
            for l in str(hg["test_trained_model_stdout"]).splitlines() if l.startswith("right")]
    assert res == want and len(res) == 6


def test_train_hmm_reproduces_the_reference_run(sra, hg, site):
    """a23 + a19: train_hmm from WAV files with the reference's seeds -- the same Dirichlet
    draws and data rows at init, the same transitions-only EM, the same stopping iteration,
    the same pickle names; load_models gives models that score like the ones in memory."""
    sr = int(hg["sr"])
    names = [str(n) for n in hg["names"]]
    files = [str(site / "wav" / f"{n}.wav") for n in names]
    labels = [n[: n.rfind("_")] for n in names]
    model_dir = str(site / "SavedModels")
    models = {}
    np.random.seed(int(hg["global_seed"]))
    with contextlib.redirect_stdout(io.StringIO()) as out:
        sra.recognizer.train_hmm(models, files, labels, sr, model_dir=model_dir,
                                 random_state=int(hg["seed"]), verbose=False)
    assert list(models) == [str(w) for w in hg["train_order"]]
    assert sorted(os.listdir(model_dir)) == [str(f) for f in hg["train_files"]]
    assert out.getvalue().replace(model_dir, "SavedModels") == str(hg["train_stdout"])
    fo = np.concatenate([[0], np.cumsum(hg["lengths"])])
    for w, m in models.items():
        p = f"train_{w}_"
        assert m.monitor_.iter == int(hg[p + "iter"]), w
        np.testing.assert_allclose(list(m.monitor_.history), hg[p + "hist"], rtol=1e-10)
        np.testing.assert_allclose(m.startprob_, hg[p + "pi"], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(m.transmat_, hg[p + "A"], rtol=1e-8, atol=1e-12)
        # means are rows of the data, covariances its variance: untouched by this M-step
        np.testing.assert_allclose(m.means_, hg[p + "mu"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(m._covars_, hg[p + "var"], rtol=1e-9)
        for attr in ("startprob_", "transmat_", "means_", "_covars_", "n_features", "n_components",
                     "monitor_", "tol", "n_iter", "implementation"):
            assert hasattr(m, attr), attr
    loaded = sra.recognizer.load_models(model_dir)
    assert sorted(loaded) == sorted(models)
    z = hg["z"]
    for w in models:
        got = np.array([loaded[w].score(z[fo[i]:fo[i + 1]]) for i in range(0, 24, 5)])
        mem = np.array([models[w].score(z[fo[i]:fo[i + 1]]) for i in range(0, 24, 5)])
        assert np.array_equal(got, mem)
        np.testing.assert_allclose(got, hg[f"train_{w}_score"][::5], rtol=1e-9)
    # the single-file form used by the fold loop: <label>_<i>_hmm_model.pkl
    one = {}
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        sra.recognizer.train_hmm(one, files[:4], labels[:4], sr, 3, model_dir=model_dir, n_iter=2,
                                 random_state=0, verbose=False)
    assert os.path.exists(os.path.join(model_dir, f"{labels[0]}_3_hmm_model.pkl"))
    with open(os.path.join(model_dir, f"{labels[0]}_3_hmm_model.pkl"), "rb") as f:
        again = pickle.load(f)
    assert again.monitor_.iter == 2 and again.mstep == "transitions_only"


def test_train_hmm_float32_history(sra, hg, site):
    """The same training in the float32 arithmetic: the log-likelihood history stays within
    1e-4 relative of the reference's over the first iterations."""
    sr = int(hg["sr"])
    names = [str(n) for n in hg["names"]]
    files = [str(site / "wav" / f"{n}.wav") for n in names]
    labels = [n[: n.rfind("_")] for n in names]
    models = {}
    np.random.seed(int(hg["global_seed"]))
    with contextlib.redirect_stdout(io.StringIO()):
        sra.recognizer.train_hmm(models, files, labels, sr, save=False, n_iter=5,
                                 random_state=int(hg["seed"]), verbose=False, dtype="float32")
    for w, m in models.items():
        h = np.array(m.monitor_.history)
        ref = hg[f"train_{w}_hist"][:len(h)]
        assert len(h) == min(5, int(hg[f"train_{w}_iter"]))
        np.testing.assert_allclose(h, ref, rtol=1e-4)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_device_resident_chain_equals_the_two_step_path(sra, hg, site, dtype):
    """recognize_many keeps the features on the device from the front-end kernels to the
    arg-max kernel.  The same files taken the long way round -- features to the host, back
    into a batch, scores to the host, numpy's arg-max -- give the same scores and labels bit
    for bit; so does the chain started from audio that is already on the device."""
    import torch
    sr = int(hg["sr"])
    rec = sra.recognizer
    models = _fold1_models(sra, dtype)
    names = list(models)
    paths = [str(site / "wav" / f"{n}.wav") for n in hg["names"]]
    pred, scores = rec.recognize_many(models, paths, sr, dtype=dtype)
    # the long way round
    feats, lengths = rec.extract_many(paths, sr, dtype=dtype)
    assert isinstance(feats, np.ndarray)
    scores2 = rec.score_matrix(models, feats, lengths, dtype=dtype)
    assert np.array_equal(scores, scores2)
    assert pred == [names[int(np.argmax(r))] for r in scores2]
    # features left on the device are the tensor the kernels wrote, and a batch made of it
    # holds that very memory
    data = rec.features_of(paths, sr, dtype=dtype)
    assert data.X.is_cuda and np.array_equal(data.X.cpu().numpy(), feats)
    ctx = sra.Context.default()
    again = sra.Ragged(ctx, data.X, lengths, dtype=dtype)
    assert again.X.data_ptr() == data.X.data_ptr()
    # from device audio
    so = np.concatenate([[0], np.cumsum(hg["sig_lengths"])])
    audio = torch.as_tensor(hg["signals"], device=ctx.device)
    labels, sc_dev, d2 = rec.recognize_device(models, audio, so, sr, dtype=dtype)
    assert labels.is_cuda and labels.dtype == torch.int32 and sc_dev.is_cuda
    assert np.array_equal(sc_dev.cpu().numpy(), scores)
    assert [names[int(k)] for k in labels.cpu().numpy()] == pred == [str(l) for l in hg["labels"]]
    # ties go to the first model in insertion order, as max(scores, key=scores.get) does
    tie = torch.tensor([[1.0, 3.0, 3.0], [2.0, 2.0, 2.0], [-1.0, -2.0, -0.5], [float("-inf")] * 3],
                       dtype=torch.float64, device=ctx.device)
    assert sra.runtime.argmax_scores(ctx, tie).cpu().tolist() == [1, 0, 2, 0]


def test_fit_from_device_features_equals_fit_from_host_features(sra, hg):
    """MyGaussianHMM.fit on a device tensor (what train_hmm hands it) against the same fit on
    the host copy of those features: same random draws, same rows as means, covariances from
    the column-moments kernel equal to np.cov's diagonal to rounding, same history."""
    import torch
    ctx = sra.Context.default()
    z = hg["z"]
    lengths = [int(v) for v in hg["lengths"]]
    for dtype, rt in (("float64", 1e-12), ("float32", 1e-6)):
        a = sra.MyGaussianHMM(n_components=3, n_iter=6, tol=1e-9, verbose=False, random_state=3, dtype=dtype)
        b = sra.MyGaussianHMM(n_components=3, n_iter=6, tol=1e-9, verbose=False, random_state=3, dtype=dtype)
        np.random.seed(11)
        a.fit(z, lengths)
        np.random.seed(11)
        b.fit(torch.as_tensor(z, device=ctx.device), lengths)
        assert np.array_equal(a.means_, b.means_)
        np.testing.assert_allclose(b._covars_, a._covars_, rtol=1e-13)
        np.testing.assert_allclose(list(b.monitor_.history), list(a.monitor_.history), rtol=rt)
        np.testing.assert_allclose(b.transmat_, a.transmat_, rtol=1e-6, atol=1e-12)
    # the class-surface views of the packed statistics record
    st = a.initialize_sufficient_statistics()
    assert st["nobs"] == 0 and st["obs**2"].shape == (3, z.shape[1])
    x0 = z[:lengths[0]]
    a.accumulate_sufficient_statistics(st, x0)
    one, _ = a.do_estep(x0, [lengths[0]])
    assert st["nobs"] == 1 and np.array_equal(st["trans"], one["trans"])
    np.testing.assert_allclose(a.compute_likelihood(x0), np.exp(a.compute_log_likelihood(x0)))
    assert a.score_log(x0, compute_posteriors=False)[0] == a.score_scaling(x0, compute_posteriors=False)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_fit_with_graph_replay_equals_launch_by_launch(sra, dtype):
    """use_graph=True replays the EM iteration as two captured graphs from the seventh iteration
    on: same kernels in the same order, so history and parameters are the same bits."""
    rng = np.random.default_rng(21)
    lengths = rng.integers(40, 120, size=30)
    z = rng.standard_normal((int(lengths.sum()), 13))
    fits = []
    for use_graph in (False, True):
        m = sra.MyGaussianHMM(n_components=3, n_iter=25, tol=-np.inf, verbose=False, random_state=5,
                              dtype=dtype, mstep="full", use_graph=use_graph)
        np.random.seed(17)
        m.fit(z, lengths)
        fits.append(m)
    a, b = fits
    assert a.monitor_.iter == b.monitor_.iter == 25
    assert list(a.monitor_.history) == list(b.monitor_.history)
    for name in ("startprob_", "transmat_", "means_", "_covars_"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
