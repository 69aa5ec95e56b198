"""The librosa-convention front-end (extract_mfcc, HMM_lib.py:11-15) through the C
ABI against the CPU restatement of librosa 0.10's published algorithm
(oracle/mfcc_oracle.py).  PARITY UNPINNED: librosa is absent from the image and
the reference holds no output of that call; what is pinned is the agreement of
the HIP path with the restatement.

Tolerances: float64 to rounding (1e-9 of the largest cepstrum); float32 1e-4 of
the largest cepstrum (frames that hit the top_db clamp sit at -80 dB below the
maximum, where float32 log-mel values differ in the 6th digit)."""
import numpy as np
import pytest

from oracle import mfcc_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sra():
    import speech_recognition_amd as s
    return s


def _signals(seed, lengths):
    rng = np.random.default_rng(seed)
    out = []
    for n in lengths:
        t = np.arange(n) / 22050.0
        s = 0.1 * rng.standard_normal(n) + 0.1 * np.sin(2 * np.pi * 440.0 * t)
        if n > 9000:
            s[3000:9000] = 0.0          # digital silence: the amin floor and the top_db clamp act
        out.append(s)
    return out


LENGTHS = [20727, 13230, 512 * 20, 512 * 20 - 1, 2048, 1500, 1, 0, 52920]


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-9), ("float32", 1e-4)])
@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
def test_librosa_mfcc_matches_oracle(sra, dtype, tol, pad_mode):
    lengths = [n for n in LENGTHS if pad_mode == "constant" or n > 1024]
    sigs = _signals(3, lengths)
    if dtype == "float32":
        sigs = [s.astype(np.float32) for s in sigs]
    cfg = mo.LibrosaConfig(sample_rate=22050, pad_mode=pad_mode)
    f, lens = sra.mfcc.librosa_mfcc_batch(sigs, 22050, 13, dtype=dtype, pad_mode=pad_mode)
    assert list(lens) == [cfg.n_frames(n) for n in lengths]
    off = np.concatenate([[0], np.cumsum(lens)])
    for k, s in enumerate(sigs):
        ref = mo.librosa_mfcc(s, cfg)
        got = f[off[k]:off[k + 1]]
        assert got.shape == ref.shape
        if ref.size:
            assert np.max(np.abs(got - ref)) <= tol * np.max(np.abs(ref)), (k, lengths[k])


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-8), ("float32", 2e-3)])
def test_librosa_extract_zscored_and_other_shapes(sra, dtype, tol):
    """extract_mfcc's z-score on top, and a non-default shape (n_fft 1024, hop 220, 40
    filters, 20 cepstra, 16 kHz, no top_db)."""
    sigs = _signals(5, [30000, 7000, 16000])
    f, lens = sra.mfcc.librosa_mfcc_batch(sigs, 22050, 13, zscore=True, dtype=dtype)
    off = np.concatenate([[0], np.cumsum(lens)])
    cfg = mo.LibrosaConfig(sample_rate=22050)
    for k, s in enumerate(sigs):
        np.testing.assert_allclose(f[off[k]:off[k + 1]], mo.librosa_extract(s, cfg), rtol=0, atol=tol)
    cfg2 = mo.LibrosaConfig(sample_rate=16000, n_mfcc=20, n_fft=1024, hop_length=220, n_mels=40, top_db=0.0)
    f2, lens2 = sra.mfcc.librosa_mfcc_batch(sigs, 16000, 20, dtype=dtype, n_fft=1024, hop_length=220,
                                            n_mels=40, top_db=0.0)
    off2 = np.concatenate([[0], np.cumsum(lens2)])
    for k, s in enumerate(sigs):
        ref = mo.librosa_mfcc(s, cfg2)
        assert np.max(np.abs(f2[off2[k]:off2[k + 1]] - ref)) <= tol * 1e-1 * np.max(np.abs(ref))


def test_hmm_lib_harness_end_to_end(sra, tmp_path):
    """extract_mfcc -> train_hmm -> recognize_lib of the HMM_lib surface on written WAV
    files: features equal the oracle's, a saved model reloads, a training file is
    recognised as its own word."""
    from scipy.io import wavfile
    rng = np.random.default_rng(11)
    files, labels = [], []
    for w, f0 in (("alpha", 300.0), ("beta", 1500.0)):
        for k in range(4):
            n = 22050 // 2 + 500 * k
            t = np.arange(n) / 22050.0
            y = (0.2 * np.sin(2 * np.pi * (f0 + 200 * t) * t) + 0.02 * rng.standard_normal(n)).astype(np.float32)
            p = tmp_path / f"{w}_{k + 1}.wav"
            wavfile.write(str(p), 22050, y)
            files.append(str(p)); labels.append(w)
    L = sra.HMM_lib
    feats = L.extract_mfcc(files[0], 22050)
    y0 = wavfile.read(files[0])[1]
    np.testing.assert_allclose(feats, mo.librosa_extract(y0, mo.LibrosaConfig()), rtol=0, atol=2e-3)
    with pytest.raises(ValueError):
        L.extract_mfcc(files[0], 16000)          # stored at another rate: no resampling
    models = {}
    L.train_hmm(models, "_hmm_model_lib.pkl", files, labels, 22050, model_dir=str(tmp_path),
                n_components=2, n_iter=5, verbose=False, random_state=0)
    assert sorted(models) == ["alpha", "beta"]
    again = L.load_models("_hmm_model_lib.pkl", model_dir=str(tmp_path))
    assert sorted(again) == ["alpha", "beta"]
    assert L.recognize_lib(again, files[0], 22050) == "alpha"
    assert L.recognize_lib(again, files[-1], 22050) == "beta"
