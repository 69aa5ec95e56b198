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


def _row_rms_error(got, ref):
    """Largest |got - ref| of each cepstral coefficient in units of that coefficient's RMS over
    the utterance (the measure of the reference front-end's tests; a coefficient that is zero
    throughout is measured against the largest RMS; an utterance of fewer than 8 frames, where
    a coefficient's RMS can be next to nothing by chance, against its largest cepstrum)."""
    if ref.shape[0] < 8:
        return float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
    rms = np.sqrt(np.mean(np.square(ref), axis=0))
    rms = np.where(rms > 0, rms, max(float(rms.max()), 1e-300))
    return float(np.max(np.abs(got - ref) / rms))


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
    worst = 0.0
    for k, s in enumerate(sigs):
        ref = mo.librosa_mfcc(s, cfg)
        got = f[off[k]:off[k + 1]]
        assert got.shape == ref.shape
        if ref.size:
            worst = max(worst, _row_rms_error(got, ref))
    print(f"librosa conventions {dtype} {pad_mode}: largest error / coefficient RMS {worst:.2e}")
    assert worst <= tol


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
        assert _row_rms_error(f2[off2[k]:off2[k + 1]], ref) <= tol * 1e-1


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
    # stored at another rate: resampled on load (polyphase filter; librosa's soxr is not in the
    # image, so only the length librosa.load would give and the features' shape are checked)
    f16 = L.extract_mfcc(files[0], 16000)
    n16 = -(-len(y0) * 16000 // 22050)
    assert f16.shape == (1 + n16 // 512, 13) and np.isfinite(f16).all()
    models = {}
    L.train_hmm(models, "_hmm_model_lib.pkl", files, labels, 22050, model_dir=str(tmp_path),
                n_components=2, n_iter=5, verbose=False, random_state=0)
    assert sorted(models) == ["alpha", "beta"]
    again = L.load_models("_hmm_model_lib.pkl", model_dir=str(tmp_path))
    assert sorted(again) == ["alpha", "beta"]
    assert L.recognize_lib(again, files[0], 22050) == "alpha"
    assert L.recognize_lib(again, files[-1], 22050) == "beta"


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-9), ("float32", 2e-4)])
def test_librosa_conventions_random_configurations(sra, dtype, tol):
    """40 seeded random configurations: transform sizes 64 .. 4096 (odd and even stage
    counts of the LDS transform), hops, filter and cepstrum counts, both paddings, clamp on
    or off, ragged batches with empty and one-sample utterances."""
    rng = np.random.default_rng(17)
    for case in range(40):
        n_fft = int(rng.choice([64, 128, 256, 512, 1024, 2048, 4096]))
        hop = int(rng.integers(max(n_fft // 8, 1), n_fft + 1))
        n_mels = int(rng.integers(8, min(128, n_fft // 4) + 1))
        n_mfcc = int(rng.integers(2, min(32, n_mels) + 1))
        sr = int(rng.choice([8000, 16000, 22050, 44100]))
        pad_mode = str(rng.choice(["constant", "reflect"]))
        top_db = float(rng.choice([0.0, 40.0, 80.0]))
        zs = bool(rng.integers(0, 2))
        lo = n_fft // 2 + 1 if pad_mode == "reflect" else 0
        lens = [int(rng.integers(lo, lo + 4 * n_fft + 3000)) for _ in range(int(rng.integers(1, 6)))]
        if pad_mode == "constant":
            lens += [0, 1]
        sigs = []
        for n in lens:
            t = np.arange(n) / sr
            s = 0.05 * rng.standard_normal(n) + 0.2 * np.sin(2 * np.pi * rng.uniform(80, 3000) * t)
            if n > 2000 and rng.random() < 0.5:
                s[n // 4: n // 2] = 0.0
            sigs.append(s.astype(np.float32).astype(np.float64) if dtype == "float32" else s)
        cfg = mo.LibrosaConfig(sample_rate=sr, n_mfcc=n_mfcc, n_fft=n_fft, hop_length=hop, n_mels=n_mels,
                               pad_mode=pad_mode, top_db=top_db, zscore=zs)
        f, ln = sra.mfcc.librosa_mfcc_batch(sigs, sr, n_mfcc, zscore=zs, dtype=dtype, n_fft=n_fft,
                                            hop_length=hop, n_mels=n_mels, pad_mode=pad_mode, top_db=top_db)
        tag = f"case {case}: n_fft={n_fft} hop={hop} n_mels={n_mels} n_mfcc={n_mfcc} sr={sr} {pad_mode} top_db={top_db} z={zs} lens={lens}"
        assert list(ln) == [cfg.n_frames(n) for n in lens], tag
        off = np.concatenate([[0], np.cumsum(ln)])
        for k, s in enumerate(sigs):
            ref = mo.librosa_extract(s, cfg)
            got = f[off[k]:off[k + 1]]
            assert got.shape == ref.shape, tag
            if ref.size == 0:
                continue
            if zs:
                # z-scores divide by a column's spread: compare where the column has one
                ok = mo.librosa_mfcc(s, cfg).std(axis=0) > 1e-6 * np.abs(mo.librosa_mfcc(s, cfg)).max()
                err = np.abs(got - ref)[:, ok].max() if ok.any() else 0.0
                assert err <= (1e-6 if dtype == "float64" else 2e-2), (tag, k, err)
            else:
                assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), (tag, k)


@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
@pytest.mark.parametrize("n_mels,n_mfcc,top_db", [(128, 13, 80.0), (128, 20, 80.0), (64, 32, 0.0),
                                                  (40, 13, 30.0), (8, 4, 80.0)])
def test_register_transform_at_2048_points_equals_generic_kernels(sra, pad_mode, n_mels, n_mfcc, top_db):
    """The 2048-point register-FFT kernel (float32, centred frames) against the shape-generic
    kernels on the same ragged batch -- 600 utterances, so that workgroups meet utterance ends
    in both frames of a wave, rows without a frame, one-sample and empty utterances (zero
    padding only), digital silence (floor and clamp) and loud clicks (every other row of the
    utterance clamps) -- and against the oracle on the first few.  Cepstral counts on both
    sides of what the kernel keeps in LDS, filter counts that make a filter wider than a
    lane's run, with and without the clamp."""
    rng = np.random.default_rng(23)
    lo = 1025 if pad_mode == "reflect" else 0
    lens = [int(v) for v in rng.integers(lo, 9000, size=600)]
    lens[5:9] = [2048, 2047, 4096, 512 * 7 + 1024]
    if pad_mode == "constant":
        lens[20:23] = [0, 1, 3]
    sigs = []
    for k, n in enumerate(lens):
        s = (0.05 * rng.standard_normal(n)).astype(np.float32)
        if n > 4000 and k % 3 == 0:
            s[1000:3500] = 0.0
        if n > 4000 and k % 7 == 0:
            s[n // 2] = 30.0
        sigs.append(s)
    kw = dict(n_fft=2048, hop_length=512, n_mels=n_mels, pad_mode=pad_mode, top_db=top_db)
    ctx = sra.Context.default()
    f, ln = sra.mfcc.librosa_mfcc_batch(sigs, 22050, n_mfcc, dtype="float32", **kw)
    ctx.set_exact_f32(True)          # the shape-generic kernels
    try:
        g, ln_g = sra.mfcc.librosa_mfcc_batch(sigs, 22050, n_mfcc, dtype="float32", **kw)
    finally:
        ctx.set_exact_f32(False)
    assert list(ln) == list(ln_g) and f.shape == g.shape
    assert np.isfinite(f).all()
    off = np.concatenate([[0], np.cumsum(ln)])
    worst = 0.0
    for k in range(len(sigs)):
        if ln[k]:
            worst = max(worst, _row_rms_error(f[off[k]:off[k + 1]], g[off[k]:off[k + 1]]))
    cfg = mo.LibrosaConfig(sample_rate=22050, n_mfcc=n_mfcc, n_fft=2048, hop_length=512, n_mels=n_mels,
                           pad_mode=pad_mode, top_db=top_db)
    worst_o = 0.0
    for k in list(range(12)) + [20, 21, 22]:
        ref = mo.librosa_mfcc(sigs[k].astype(np.float64), cfg)
        assert ref.shape == (ln[k], n_mfcc)
        if ref.size:
            worst_o = max(worst_o, _row_rms_error(f[off[k]:off[k + 1]], ref))
    print(f"2048-point register kernel {pad_mode} {n_mels}/{n_mfcc}/{top_db}: vs generic {worst:.2e}, vs oracle {worst_o:.2e}")
    assert worst <= 2e-4 and worst_o <= 2e-4
