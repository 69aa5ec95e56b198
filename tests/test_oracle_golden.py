"""Pin the CPU oracle to the reference's own outputs (tests/golden/*.npz were
produced by importing the reference, see tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import hmm_oracle as ho
from oracle import mfcc_oracle as mo
from conftest import golden_params

CASES = ["s3d26", "s8d39", "s16d39", "s3d13", "s5d7lr"]


def _cfg(sr):
    return mo.MfccConfig(sample_rate=int(sr))


def test_mel_filterbank_matches_reference(mfcc_golden):
    np.testing.assert_array_equal(mo.mel_filterbank(1024), mfcc_golden["fbank1024"])
    np.testing.assert_array_equal(mo.mel_filterbank(512), mfcc_golden["fbank512"])
    assert np.count_nonzero(mfcc_golden["fbank1024"]) == 944


@pytest.mark.parametrize("i", range(9))
def test_mfcc_oracle_matches_reference(mfcc_golden, i):
    g = mfcc_golden
    sig, sr = g[f"sig{i}"].astype(np.float64), int(g[f"sr{i}"])
    cfg = _cfg(sr)
    cep = mo.cepstra(sig, cfg)
    assert cep.shape == g[f"cep{i}"].shape
    np.testing.assert_allclose(cep, g[f"cep{i}"], rtol=1e-9, atol=1e-9)
    feat = mo.features(sig, cfg)
    np.testing.assert_allclose(feat.T, g[f"feat{i}"], rtol=1e-9, atol=1e-9)
    z = mo.extract(sig, cfg)
    assert z.shape == g[f"z{i}"].shape
    np.testing.assert_allclose(z, g[f"z{i}"], rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("loops", [True, False])
def test_hmm_oracle_matches_reference(hmm_golden, case, loops):
    g, p = hmm_golden, case + "_"
    prm = golden_params(g, case)
    X, L = g[p + "X"], g[p + "lengths"]
    np.testing.assert_allclose(ho.log_density_diag(X, prm.means, prm.covars),
                               g[p + "loglik"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ho.score(prm, X, L, loops=loops), g[p + "utt_logprob"],
                               rtol=1e-12)
    st, lps = ho.estep(prm, X, L, loops=loops)
    assert st["nobs"] == int(g[p + "st_nobs"])
    for k, n in (("start", "start"), ("trans", "trans"), ("post", "post"),
                 ("obs", "obs"), ("obs**2", "obs2")):
        np.testing.assert_allclose(st[k], g[p + "st_" + n], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(lps.sum(), g[p + "logprob"], rtol=1e-12)
    q = ho.mstep_transitions(prm, st)
    np.testing.assert_allclose(q.pi, g[p + "em_pi"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(q.A, g[p + "em_A"], rtol=1e-12, atol=1e-15)
    q = ho.mstep_full(prm, st)
    np.testing.assert_allclose(q.pi, g[p + "m_pi"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(q.A, g[p + "m_A"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(q.means, g[p + "m_mu"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(q.covars, g[p + "m_var"], rtol=1e-9, atol=1e-12)
    st2, lps2 = ho.estep(q, X, L, loops=loops)
    np.testing.assert_allclose(lps2.sum(), g[p + "logprob2"], rtol=1e-11)
    q2 = ho.mstep_full(q, st2)
    np.testing.assert_allclose(q2.means, g[p + "m2_mu"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(q2.covars, g[p + "m2_var"], rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("case", CASES)
def test_lattices_match_reference(hmm_golden, case):
    g, p = hmm_golden, case + "_"
    prm = golden_params(g, case)
    x = g[p + "X"][: int(g[p + "lengths"][0])]
    B = np.exp(ho.log_density_diag(x, prm.means, prm.covars))
    for fwd, bwd, xis in ((ho.forward_scaling_loops, ho.backward_scaling_loops, ho.xi_sum_loops),
                          (ho.forward_scaling_vec, ho.backward_scaling_vec, ho.xi_sum_vec)):
        lp, al, c = fwd(prm.pi, prm.A, B)
        be = bwd(prm.pi, prm.A, B, c)
        np.testing.assert_allclose(al, g[p + "alpha0"], rtol=1e-11, atol=1e-300)
        np.testing.assert_allclose(be, g[p + "beta0"], rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(ho.posteriors(al, be), g[p + "gamma0"], rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(xis(al, prm.A, be, B), g[p + "xi0"], rtol=1e-10, atol=1e-300)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("loops", [True, False])
def test_posteriors_match_reference(hmm_golden, posteriors_golden, case, loops):
    g, p = hmm_golden, case + "_"
    prm = golden_params(g, case)
    lps, post = ho.score_posteriors(prm, g[p + "X"], g[p + "lengths"], loops=loops)
    np.testing.assert_allclose(lps.sum(), posteriors_golden[p + "logprob"], rtol=1e-12)
    np.testing.assert_allclose(post, posteriors_golden[p + "posteriors"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(post[: int(g[p + "lengths"][0])], g[p + "gamma0"], rtol=1e-9,
                               atol=1e-300)


@pytest.mark.parametrize("case", CASES)
def test_viterbi_matches_reference(hmm_golden, case):
    g, p = hmm_golden, case + "_"
    prm = golden_params(g, case)
    lp, path = ho.viterbi(prm, g[p + "X"], g[p + "lengths"])
    np.testing.assert_allclose(lp, g[p + "vit_logprob"], rtol=1e-13)
    np.testing.assert_array_equal(path, g[p + "vit_path"])


def test_endpoint_oracle_matches_reference(endpoint_golden):
    """Block energies / zero crossings, thresholds, segments and the cut signal of every
    fixture chunk (outputs of the reference's AudioMonitor)."""
    from oracle import endpoint_oracle as eo
    g = endpoint_golden
    n_speech = 0
    for i in range(int(g["n"])):
        x, fs = g[f"x{i}"].astype(np.float64), int(g[f"fs{i}"])
        energy, zcr = eo.block_features(x, eo.block_size(fs))
        np.testing.assert_allclose(energy, g[f"energy{i}"], rtol=1e-12)
        np.testing.assert_array_equal(zcr, g[f"zcr{i}"])
        segs, itu, itl, izct = eo.detect(x, fs)
        assert np.array_equal(np.array(segs, dtype=np.int64).reshape(-1, 4), g[f"segs{i}"])
        if len(segs):
            n_speech += 1
            np.testing.assert_allclose([itu, itl, izct], g[f"thr{i}"], rtol=1e-12)
        np.testing.assert_array_equal(eo.combine(x, segs, fs).astype(np.float32), g[f"cut{i}"])
    assert n_speech >= 10


def test_split_lengths_error():
    with pytest.raises(ValueError):
        ho.split_lengths(np.zeros((5, 2)), [2, 2])
    with pytest.raises(ValueError):
        ho.forward_scaling_loops(np.ones(2) / 2, np.ones((3, 3)) / 3, np.ones((4, 3)))


# ---- compiled C oracle (oracle/srx_oracle.c) pinned to the same vectors ------
from oracle import c_oracle as co

needs_c = pytest.mark.skipif(not co.available(), reason="run `make -C oracle` first")


@needs_c
@pytest.mark.parametrize("case", CASES)
def test_c_oracle_matches_reference(hmm_golden, case):
    g, p = hmm_golden, case + "_"
    prm = golden_params(g, case)
    X, L = g[p + "X"], g[p + "lengths"]
    np.testing.assert_allclose(co.score(prm, X, L), g[p + "utt_logprob"], rtol=1e-12)
    rec = co.estep(prm, X, L)
    S, D = prm.S, prm.D
    assert rec[0] == int(g[p + "st_nobs"])
    np.testing.assert_allclose(rec[1], g[p + "logprob"], rtol=1e-12)
    o = 2
    for n, size in (("start", S), ("trans", S * S), ("post", S), ("obs", S * D), ("obs2", S * D)):
        np.testing.assert_allclose(rec[o:o + size], g[p + "st_" + n].ravel(), rtol=1e-10, atol=1e-12)
        o += size
    lp, path = co.viterbi(prm, X, L)
    np.testing.assert_allclose(lp, g[p + "vit_logprob"], rtol=1e-13)
    np.testing.assert_array_equal(path, g[p + "vit_path"])


@needs_c
@pytest.mark.parametrize("i", range(9))
def test_c_oracle_cepstra_match_reference(mfcc_golden, i):
    g = mfcc_golden
    sig, sr = g[f"sig{i}"].astype(np.float64), int(g[f"sr{i}"])
    cfg = _cfg(sr)
    cep = co.cepstra(sig, cfg, g["fbank1024"])
    np.testing.assert_allclose(cep, g[f"cep{i}"], rtol=1e-9, atol=1e-8)


def test_large_model_oracle_pins(hmm_golden):
    """The large-model oracle pieces against pinned quantities: a mixture of ONE
    component with weight 1 is the reference's diagonal density (golden loglik),
    and the vectorised dense Viterbi equals the loop-faithful one (golden path)."""
    g = hmm_golden
    for case in ("s3d26", "s8d39", "s16d39"):
        p = case + "_"
        mu, var = g[p + "mu"], g[p + "var"]
        ll = ho.gmm_log_density(g[p + "X"], np.ones((mu.shape[0], 1)), mu[:, None, :], var[:, None, :])
        np.testing.assert_allclose(ll, g[p + "loglik"], rtol=1e-12, atol=1e-12)
        lp, path = ho.viterbi_dense(g[p + "pi"], g[p + "A"], g[p + "loglik"], g[p + "lengths"])
        np.testing.assert_array_equal(path, g[p + "vit_path"])
        np.testing.assert_allclose(lp, g[p + "vit_logprob"], rtol=1e-12)
    # two equal components with weights 1/2 each: the same density again
    p = "s8d39_"
    mu, var = g[p + "mu"], g[p + "var"]
    ll2 = ho.gmm_log_density(g[p + "X"], np.full((8, 2), 0.5), np.stack([mu, mu], 1), np.stack([var, var], 1))
    np.testing.assert_allclose(ll2, g[p + "loglik"], rtol=1e-12, atol=1e-10)


def test_librosa_restatement_invariants():
    """The librosa-convention oracle has no reference output to be pinned on (librosa is
    absent; parity unpinned).  What can be checked without it: the one filter value
    librosa's documentation prints for filters.mel(sr=22050, n_fft=2048) (row 0, bin 1:
    0.016), unit-area triangles, the frame count of a centred transform, the top_db
    clamp, and that libsrx builds the same Slaney filter matrix (host code, no GPU)."""
    from oracle import mfcc_oracle as mo
    fb = mo.slaney_filterbank(22050, 2048, 128)
    assert fb.shape == (128, 1025)
    assert round(float(fb[0, 1]), 3) == 0.016 and fb[0, 0] == 0.0
    freqs = np.linspace(0, 11025.0, 1025)
    area = np.array([np.trapezoid(r, freqs) for r in fb[20:]])
    np.testing.assert_allclose(area, 1.0, atol=0.05)        # sampled triangles, 10.8 Hz bins
    np.testing.assert_allclose(mo.slaney_mel_to_hz(mo.slaney_hz_to_mel([0.0, 440.0, 1000.0, 8000.0])),
                               [0.0, 440.0, 1000.0, 8000.0], rtol=1e-12)
    cfg = mo.LibrosaConfig()
    assert [cfg.n_frames(n) for n in (0, 1, 511, 512, 20727)] == [0, 1, 1, 2, 41]
    rng = np.random.default_rng(0)
    y = 0.1 * rng.standard_normal(12000)
    y[2000:8000] = 0.0
    S = mo.librosa_melspectrogram(y, cfg)
    db = 10 * np.log10(np.maximum(1e-10, S))
    assert db.min() < db.max() - 80.0                       # the clamp has something to do
    c = mo.librosa_mfcc(y, cfg)
    c_noclamp = mo.librosa_mfcc(y, mo.LibrosaConfig(top_db=0.0))
    assert c.shape == (24, 13) and np.abs(c - c_noclamp).max() > 1.0
    z = mo.zscore(mo.librosa_mfcc(y, cfg))
    np.testing.assert_allclose(z.mean(axis=0), 0.0, atol=1e-12)
    import speech_recognition_amd as sra
    np.testing.assert_allclose(sra.runtime.filterbank_host(2048, 22050, 128, "slaney"), fb, atol=1e-15)
    np.testing.assert_allclose(sra.runtime.filterbank_host(1024, 16000, 40, "slaney"),
                               mo.slaney_filterbank(16000, 1024, 40), atol=1e-15)


def test_oracles_match_reference_harness_run():
    """The 24 recordings the end-to-end GPU test uses: the front-end oracle reproduces the
    reference's per-file features, the HMM oracle its 24 x 6 scores and labels, and a
    transitions-only EM from the stored initial parameters its training histories."""
    import os
    from conftest import GOLDEN
    hg = np.load(os.path.join(GOLDEN, "harness_golden.npz"))
    fg = np.load(os.path.join(GOLDEN, "folds_golden.npz"))
    cfg = mo.MfccConfig(sample_rate=int(hg["sr"]))
    so = np.concatenate([[0], np.cumsum(hg["sig_lengths"])])
    fo = np.concatenate([[0], np.cumsum(hg["lengths"])])
    for i in range(0, 24, 3):
        sig = hg["signals"][so[i]:so[i + 1]].astype(np.float64)
        np.testing.assert_allclose(mo.features(sig, cfg), hg["feat"][fo[i]:fo[i + 1]], rtol=0,
                                   atol=1e-9 * np.abs(hg["feat"]).max())
        np.testing.assert_allclose(mo.extract(sig, cfg), hg["z"][fo[i]:fo[i + 1]], rtol=0, atol=1e-8)
    words = [str(w) for w in hg["words"]]
    prm = [ho.Params(fg[f"f1_{w}_pi"], fg[f"f1_{w}_A"], fg[f"f1_{w}_mu"], fg[f"f1_{w}_var"]) for w in words]
    sc = np.stack([ho.score(p, hg["z"], hg["lengths"], loops=False) for p in prm], axis=1)
    np.testing.assert_allclose(sc, hg["scores"], rtol=1e-10)
    assert [words[int(np.argmax(r))] for r in sc] == [str(l) for l in hg["labels"]]
    # training: from the trained transition rows back to the history is not possible, but the
    # last history entry is the log-probability under the parameters before the last update,
    # and the stored means / covariances never move: score the final model instead
    names = [str(n) for n in hg["names"]]
    for w in words:
        p = ho.Params(hg[f"train_{w}_pi"], hg[f"train_{w}_A"], hg[f"train_{w}_mu"], hg[f"train_{w}_var"])
        got = ho.score(p, hg["z"], hg["lengths"], loops=False)
        np.testing.assert_allclose(got, hg[f"train_{w}_score"], rtol=1e-10)
        idx = [i for i, n in enumerate(names) if n[: n.rfind("_")] == w]
        total = float(np.sum(got[idx]))
        hist = hg[f"train_{w}_hist"]
        assert total >= hist[-1] - 1e-6 * abs(hist[-1])     # EM never lowers the likelihood


def test_oracle_fit_replays_reference_training():
    """init (my_hmmlearn/hmm.py:116-134) and the fit loop (hmm.py:198-222, do_em_mstep only)
    restated with the oracle's E-step: the same seeds give the reference's training histories,
    iteration counts and parameters for the six word models of the harness fixture."""
    import os
    from conftest import GOLDEN
    hg = np.load(os.path.join(GOLDEN, "harness_golden.npz"))
    names = [str(n) for n in hg["names"]]
    fo = np.concatenate([[0], np.cumsum(hg["lengths"])])
    np.random.seed(int(hg["global_seed"]))
    for w in [str(x) for x in hg["train_order"]]:
        idx = [i for i, n in enumerate(names) if n[: n.rfind("_")] == w]
        X = np.vstack([hg["z"][fo[i]:fo[i + 1]] for i in idx])
        L = [int(hg["lengths"][i]) for i in idx]
        S = 3
        rs = np.random.mtrand._rand
        pi = rs.dirichlet(np.full(S, 1.0 / S))
        A = rs.dirichlet(np.full(S, 1.0 / S), size=S)
        np.random.seed(int(hg["seed"]))
        rows = np.random.choice(X.shape[0], S, replace=False)
        var = np.tile(np.diag(np.cov(X.T) + 1e-3 * np.eye(X.shape[1])), (S, 1))
        p = ho.Params(pi, A, X[rows], var)
        hist = []
        for _ in range(2000):
            st, lps = ho.estep(p, X, L, loops=False)
            p = ho.mstep_transitions(p, st)
            hist.append(float(np.sum(lps)))
            if len(hist) >= 2 and hist[-1] - hist[-2] < 1e-4:
                break
        assert len(hist) == int(hg[f"train_{w}_iter"]), w
        np.testing.assert_allclose(hist, hg[f"train_{w}_hist"], rtol=1e-11)
        np.testing.assert_allclose(p.A, hg[f"train_{w}_A"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(p.pi, hg[f"train_{w}_pi"], rtol=1e-9, atol=1e-13)
        np.testing.assert_array_equal(p.means, hg[f"train_{w}_mu"])
