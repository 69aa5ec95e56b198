"""Parity of the HIP path (through the C ABI) with the reference's outputs
(tests/golden) and with the CPU oracle on seeded inputs.

Tolerances: float64 is the validation arithmetic and must agree to rounding;
float32 must meet 1e-4 relative on log-likelihoods / MFCCs and 1e-5 relative
on re-estimated parameters (transition/start probabilities: absolute 1e-6,
they can be arbitrarily small)."""
import numpy as np
import pytest

from conftest import assert_ties_only, golden_params
from oracle import hmm_oracle as ho
from oracle import mfcc_oracle as mo

pytestmark = pytest.mark.gpu

CASES = ["s3d26", "s8d39", "s16d39", "s3d13", "s5d7lr"]


@pytest.fixture(scope="module")
def sra():
    import speech_recognition_amd as s
    return s


@pytest.fixture(scope="module")
def ctx(sra):
    return sra.Context.default()


def _bank(sra, ctx, prm_list):
    return sra.HMMBank(ctx, np.stack([p.pi for p in prm_list]), np.stack([p.A for p in prm_list]),
                       np.stack([p.means for p in prm_list]), np.stack([p.covars for p in prm_list]))


def _rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dtype,tol", [("float64", 1e-11), ("float32", 2e-5)])
def test_emission_and_score_match_reference(sra, ctx, hmm_golden, case, dtype, tol):
    g, p = hmm_golden, case + "_"
    prm = golden_params(g, case)
    bank = _bank(sra, ctx, [prm])
    data = sra.Ragged(ctx, g[p + "X"], g[p + "lengths"], dtype=dtype)
    ll = bank.emission_loglik(data).cpu().numpy()
    np.testing.assert_allclose(ll, g[p + "loglik"], rtol=tol, atol=tol * 10)
    for generic in (False, True):
        sc = bank.score(data, generic=generic).cpu().numpy()[:, 0]
        assert _rel(sc, g[p + "utt_logprob"]) < (1e-4 if dtype == "float32" else 1e-11)
        if dtype == "float32":   # far inside the 1e-4 bar in practice (split-bf16 emission)
            assert _rel(sc, g[p + "utt_logprob"]) < 2e-5
    if dtype == "float32":       # exact-f32 matrix path
        ctx.set_exact_f32(True)
        try:
            sc = bank.score(data).cpu().numpy()[:, 0]
        finally:
            ctx.set_exact_f32(False)
        assert _rel(sc, g[p + "utt_logprob"]) < 5e-6


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dtype,path", [("float64", "fused"), ("float64", "lane"), ("float32", "fused"),
                                        ("float32", "lane"), ("float32", "generic")])
def test_estep_and_mstep_match_reference(sra, ctx, hmm_golden, case, dtype, path):
    """float32: the fused matrix-core E-step, the lane-per-utterance E-step of small jobs (at
    most 8 states; larger models fall through to the fused kernel) and the shape-generic
    kernels; float64: the sub-wave chain kernels ("fused" = no lane path) and the
    lane-per-utterance E-step."""
    generic = path == "generic"
    ctx.set_estep_path("auto" if generic else path)
    try:
        _check_estep_and_mstep(sra, ctx, hmm_golden, case, dtype, generic)
    finally:
        ctx.set_estep_path("auto")


def _check_estep_and_mstep(sra, ctx, hmm_golden, case, dtype, generic):
    g, p = hmm_golden, case + "_"
    prm = golden_params(g, case)
    bank = _bank(sra, ctx, [prm])
    data = sra.Ragged(ctx, g[p + "X"], g[p + "lengths"], dtype=dtype)
    stats = bank.estep(data, generic=generic)
    st = bank.unpack_stats(stats)[0]
    rt = 1e-9 if dtype == "float64" else 2e-5
    assert st["nobs"] == int(g[p + "st_nobs"])
    assert abs(st["logprob"] - float(g[p + "logprob"])) < rt * abs(float(g[p + "logprob"]))
    for k, n in (("start", "start"), ("trans", "trans"), ("post", "post"), ("obs", "obs"),
                 ("obs**2", "obs2")):
        ref = g[p + "st_" + n]
        np.testing.assert_allclose(st[k], ref, rtol=rt, atol=rt * max(1.0, np.abs(ref).max()))
    # M-step on the device from the device statistics
    bank.mstep(stats, mode="full")
    sp, A, mu, cv = bank.params_host()
    pt = 1e-9 if dtype == "float64" else 1e-5
    np.testing.assert_allclose(sp[0], g[p + "m_pi"], rtol=pt, atol=1e-6 if dtype == "float32" else 1e-13)
    np.testing.assert_allclose(A[0], g[p + "m_A"], rtol=pt, atol=1e-6 if dtype == "float32" else 1e-13)
    np.testing.assert_allclose(mu[0], g[p + "m_mu"], rtol=pt, atol=pt)
    np.testing.assert_allclose(cv[0], g[p + "m_var"], rtol=pt, atol=pt)
    # second iteration from the re-estimated model
    own = ho.Params(sp[0], A[0], mu[0], cv[0])      # the device's own parameters after iteration 1
    stats2 = bank.estep(data, generic=generic)
    lp2 = float(stats2[0, 1].item())
    assert abs(lp2 - float(g[p + "logprob2"])) < (1e-9 if dtype == "float64" else 1e-5) * abs(lp2)
    bank.mstep(stats2, mode="full")
    sp, A, mu, cv = bank.params_host()
    # against the reference's trajectory (its iteration 2 started from ITS iteration 1) ...
    np.testing.assert_allclose(mu[0], g[p + "m2_mu"], rtol=pt * 5, atol=pt * 5)
    np.testing.assert_allclose(cv[0], g[p + "m2_var"], rtol=pt * 5, atol=pt * 5)
    # ... and the per-iteration bar proper: one oracle iteration from the parameters this
    # iteration started from
    st_o, lps_o = ho.estep(own, g[p + "X"], g[p + "lengths"], loops=False)
    q = ho.mstep_full(own, st_o)
    assert abs(lp2 - lps_o.sum()) < (1e-9 if dtype == "float64" else 1e-5) * abs(lp2)
    np.testing.assert_allclose(sp[0], q.pi, rtol=pt, atol=1e-6 if dtype == "float32" else 1e-13)
    np.testing.assert_allclose(A[0], q.A, rtol=pt, atol=1e-6 if dtype == "float32" else 1e-13)
    np.testing.assert_allclose(mu[0], q.means, rtol=pt, atol=pt)
    np.testing.assert_allclose(cv[0], q.covars, rtol=pt, atol=pt)


@pytest.mark.parametrize("case", CASES)
def test_transitions_only_mstep(sra, ctx, hmm_golden, case):
    g, p = hmm_golden, case + "_"
    prm = golden_params(g, case)
    bank = _bank(sra, ctx, [prm])
    data = sra.Ragged(ctx, g[p + "X"], g[p + "lengths"], dtype="float64")
    bank.mstep(bank.estep(data), mode="transitions_only")
    sp, A, mu, cv = bank.params_host()
    np.testing.assert_allclose(sp[0], g[p + "em_pi"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(A[0], g[p + "em_A"], rtol=1e-9, atol=1e-13)
    np.testing.assert_array_equal(mu[0], g[p + "mu"])       # untouched, as in fit()
    np.testing.assert_array_equal(cv[0], g[p + "var"])
    # structural zeros survive
    assert np.all(A[0][g[p + "A"] == 0] == 0)


@pytest.mark.parametrize("case", CASES)
def test_viterbi_paths_bit_exact(sra, ctx, hmm_golden, case):
    g, p = hmm_golden, case + "_"
    prm = golden_params(g, case)
    bank = _bank(sra, ctx, [prm])
    data = sra.Ragged(ctx, g[p + "X"], g[p + "lengths"], dtype="float64")
    lp, path = bank.viterbi(data)
    np.testing.assert_array_equal(path.cpu().numpy(), g[p + "vit_path"])
    np.testing.assert_allclose(lp.cpu().numpy(), g[p + "vit_logprob"], rtol=1e-12)
    # float32: ties excepted -- a differing frame must be a near-tie
    data32 = sra.Ragged(ctx, g[p + "X"], g[p + "lengths"], dtype="float32")
    lp32, path32 = bank.viterbi(data32)
    n_diff, gap, short = assert_ties_only(prm.pi, prm.A, g[p + "loglik"], g[p + "lengths"],
                                          path32.cpu().numpy(), g[p + "vit_path"], g[p + "vit_logprob"])
    print(f"viterbi float32 {case}: {n_diff} of {len(g[p + 'vit_path'])} frames differ, "
          f"largest tie gap {gap:.2e}, largest shortfall {short:.2e}")
    assert _rel(lp32.cpu().numpy(), g[p + "vit_logprob"]) < 1e-5


def test_multi_model_bank_and_ragged_edges(sra, ctx):
    rng = np.random.default_rng(11)
    S, D, M = 8, 39, 6
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    lengths = [1, 2, 200, 17, 64, 199, 3, 128]
    X = rng.standard_normal((sum(lengths), D))
    ref = np.stack([ho.score(p, X, lengths, loops=False) for p in prms], axis=1)
    bank = _bank(sra, ctx, prms)
    for dtype, tol in (("float64", 1e-11), ("float32", 2e-5)):
        data = sra.Ragged(ctx, X, lengths, dtype=dtype)
        for generic in (False, True):
            got = bank.score(data, generic=generic).cpu().numpy()
            assert got.shape == (len(lengths), M)
            assert _rel(got, ref) < tol
    ctx.set_exact_f32(True)
    try:
        got = bank.score(sra.Ragged(ctx, X, lengths, dtype="float32")).cpu().numpy()
    finally:
        ctx.set_exact_f32(False)
    assert _rel(got, ref) < 5e-6
    # batched E-step with a different model per utterance
    um = np.array([u % M for u in range(len(lengths))], dtype=np.int32)
    tr = sra.parallel.BaumWelchTrainer(bank, sra.Ragged(ctx, X, lengths, dtype="float64"), um)
    tr.bank.estep(tr.data, tr.utt_model, tr.utt_order, out=tr.stats)
    got = bank.unpack_stats(tr.stats)
    off = np.concatenate([[0], np.cumsum(lengths)])
    for m in range(M):
        sel = [u for u in range(len(lengths)) if um[u] == m]
        Xm = np.vstack([X[off[u]:off[u + 1]] for u in sel])
        st, lps = ho.estep(prms[m], Xm, [lengths[u] for u in sel], loops=False)
        assert got[m]["nobs"] == len(sel)
        np.testing.assert_allclose(got[m]["logprob"], lps.sum(), rtol=1e-10)
        for k in ("start", "trans", "post", "obs", "obs**2"):
            np.testing.assert_allclose(got[m][k], st[k], rtol=1e-8, atol=1e-10)
    # the same batch through both float32 E-steps (fused matrix-core kernel, one lane per
    # utterance), in label order and unsorted
    d32 = sra.Ragged(ctx, X, lengths, dtype="float32")
    umd = ctx.tensor(um)
    for path, order in (("fused", tr.utt_order), ("fused", None), ("lane", tr.utt_order), ("lane", None)):
        ctx.set_estep_path(path)
        try:
            got32 = bank.unpack_stats(bank.estep(d32, umd, order))
        finally:
            ctx.set_estep_path("auto")
        for m in range(M):
            sel = [u for u in range(len(lengths)) if um[u] == m]
            Xm = np.vstack([X[off[u]:off[u + 1]] for u in sel])
            st, lps = ho.estep(prms[m], Xm, [lengths[u] for u in sel], loops=False)
            assert got32[m]["nobs"] == len(sel)
            np.testing.assert_allclose(got32[m]["logprob"], lps.sum(), rtol=1e-5)
            for k in ("start", "trans", "post", "obs", "obs**2"):
                np.testing.assert_allclose(got32[m][k], st[k], rtol=2e-5,
                                           atol=2e-5 * max(1.0, np.abs(st[k]).max()))


def test_estep_order_independent(sra, ctx):
    rng = np.random.default_rng(5)
    prm = ho.random_params(rng, 4, 9)
    lengths = rng.integers(1, 60, size=40)
    X = rng.standard_normal((int(lengths.sum()), 9))
    bank = _bank(sra, ctx, [prm])
    data = sra.Ragged(ctx, X, lengths, dtype="float64")
    a = bank.estep(data).cpu().numpy()
    order = ctx.tensor(rng.permutation(len(lengths)).astype(np.int64))
    b = bank.estep(data, None, order).cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("i", range(9))
def test_mfcc_matches_reference(sra, ctx, mfcc_golden, i):
    g = mfcc_golden
    sig, sr = g[f"sig{i}"], int(g[f"sr{i}"])
    # Error measure for un-normalised cepstra: |got - ref| over the RMS of that coefficient
    # (a column of cep, a row of feat) across the utterance, so that c0 ~ -680 does not set the
    # scale for coefficients of size 1..30.  Columns that are numerically zero in the
    # reference (RMS below 1e-6 of the largest value: the higher cepstra and the deltas of an
    # all-zero signal, pure rounding noise of the DCT of a constant vector) have no scale of
    # their own and are measured against the largest value instead.
    def row_err(got, ref):
        if not ref.size:
            return 0.0
        rms = np.sqrt(np.mean(ref ** 2, axis=0, keepdims=True))
        top = np.abs(ref).max()
        return float(np.max(np.abs(got - ref) / np.where(rms > 1e-6 * top, rms, max(top, 1e-300))))

    for dtype, tol in (("float64", 1e-9), ("float32", 1e-4)):
        cep = sra.mfcc.compute_mfcc(sig, sr, dtype=dtype)
        assert cep.shape == g[f"cep{i}"].shape
        e_cep = row_err(cep, g[f"cep{i}"])
        f, lens = sra.mfcc.extract_batch([sig], sr, zscore=False, dtype=dtype)
        assert f.shape == g[f"feat{i}"].T.shape
        e_feat = row_err(f, g[f"feat{i}"].T)
        z = sra.mfcc.extract_features(sig, sr, dtype=dtype)
        assert z.shape == g[f"z{i}"].shape
        # z-scored features are O(1): absolute error (measured on MI355X: <= 1.4e-5 in float32)
        e_z = float(np.max(np.abs(z - g[f"z{i}"]))) if z.size else 0.0
        print(f"mfcc sig{i} {dtype}: cepstra {e_cep:.2e}, features {e_feat:.2e} of row RMS; "
              f"z-scored {e_z:.2e} absolute")
        assert e_cep < tol and e_feat < tol
        assert e_z < (1e-7 if dtype == "float64" else 1e-4)


def test_mfcc_batch_ragged_equals_single(sra, ctx, mfcc_golden):
    g = mfcc_golden
    sigs = [g[f"sig{i}"] for i in range(9) if int(g[f"sr{i}"]) == 16000]
    sigs.insert(2, np.zeros(100, dtype=np.float32))     # shorter than one frame -> 0 frames
    f, lens = sra.mfcc.extract_batch(sigs, 16000, dtype="float64")
    assert lens[2] == 0
    off = np.concatenate([[0], np.cumsum(lens)])
    for k, s in enumerate(sigs):
        one = sra.mfcc.extract_features(s, 16000, dtype="float64")
        np.testing.assert_array_equal(f[off[k]:off[k + 1]], one)
    # benchmark-shaped configuration against the oracle: 512-point FFT, 40 mel, delta + delta-delta
    cfg = mo.MfccConfig(sample_rate=16000, nfft=512, n_mel=40, delta_order=2)
    f2, lens2 = sra.mfcc.extract_batch(sigs, 16000, nfft=512, n_mel=40, delta_order=2, dtype="float64")
    off2 = np.concatenate([[0], np.cumsum(lens2)])
    for k, s in enumerate(sigs):
        ref = mo.extract(s.astype(np.float64), cfg)
        assert ref.shape == (lens2[k], 39)
        if not np.any(s):
            continue   # all-zero input: columns are rounding noise, their z-score is arbitrary
        np.testing.assert_allclose(f2[off2[k]:off2[k + 1]], ref, rtol=0, atol=1e-7)
    # float32: the register-FFT kernel (default) and the generic float32 kernel, un-normalised
    cfg_raw = mo.MfccConfig(sample_rate=16000, nfft=512, n_mel=40, delta_order=2, zscore=False)
    for exact in (False, True):
        ctx.set_exact_f32(exact)
        try:
            f3, lens3 = sra.mfcc.extract_batch(sigs, 16000, nfft=512, n_mel=40, delta_order=2,
                                               zscore=False, dtype="float32")
        finally:
            ctx.set_exact_f32(False)
        off3 = np.concatenate([[0], np.cumsum(lens3)])
        for k, s in enumerate(sigs):
            ref = mo.extract(s.astype(np.float64), cfg_raw)
            if ref.size == 0 or not np.any(s):
                continue
            np.testing.assert_allclose(f3[off3[k]:off3[k + 1]], ref, rtol=1e-4,
                                       atol=1e-4 * np.abs(ref).max())


@pytest.mark.parametrize("seconds,delta_order,zscore", [(3.5, 2, True), (9.0, 2, True), (9.0, 1, True),
                                                        (9.0, 0, True), (5.0, 2, False), (0.5, 1, False)])
def test_mfcc_long_utterances(sra, ctx, seconds, delta_order, zscore):
    """Utterances longer than what one thread keeps in registers (more than 16 rows per
    row segment) and longer than one staging tile of the delta / z-score stage, mixed
    with short ones in one batch, against the oracle (float64 1e-7 absolute on z-scored,
    float32 1e-4 of the largest value un-normalised, 3e-3 absolute z-scored)."""
    rng = np.random.default_rng(int(seconds * 10) + delta_order)
    n = int(seconds * 16000)
    t = np.arange(n) / 16000.0
    long_sig = 0.1 * rng.standard_normal(n) + 0.2 * np.sin(2 * np.pi * (300 + 40 * t) * t)
    sigs = [0.1 * rng.standard_normal(4000), long_sig, 0.1 * rng.standard_normal(16000)]
    cfg = mo.MfccConfig(sample_rate=16000, nfft=512, n_mel=40, delta_order=delta_order, zscore=zscore)
    refs = [mo.extract(s.astype(np.float64), cfg) for s in sigs]
    for dtype in ("float64", "float32"):
        f, lens = sra.mfcc.extract_batch([s.astype(np.float32) if dtype == "float32" else s for s in sigs],
                                         16000, nfft=512, n_mel=40, delta_order=delta_order,
                                         zscore=zscore, dtype=dtype)
        off = np.concatenate([[0], np.cumsum(lens)])
        for k, ref in enumerate(refs):
            got = f[off[k]:off[k + 1]]
            assert got.shape == ref.shape
            if dtype == "float64":
                np.testing.assert_allclose(got, ref, rtol=0, atol=1e-7 * max(1.0, np.abs(ref).max()))
            elif zscore:
                np.testing.assert_allclose(got, ref, rtol=0, atol=3e-3)
            else:
                np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


def test_class_surface_and_errors(sra, ctx, hmm_golden):
    g, p = hmm_golden, "s3d26_"
    m = sra.MyGaussianHMM(n_components=3, tol=1e-4, n_iter=2, implementation="scaling",
                          verbose=False, mstep="full")
    m.startprob_, m.transmat_ = g[p + "pi"].copy(), g[p + "A"].copy()
    m.means_, m._covars_ = g[p + "mu"].copy(), g[p + "var"].copy()
    X, L = g[p + "X"], g[p + "lengths"]
    assert abs(m.score(X, L) - float(g[p + "score_all"])) < 1e-9 * abs(float(g[p + "score_all"]))
    st, lp = m.do_estep(X, L)
    np.testing.assert_allclose(st["trans"], g[p + "st_trans"], rtol=1e-9)
    lpv, path = m.decode(X, L)
    np.testing.assert_array_equal(path, g[p + "vit_path"])
    m.fit_from_current(X, L)
    assert list(m.monitor_.history)[0] == pytest.approx(float(g[p + "logprob"]), rel=1e-11)
    assert list(m.monitor_.history)[1] == pytest.approx(float(g[p + "logprob2"]), rel=1e-10)
    np.testing.assert_allclose(m.means_, g[p + "m2_mu"], rtol=1e-8, atol=1e-10)
    with pytest.raises(ValueError):
        m.score(X, [5, 5])                       # lengths do not sum to n_samples
    with pytest.raises(ValueError):
        m.score(X[:, :5])                        # wrong feature count
    with pytest.raises(ValueError):
        m.covars_ = -np.ones((3, 26))
    import pickle
    m2 = pickle.loads(pickle.dumps(m))
    assert m2.score(X, L) == m.score(X, L)
    # the reference's own init path runs and converges from its random start
    np.random.seed(0)
    m3 = sra.MyGaussianHMM(n_components=3, n_iter=5, tol=1e-4, verbose=False, random_state=3)
    m3.fit(X, L)
    h = list(m3.monitor_.history)
    assert len(h) >= 2 and h[-1] >= h[0]


def test_fold_accuracies_identical(sra, ctx, folds_golden):
    """The 5-fold command-word experiment: train 6 word models per fold from
    the fixtures' deterministic start for the same number of iterations, score
    the fold's validation utterances -- labels and right-counts must equal the
    reference's, parameters agree per model."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import fold_init
    g = folds_golden
    names = list(g["names"])
    words = list(g["words"])
    lengths = g["lengths"]
    off = np.concatenate([[0], np.cumsum(lengths)])
    feats = g["feats"].astype(np.float64)
    index = {n: i for i, n in enumerate(names)}
    n_iter = int(g["n_iter"])

    def utt(n):
        i = index[n]
        return feats[off[i]:off[i + 1]]

    for fold in range(1, 6):
        train, val = list(g[f"f{fold}_train"]), list(g[f"f{fold}_val"])
        inits, xs, um = [], [], []
        for w_i, w in enumerate(words):
            sel = [n for n in train if n[: n.rfind("_")] == w]
            Xw = np.vstack([utt(n) for n in sel])
            inits.append(ho.Params(*fold_init(Xw, 3)))
            xs += [utt(n) for n in sel]
            um += [w_i] * len(sel)
        bank = _bank(sra, ctx, inits)
        data = sra.Ragged(ctx, np.vstack(xs), [len(x) for x in xs], dtype="float64")
        tr = sra.parallel.BaumWelchTrainer(bank, data, np.array(um, dtype=np.int32))
        hist = np.stack([tr.step().cpu().numpy() for _ in range(n_iter)], axis=1)
        sp, A, mu, cv = bank.params_host()
        for w_i, w in enumerate(words):
            q = f"f{fold}_{w}_"
            np.testing.assert_allclose(hist[w_i], g[q + "hist"], rtol=1e-9)
            np.testing.assert_allclose(mu[w_i], g[q + "mu"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(cv[w_i], g[q + "var"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(A[w_i], g[q + "A"], rtol=1e-5, atol=1e-9)
        vdata = sra.Ragged(ctx, np.vstack([utt(n) for n in val]), [len(utt(n)) for n in val],
                           dtype="float64")
        scores = bank.score(vdata).cpu().numpy()
        np.testing.assert_allclose(scores, g[f"f{fold}_scores"], rtol=1e-7)
        pred = [words[int(np.argmax(r))] for r in scores]
        assert pred == list(g[f"f{fold}_pred"])
        assert sum(p == n[: n.rfind("_")] for p, n in zip(pred, val)) == int(g[f"f{fold}_right"])
        # float32 fast path gives the same labels
        s32 = bank.score(sra.Ragged(ctx, vdata.X, offsets=vdata.offsets, dtype="float32")).cpu().numpy()
        assert [words[int(np.argmax(r))] for r in s32] == pred


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_cross_validate_batched_matches_reference(sra, ctx, folds_golden, dtype):
    """All 30 (fold, word) models trained in one batched job: histories,
    parameters, scores, labels and right-counts as the reference's per-fold runs."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import fold_init
    g = folds_golden
    words = list(g["words"])
    folds = [(list(g[f"f{k}_train"]), list(g[f"f{k}_val"])) for k in range(1, 6)]
    res = sra.recognizer.cross_validate(g["feats"].astype(np.float64), g["lengths"], list(g["names"]),
                                        folds, words, n_components=3, n_iter=int(g["n_iter"]),
                                        init_fn=fold_init, dtype=dtype, ctx=ctx)
    sp, A, mu, cv = res["bank"].params_host()
    tight = dtype == "float64"
    worst = {"mu": 0.0, "A": 0.0, "hist": 0.0, "scores": 0.0}
    for k in range(1, 6):
        for w_i, w in enumerate(words):
            q, m = f"f{k}_{w}_", (k - 1) * len(words) + w_i
            worst["mu"] = max(worst["mu"], float(np.max(np.abs(mu[m] - g[q + "mu"]) / (np.abs(g[q + "mu"]) + 0.1))))
            worst["A"] = max(worst["A"], float(np.max(np.abs(A[m] - g[q + "A"]) / (np.abs(g[q + "A"]) + 1e-2))))
            worst["hist"] = max(worst["hist"], float(np.max(np.abs(res["hist"][k - 1, w_i] / g[q + "hist"] - 1))))
        worst["scores"] = max(worst["scores"], float(np.max(np.abs(res["folds"][k - 1]["scores"] / g[f"f{k}_scores"] - 1))))
    print(f"cross_validate {dtype}: largest deviations after {int(g['n_iter'])} EM iterations", worst)
    for k in range(1, 6):
        fr = res["folds"][k - 1]
        assert fr["pred"] == list(g[f"f{k}_pred"])
        assert fr["right"] == int(g[f"f{k}_right"])
        # (measured on MI355X after 8 EM iterations: float64 1e-15 scores, 6e-14 means; float32
        # 2.2e-6 scores, 1.4e-7 histories, 1.1e-5 of |mean| + 0.1, 3.2e-6 of |a_ij| + 0.01)
        np.testing.assert_allclose(fr["scores"], g[f"f{k}_scores"], rtol=1e-12 if tight else 2e-5)
        for w_i, w in enumerate(words):
            q, m = f"f{k}_{w}_", (k - 1) * len(words) + w_i
            np.testing.assert_allclose(res["hist"][k - 1, w_i], g[q + "hist"], rtol=1e-12 if tight else 2e-6)
            np.testing.assert_allclose(mu[m], g[q + "mu"], rtol=1e-10 if tight else 5e-5, atol=1e-11 if tight else 5e-6)
            np.testing.assert_allclose(A[m], g[q + "A"], rtol=1e-10 if tight else 2e-5, atol=1e-12 if tight else 2e-7)


@pytest.mark.parametrize("M,S,D", [(30, 3, 26), (12, 8, 39), (7, 8, 13), (13, 5, 39), (50, 4, 20),
                                   (1, 16, 39), (6, 16, 39), (3, 9, 13), (7, 12, 26), (2, 16, 48)])
def test_scoring_banks_larger_than_three_tiles(sra, ctx, M, S, D):
    """More state rows than one pass of the matrix-core kernel holds (3 tiles of
    16): the bank is scored in slices of the grid, same results as the
    shape-generic kernels and the oracle.  Models of 9 to 16 states take a whole tile
    each."""
    rng = np.random.default_rng(M * 100 + S)
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    lengths = [1, 2, 50, 17, 33, 64, 3, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 5]
    X = rng.standard_normal((sum(lengths), D))
    ref = np.stack([ho.score(p, X, lengths, loops=False) for p in prms], axis=1)
    bank = _bank(sra, ctx, prms)
    data = sra.Ragged(ctx, X, lengths, dtype="float32")
    got = bank.score(data).cpu().numpy()
    gen = bank.score(data, generic=True).cpu().numpy()
    assert got.shape == (len(lengths), M)
    assert _rel(got, ref) < 2e-5
    assert _rel(got, gen) < 2e-5


@pytest.mark.parametrize("dtype,M,S,tol", [("float32", 12, 8, 2e-5), ("float32", 40, 3, 2e-5), ("float32", 5, 16, 2e-5),
                                           ("float64", 9, 8, 1e-12), ("float64", 21, 4, 1e-12), ("float64", 3, 16, 1e-12)])
def test_wide_banks_over_many_batches(sra, ctx, dtype, M, S, tol):
    """Banks of several slices over many batches of 16 utterances (the slices of a batch are
    taken by neighbouring waves; a last batch that is partly filled; lengths 0 and 1 among
    them): every (utterance, model) score against the shape-generic kernels, and a sample of
    them against the oracle."""
    D = 39
    rng = np.random.default_rng(3 * M + S)
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    U = 1003
    lengths = rng.integers(1, 60, size=U)
    lengths[[0, 500, 1002]] = 0
    lengths[[1, 999]] = 1
    X = rng.standard_normal((int(lengths.sum()), D))
    bank = _bank(sra, ctx, prms)
    data = sra.Ragged(ctx, X, lengths, dtype=dtype)
    got = bank.score(data).cpu().numpy()
    gen = bank.score(data, generic=True).cpu().numpy()
    live = lengths > 0
    assert got.shape == (U, M)
    np.testing.assert_allclose(got[live], gen[live], rtol=tol, atol=tol)
    np.testing.assert_array_equal(got[~live], gen[~live])
    off = np.concatenate([[0], np.cumsum(lengths)])
    for u in (7, 640, 1001):
        for m in (0, M // 2, M - 1):
            ref = ho.score(prms[m], X[off[u]:off[u + 1]], loops=False)
            assert abs(got[u, m] - ref) <= max(tol, 2e-5 if dtype == "float32" else 1e-11) * abs(ref)


# ------------------------------------------------------------------ posteriors
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dtype,tol", [("float64", 1e-10), ("float32", 2e-5)])
def test_posteriors_match_reference(sra, ctx, hmm_golden, posteriors_golden, case, dtype, tol):
    """srx_posteriors against the reference's score_scaling(compute_posteriors=True)
    output for every utterance of the case (absolute tolerance: posteriors lie in
    [0, 1]; float32 2e-5, float64 1e-10)."""
    g, p = hmm_golden, case + "_"
    bank = _bank(sra, ctx, [golden_params(g, case)])
    data = sra.Ragged(ctx, g[p + "X"], g[p + "lengths"], dtype=dtype)
    lp, gam = bank.posteriors(data)
    gam = gam.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(gam, posteriors_golden[p + "posteriors"], rtol=0, atol=tol)
    np.testing.assert_allclose(gam.sum(axis=1), 1.0, rtol=0, atol=10 * tol)
    assert _rel(lp.cpu().numpy(), g[p + "utt_logprob"]) < (2e-5 if dtype == "float32" else 1e-11)


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-10), ("float32", 2e-5)])
@pytest.mark.parametrize("S,D", [(3, 13), (8, 39), (16, 39), (24, 20), (5, 50)])
def test_posteriors_ragged_multi_model(sra, ctx, S, D, dtype, tol):
    """Every utterance under its own model, ragged lengths with 1-frame and empty
    utterances, against the oracle."""
    rng = np.random.default_rng(S * 100 + D)
    M = 4
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    lengths = np.array([17, 1, 0, 64, 5, 33, 2, 90, 41, 8, 0, 29])
    um = rng.integers(0, M, size=len(lengths)).astype(np.int32)
    X = rng.standard_normal((lengths.sum(), D))
    bank = _bank(sra, ctx, prms)
    lp, gam = bank.posteriors(sra.Ragged(ctx, X, lengths, dtype=dtype), um)
    lp, gam = lp.cpu().numpy(), gam.cpu().numpy().astype(np.float64)
    off = np.concatenate([[0], np.cumsum(lengths)])
    for u, n in enumerate(lengths):
        if n == 0:
            assert lp[u] == 0.0
            continue
        rl, rg = ho.score_posteriors(prms[um[u]], X[off[u]:off[u + 1]], [n], loops=False)
        assert abs(lp[u] - rl[0]) <= (2e-5 if dtype == "float32" else 1e-11) * abs(rl[0])
        np.testing.assert_allclose(gam[off[u]:off[u + 1]], rg, rtol=0, atol=tol)


@pytest.mark.parametrize("S", [1, 2, 3, 4, 5, 6, 7, 8])
def test_posteriors_small_models_every_state_count(sra, ctx, S):
    """float32 posteriors of models of 1 to 8 states (the lane-per-utterance kernel that moves
    its data in tiles of 8 frames): several waves, the last one partly filled, lengths on
    both sides of every multiple of 8, empty utterances, against the oracle."""
    rng = np.random.default_rng(40 + S)
    M, D = 3, 6
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    lengths = np.concatenate([[0, 1, 2, 7, 8, 9, 15, 16, 17, 23, 24, 25, 64, 65, 0],
                              rng.integers(0, 70, size=135)])
    um = rng.integers(0, M, size=len(lengths)).astype(np.int32)
    X = rng.standard_normal((lengths.sum(), D))
    bank = _bank(sra, ctx, prms)
    lp, gam = bank.posteriors(sra.Ragged(ctx, X, lengths, dtype="float32"), um)
    lp, gam = lp.cpu().numpy(), gam.cpu().numpy().astype(np.float64)
    off = np.concatenate([[0], np.cumsum(lengths)])
    for u, n in enumerate(lengths):
        if n == 0:
            assert lp[u] == 0.0
            continue
        rl, rg = ho.score_posteriors(prms[um[u]], X[off[u]:off[u + 1]], [n], loops=False)
        assert abs(lp[u] - rl[0]) <= 2e-5 * abs(rl[0])
        np.testing.assert_allclose(gam[off[u]:off[u + 1]], rg, rtol=0, atol=2e-5)


def test_posteriors_into_an_unaligned_array(sra, ctx):
    """The tiled kernel moves 16 bytes at a time; an output array that is only 4-byte aligned
    takes the kernel that reads and writes row by row.  Same posteriors either way."""
    import torch
    rng = np.random.default_rng(77)
    S, D = 8, 9
    prms = [ho.random_params(rng, S, D) for _ in range(2)]
    lengths = rng.integers(0, 60, size=90)
    um = rng.integers(0, 2, size=len(lengths)).astype(np.int32)
    X = rng.standard_normal((lengths.sum(), D))
    bank = _bank(sra, ctx, prms)
    data = sra.Ragged(ctx, X, lengths, dtype="float32")
    lp0, g0 = bank.posteriors(data, um)
    flat = torch.zeros(data.n_frames * S + 1, dtype=torch.float32, device=g0.device)
    out = flat[1:].view(data.n_frames, S)
    assert out.data_ptr() % 16 == 4
    lp1, g1 = bank.posteriors(data, um, out=out)
    assert g1.data_ptr() == out.data_ptr() and float(flat[0]) == 0.0
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(lp1.cpu().numpy(), lp0.cpu().numpy(), rtol=1e-6)


def test_model_class_posterior_surface(sra, hmm_golden, posteriors_golden):
    """score_scaling / score_samples / predict_proba / predict / sample of the
    model class."""
    g, p = hmm_golden, "s3d26_"
    m = sra.MyGaussianHMM(n_components=3, verbose=False)
    prm = golden_params(g, "s3d26")
    m.startprob_, m.transmat_, m.means_, m._covars_ = prm.pi, prm.A, prm.means, prm.covars
    X, L = g[p + "X"], g[p + "lengths"]
    lp, post = m.score_scaling(X, L, compute_posteriors=True)
    np.testing.assert_allclose(lp, posteriors_golden[p + "logprob"], rtol=1e-11)
    np.testing.assert_allclose(post, posteriors_golden[p + "posteriors"], rtol=0, atol=1e-10)
    lp0, none = m.score_scaling(X, L, compute_posteriors=False)
    assert none.shape == (0, 3) and abs(lp0 - lp) < 1e-9 * abs(lp)
    np.testing.assert_array_equal(m.predict_proba(X, L), post)
    assert m.score_samples(X, L)[0] == lp
    np.testing.assert_array_equal(m.predict(X, L), g[p + "vit_path"])
    Xs, states = m.sample(25, random_state=np.random.RandomState(3))
    assert Xs.shape == (25, 26) and states.shape == (25,) and states.max() < 3
    Xs2, states2 = m.sample(25, random_state=np.random.RandomState(3))
    np.testing.assert_array_equal(Xs, Xs2)
    np.testing.assert_array_equal(states, states2)


# ------------------------------------------------------------------ end-pointing
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_endpoint_batch_matches_reference(sra, ctx, endpoint_golden, dtype):
    """All fixture chunks of one sample rate as one batch: block energies (1e-12
    relative; float32 input is accumulated in float64 like the float64 reference run),
    zero-crossing counts, segments and thresholds of the reference's AudioMonitor."""
    g = endpoint_golden
    for fs in (22050, 16000):
        idx = [i for i in range(int(g["n"])) if int(g[f"fs{i}"]) == fs]
        sigs = [g[f"x{i}"] for i in idx]
        so = np.concatenate([[0], np.cumsum([len(s) for s in sigs])])
        res = sra.endpoint.detect_batch(np.concatenate(sigs), so, fs, dtype=dtype, ctx=ctx)
        energy, zcr = res.energy.cpu().numpy(), res.zcr.cpu().numpy()
        segs, thr = res.segments_host(), res.thresholds.cpu().numpy()
        bo = res.block_offsets
        for k, i in enumerate(idx):
            np.testing.assert_allclose(energy[bo[k]:bo[k + 1]], g[f"energy{i}"], rtol=1e-12)
            np.testing.assert_array_equal(zcr[bo[k]:bo[k + 1]], g[f"zcr{i}"].astype(np.int32))
            assert np.array_equal(np.array(segs[k], dtype=np.int64).reshape(-1, 4), g[f"segs{i}"]), (i, segs[k])
            if len(segs[k]):
                np.testing.assert_allclose(thr[k], g[f"thr{i}"], rtol=1e-12)
        # the speech pieces, cut on the device
        packed, new_so, _ = sra.endpoint.cut_batch(np.concatenate(sigs), so, fs, dtype=dtype, ctx=ctx)
        packed = packed.cpu().numpy()
        for k, i in enumerate(idx):
            np.testing.assert_array_equal(packed[new_so[k]:new_so[k + 1]].astype(np.float32), g[f"cut{i}"])


def test_endpoint_class_surface_and_random_chunks(sra, ctx, endpoint_golden):
    """AudioMonitor.start_saved_audio / combine_audio in the reference's formats, and
    a seeded batch of random burst patterns (ragged lengths, an empty and a 3-sample
    recording) against the oracle."""
    from oracle import endpoint_oracle as eo
    g = endpoint_golden
    for i in (4, 5, 6, 9, 12):
        x, fs = g[f"x{i}"].astype(np.float64), int(g[f"fs{i}"])
        mon = sra.endpoint.AudioMonitor(device=0, fs=fs, ctx=ctx)
        th = mon.start_saved_audio(x)
        if len(g[f"segs{i}"]):
            assert th[0] == g[f"segs{i}"].tolist()
            np.testing.assert_allclose(th[1:], g[f"thr{i}"], rtol=1e-12)
            np.testing.assert_array_equal(np.asarray(mon.combine_audio(x, th[0], fs), dtype=np.float32),
                                          g[f"cut{i}"])
        else:
            assert th == [] or th[0] == [[0, 0, 0, 0]]
        np.testing.assert_allclose(mon.energies, g[f"energy{i}"], rtol=1e-12)
    rng = np.random.default_rng(99)
    fs = 16000
    sigs = []
    for r in range(40):
        n = int(rng.integers(2000, 40000))
        x = rng.uniform(0.0002, 0.002) * rng.standard_normal(n)
        for _ in range(int(rng.integers(0, 4))):
            a = int(rng.integers(0, n))
            b = min(n, a + int(rng.integers(200, 9000)))
            x[a:b] += rng.uniform(0.05, 0.4) * np.sin(2 * np.pi * rng.uniform(100, 3000) * np.arange(b - a) / fs)
        sigs.append(x)
    sigs.insert(7, np.zeros(0))
    sigs.insert(20, np.array([0.1, -0.1, 0.1]))
    so = np.concatenate([[0], np.cumsum([len(s) for s in sigs])])
    res = sra.endpoint.detect_batch(np.concatenate(sigs), so, fs, dtype="float64", ctx=ctx, max_segments=32)
    segs, thr = res.segments_host(), res.thresholds.cpu().numpy()
    found = 0
    for k, x in enumerate(sigs):
        if len(x) == 0:
            assert segs[k] == []
            continue
        ref_segs, itu, itl, izct = eo.detect(x, fs)
        assert segs[k] == ref_segs, (k, segs[k], ref_segs)
        found += len(ref_segs)
        np.testing.assert_allclose(thr[k], [itu, itl, izct], rtol=1e-11)
    assert found >= 20


# ------------------------------------------------------------------ hmmlearn-style class
def test_hmmlearn_style_class(sra, ctx):
    """GaussianHMM as HMM_lib.py uses it: k-means / variance initialisation, EM from
    that start equal to the oracle's EM from the same start, params / init_params,
    the "map" decoder.  (hmmlearn itself is absent: parity with it is unpinned; the EM
    arithmetic is the pinned one.)"""
    rng = np.random.default_rng(4)
    S, D = 3, 13
    centres = 3.0 * rng.standard_normal((S, D))
    lengths = [60, 45, 80, 30]
    states = np.concatenate([np.sort(rng.integers(0, S, size=n)) for n in lengths])
    X = centres[states] + rng.standard_normal((sum(lengths), D))
    m = sra.GaussianHMM(n_components=S, tol=1e-4, n_iter=6, implementation="log", random_state=0)
    m.init(X, lengths)
    np.testing.assert_allclose(m.startprob_, 1.0 / S)
    np.testing.assert_allclose(m.transmat_, 1.0 / S)
    assert np.linalg.norm(m.means_[:, None] - centres[None], axis=2).min(axis=1).max() < 1.5
    np.testing.assert_allclose(m._covars_, np.tile(X.var(axis=0, ddof=1) + 1e-3, (S, 1)))
    p = ho.Params(m.startprob_, m.transmat_, m.means_, m._covars_)
    hist = []
    for _ in range(6):
        st, lps = ho.estep(p, X, lengths, loops=False)
        hist.append(lps.sum())
        p = ho.mstep_full(p, st)
        if len(hist) >= 2 and hist[-1] - hist[-2] < 1e-4:
            break
    m.fit(X, lengths)
    np.testing.assert_allclose(list(m.monitor_.history), hist, rtol=1e-9)
    np.testing.assert_allclose(m.means_, p.means, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(m.transmat_, p.A, rtol=1e-7, atol=1e-12)
    assert all(b >= a - 1e-6 for a, b in zip(hist, hist[1:]))
    # params: only the transitions move
    m2 = sra.GaussianHMM(n_components=S, n_iter=3, params="t", init_params="", random_state=0)
    m2.startprob_, m2.transmat_, m2.means_, m2._covars_ = (np.full(S, 1 / S), np.full((S, S), 1 / S),
                                                          p.means.copy(), p.covars.copy())
    m2.fit(X, lengths)
    np.testing.assert_array_equal(m2.means_, p.means)
    np.testing.assert_array_equal(m2.startprob_, np.full(S, 1 / S))
    assert np.abs(m2.transmat_ - 1 / S).max() > 0.05
    # decoders
    lp_v, path_v = m.decode(X, lengths)
    lp_m, path_m = m.decode(X, lengths, algorithm="map")
    post = m.predict_proba(X, lengths)
    np.testing.assert_array_equal(path_m, post.argmax(axis=1))
    assert np.mean(path_v == path_m) > 0.95


def test_empty_inputs_of_the_added_entry_points(sra, ctx):
    """No recordings / no utterances / only empty ones: nothing is launched out of range,
    shapes come back empty."""
    res = sra.endpoint.detect_batch(np.zeros(0, dtype=np.float32), [0], 16000, ctx=ctx)
    assert res.counts.numel() == 0 and res.energy.numel() == 0
    res = sra.endpoint.detect_batch(np.zeros(0, dtype=np.float32), [0, 0, 0], 16000, ctx=ctx)
    assert res.counts.cpu().tolist() == [0, 0] and res.segments_host() == [[], []]
    packed, new_so, _ = sra.endpoint.cut_batch(np.zeros(100, dtype=np.float32), [0, 100], 16000, ctx=ctx)
    assert packed.numel() == 0 and new_so.tolist() == [0, 0]
    rng = np.random.default_rng(0)
    bank = _bank(sra, ctx, [ho.random_params(rng, 5, 7)])
    for dtype in ("float32", "float64"):
        lp, gam = bank.posteriors(sra.Ragged(ctx, np.zeros((0, 7)), [0, 0, 0], dtype=dtype))
        assert gam.shape == (0, 5) and lp.cpu().tolist() == [0.0, 0.0, 0.0]
        lp, gam = bank.posteriors(sra.Ragged(ctx, rng.standard_normal((3, 7)), [0, 3, 0], dtype=dtype))
        assert gam.shape == (3, 5) and lp[0] == 0 and lp[2] == 0 and float(lp[1]) < 0
        np.testing.assert_allclose(gam.sum(dim=1).cpu().numpy(), 1.0, atol=1e-5)


def test_clean_recordings_matches_reference_chunks(sra, ctx, endpoint_golden):
    """The two raw recordings of the fixtures (three 1 s chunks each), cleaned in one
    batch: equal to the reference's per-chunk cuts put end to end."""
    g = endpoint_golden
    n = int(g["n"])
    recs = [[n - 6, n - 5, n - 4], [n - 3, n - 2, n - 1]]          # chunk indices per recording
    signals = [np.concatenate([g[f"x{i}"] for i in idx]) for idx in recs]
    assert all(len(s) == 66150 for s in signals)
    for dtype in ("float32", "float64"):
        got = sra.endpoint.clean_recordings(signals, 22050, dtype=dtype, ctx=ctx)
        for idx, out in zip(recs, got):
            want = np.concatenate([g[f"cut{i}"] for i in idx])
            np.testing.assert_array_equal(out.astype(np.float32), want)
    assert sra.endpoint.clean_recordings([], 22050, ctx=ctx) == []


def test_random_shapes_against_oracle():
    """tools/fuzz_shapes.py, 60 seeded cases: random numbers of models (1..39), states
    (1..16), features (1..48), ragged batches with empty utterances, through scoring,
    Viterbi, posteriors and the E-step in float32 against the numpy oracle (1e-4 relative on
    log-probabilities, 5e-5 on posteriors and statistics)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_shapes.py"), "60", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "60 cases ok" in r.stdout


def test_random_shapes_against_oracle_float64():
    """The same fuzzer in float64, 60 seeded cases (states up to 40, features up to 69): models
    of up to 16 states and 48 features score on the f64 matrix-core kernel, the rest on the
    shape-generic kernels; tolerances 1e-10 relative on log-probabilities."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_shapes.py"), "60", "9", "float64"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "60 cases ok" in r.stdout


@pytest.mark.parametrize("M,S,D", [(1, 1, 1), (5, 4, 9), (7, 3, 48), (3, 8, 40), (9, 7, 17), (33, 2, 5),
                                   (2, 16, 39), (3, 9, 13), (5, 13, 48)])
def test_float64_matrix_core_scoring_equals_generic(sra, ctx, M, S, D):
    """srx_forward_score in float64 (f64 matrix instruction) against the shape-generic
    float64 kernel on ragged batches with empty and one-frame utterances, banks that end
    inside a tile and banks of many tiles: 1e-12 relative."""
    rng = np.random.default_rng(100 * M + 10 * S + D)
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    lengths = np.array([0, 1, 2, 37, 0, 120, 5, 64, 1, 0, 33, 16, 17, 250, 3, 8, 9, 0], dtype=np.int64)
    X = rng.standard_normal((int(lengths.sum()), D))
    bank = _bank(sra, ctx, prms)
    data = sra.Ragged(ctx, X, lengths, dtype="float64")
    fast = bank.score(data).cpu().numpy()
    gen = bank.score(data, generic=True).cpu().numpy()
    assert fast.shape == (len(lengths), M)
    live = lengths > 0
    np.testing.assert_allclose(fast[live], gen[live], rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(fast[~live], gen[~live])
    for k in (3, 13):                                   # and against the oracle itself
        x = X[lengths[:k].sum():lengths[:k + 1].sum()]
        for mi in (0, M - 1):
            np.testing.assert_allclose(fast[k, mi], ho.score(prms[mi], x, loops=False), rtol=1e-11)


def test_offset_features_stay_within_the_expanded_form_bound(sra, ctx):
    """Features that are not z-scored (mean 3e2, variances 1e-2 .. 1e2): the matrix-core
    paths evaluate cst + x mu/var - x^2/(2 var), whose terms cancel here.  Scores stay within
    eps * sum over frames of max_s sum_d (|x mu|/var + x^2/(2 var)) of the oracle's, eps the
    unit of the path's arithmetic: float64 keeps 1e-9 relative even here, float32 does not
    (the front-end's z-scored features are what the float32 paths are for)."""
    rng = np.random.default_rng(77)
    M, S, D = 3, 5, 13
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    var0 = 10.0 ** rng.uniform(-2, 2, size=(S, D))
    mu0 = 300.0 + 2.0 * np.sqrt(var0) * rng.standard_normal((S, D))
    for p in prms:                                            # word models a fraction of a sigma apart
        p.covars[:] = var0 * rng.uniform(0.8, 1.25, size=(S, D))
        p.means[:] = mu0 + 0.3 * np.sqrt(var0) * rng.standard_normal((S, D))
    lengths = np.array([40, 1, 120, 64], dtype=np.int64)
    states = rng.integers(0, S, size=int(lengths.sum()))
    X = prms[0].means[states] + np.sqrt(prms[0].covars[states]) * rng.standard_normal((len(states), D))
    X = X.astype(np.float32).astype(np.float64)              # the same numbers in both dtypes
    off = np.concatenate([[0], np.cumsum(lengths)])
    want = np.array([[float(np.sum(ho.score(p, X[off[u]:off[u + 1]], loops=False))) for p in prms]
                     for u in range(len(lengths))])
    mag = np.zeros_like(want)
    for mi, p in enumerate(prms):
        terms = (np.abs(X[:, None, :] * p.means[None]) / p.covars[None] + X[:, None, :] ** 2 / (2 * p.covars[None])).sum(-1)
        for u in range(len(lengths)):
            mag[u, mi] = terms[off[u]:off[u + 1]].max(axis=1).sum()
    bank = _bank(sra, ctx, prms)
    assert bank.conditioning() > 1e6
    got64 = bank.score(sra.Ragged(ctx, X, lengths, dtype="float64")).cpu().numpy()
    e64 = np.abs(got64 - want)
    assert (e64 <= 16 * 2.0 ** -52 * mag).all() and (e64 <= 1e-9 * np.abs(want)).all()
    gen64 = bank.score(sra.Ragged(ctx, X, lengths, dtype="float64"), generic=True).cpu().numpy()
    np.testing.assert_allclose(gen64, want, rtol=1e-12)
    got32 = bank.score(sra.Ragged(ctx, X, lengths, dtype="float32")).cpu().numpy()
    e32 = np.abs(got32 - want)
    print("offset features: float64 error / bound", float((e64 / (2.0 ** -52 * mag)).max()),
          "float32 (split) error / bound", float((e32 / (2.0 ** -16 * mag)).max()),
          "relative", float((e32 / np.abs(want)).max()))
    assert (e32 <= 4 * 2.0 ** -16 * mag).all()
    ctx.set_exact_f32(True)
    try:
        gotx = bank.score(sra.Ragged(ctx, X, lengths, dtype="float32")).cpu().numpy()
    finally:
        ctx.set_exact_f32(False)
    # the model class says so when asked for float32 on such a model, and float64 agrees
    m = sra.MyGaussianHMM(n_components=S, verbose=False, dtype="float32")
    m.startprob_, m.transmat_, m.means_, m._covars_ = prms[0].pi, prms[0].A, prms[0].means, prms[0].covars
    with pytest.warns(RuntimeWarning, match="standard deviations from zero"):
        m.score(X, lengths)
    m.dtype = "float64"
    assert abs(m.score(X, lengths) - want[:, 0].sum()) <= 1e-9 * abs(want[:, 0].sum())
    ex = np.abs(gotx - want)
    print("exact float32 error / bound", float((ex / (2.0 ** -23 * mag)).max()), "relative", float((ex / np.abs(want)).max()))
    assert (ex <= 16 * 2.0 ** -23 * mag).all()


@pytest.mark.parametrize("S,D,spread,alpha,tol", [(16, 39, 3.0, 0.2, 1e-5), (5, 20, 3.0, 0.2, 1e-5),
                                                  (16, 13, 6.0, 0.05, 5e-5), (8, 39, 4.0, 0.05, 5e-5)])
def test_fused_estep_on_peaked_models(sra, ctx, S, D, spread, alpha, tol):
    """The fused float32 E-step takes b~_t / c_t of its backward pass from the stored forward
    vectors (alpha_t over the prediction A^T alpha_{t-1}).  Models with far-apart means, narrow
    variances and sparse transition rows make those predictions and forward entries very small
    or zero: utterances sampled from such models, statistics finite and equal to the float64
    kernels' at 1e-5 of each array's largest entry (5e-5 for the widest spreads, where the
    float32 emission itself -- mu^2 / sigma^2 of several thousand -- is no better; the
    log-probability within 10 times that per frame for the same reason), and the float64 kernels equal
    to the oracle on one model's utterances."""
    rng = np.random.default_rng(S * 7 + D)
    M, U = 3, 96
    prms = []
    for _ in range(M):
        prms.append(ho.Params(rng.dirichlet(np.ones(S)), rng.dirichlet(np.ones(S) * alpha, size=S),
                              rng.standard_normal((S, D)) * spread, rng.uniform(0.02, 0.1, (S, D))))
    lengths = rng.integers(1, 120, size=U)
    um = (np.arange(U) % M).astype(np.int32)
    chunks = []
    for u in range(U):
        pr = prms[um[u]]
        st = rng.choice(S, p=pr.pi)
        rows = []
        for _ in range(lengths[u]):
            rows.append(pr.means[st] + rng.standard_normal(D) * np.sqrt(pr.covars[st]))
            st = rng.choice(S, p=pr.A[st])
        chunks.append(np.array(rows))
    X = np.vstack(chunks)
    bank = _bank(sra, ctx, prms)
    umd = ctx.tensor(um)
    ctx.set_estep_path("fused")
    try:
        s32 = bank.estep(sra.Ragged(ctx, X, lengths, dtype="float32"), umd).cpu().numpy()
    finally:
        ctx.set_estep_path("auto")
    s64 = bank.estep(sra.Ragged(ctx, X, lengths, dtype="float64"), umd).cpu().numpy()
    assert np.isfinite(s32).all() and np.isfinite(s64).all()
    g32, g64 = bank.unpack_stats(ctx.tensor(s32)), bank.unpack_stats(ctx.tensor(s64))
    for m in range(M):
        assert g32[m]["nobs"] == g64[m]["nobs"]
        assert abs(g32[m]["logprob"] - g64[m]["logprob"]) <= 10 * tol * lengths[um == m].sum()
        for k in ("start", "trans", "post", "obs", "obs**2"):
            np.testing.assert_allclose(g32[m][k], g64[m][k], rtol=0, atol=tol * np.abs(g64[m][k]).max())
    # posteriors of the same batch (models of more than 8 states: the same kernel, writing gamma)
    lp32, p32 = bank.posteriors(sra.Ragged(ctx, X, lengths, dtype="float32"), umd)
    lp64, p64 = bank.posteriors(sra.Ragged(ctx, X, lengths, dtype="float64"), umd)
    p32, p64 = p32.cpu().numpy().astype(np.float64), p64.cpu().numpy()
    assert np.isfinite(p32).all()
    print("posteriors: largest |float32 - float64|", float(np.abs(p32 - p64).max()))
    np.testing.assert_allclose(p32, p64, rtol=0, atol=20 * tol)
    np.testing.assert_allclose(p32.sum(axis=1), 1.0, rtol=0, atol=1e-5)
    # the float64 kernels against the oracle on model 0's utterances
    sel = [u for u in range(U) if um[u] == 0]
    off = np.concatenate([[0], np.cumsum(lengths)])
    st, lps = ho.estep(prms[0], np.vstack([X[off[u]:off[u + 1]] for u in sel]), [int(lengths[u]) for u in sel],
                       loops=False)
    assert abs(g64[0]["logprob"] - float(np.sum(lps))) <= 1e-9 * abs(float(np.sum(lps)))
    for k in ("start", "trans", "post", "obs", "obs**2"):
        np.testing.assert_allclose(g64[0][k], st[k], rtol=0, atol=1e-9 * max(1.0, np.abs(st[k]).max()))


@pytest.mark.parametrize("eps", [1e-33, 1e-36, 3e-31])
def test_fused_estep_rare_transition_with_dominant_emission(sra, ctx, eps):
    """A state that the transition matrix reaches with probability ~1e-33 but whose emission
    beats every other state's by e^104: the forward pass gives it alpha = 1, and the backward
    pass of the fused float32 kernel, which divides alpha_t by the prediction A^T alpha_{t-1}
    (here ~1e-33, far below any threshold that is safe to call "zero" short of the subnormal
    range), has to carry it too.  Statistics and posteriors against the float64 kernels, and
    those against the oracle."""
    rng = np.random.default_rng(5)
    S, D, U, m = 16, 13, 40, 4.0
    pi = np.zeros(S); pi[0] = 1.0
    A = np.eye(S)
    A[0, 0], A[0, 1] = 1.0 - eps, eps
    A[1, 0], A[1, 1] = 0.5, 0.5
    means = rng.standard_normal((S, D)) * 0.01
    means[1] += m
    means[2:] -= 3 * m                       # unreachable states, far from every frame
    prm = ho.Params(pi, A, means, np.ones((S, D)))
    lengths = rng.integers(3, 60, size=U)
    chunks = []
    for T in lengths:
        x = rng.standard_normal((T, D)) * 0.3
        t0 = rng.integers(1, T)              # from here on the frames sit on state 1's mean for a while
        x[t0:t0 + rng.integers(1, 6)] += m
        chunks.append(x)
    X = np.vstack(chunks)
    bank = _bank(sra, ctx, [prm])
    ctx.set_estep_path("fused")
    try:
        s32 = bank.estep(sra.Ragged(ctx, X, lengths, dtype="float32")).cpu().numpy()
    finally:
        ctx.set_estep_path("auto")
    s64 = bank.estep(sra.Ragged(ctx, X, lengths, dtype="float64")).cpu().numpy()
    assert np.isfinite(s32).all() and np.isfinite(s64).all()
    g32, g64 = bank.unpack_stats(ctx.tensor(s32))[0], bank.unpack_stats(ctx.tensor(s64))[0]
    st, lps = ho.estep(prm, X, [int(t) for t in lengths], loops=False)
    assert abs(g64["logprob"] - float(np.sum(lps))) <= 1e-9 * abs(float(np.sum(lps)))
    assert st["post"][1] > U                 # the rare state does hold the frames moved onto it
    assert abs(g32["logprob"] - g64["logprob"]) <= 1e-5 * abs(g64["logprob"])
    for k in ("start", "trans", "post", "obs", "obs**2"):
        np.testing.assert_allclose(g64[k], st[k], rtol=0, atol=1e-9 * max(1.0, np.abs(st[k]).max()))
        np.testing.assert_allclose(g32[k], g64[k], rtol=0, atol=1e-5 * np.abs(g64[k]).max(), err_msg=k)
    lp32, p32 = bank.posteriors(sra.Ragged(ctx, X, lengths, dtype="float32"))
    lp64, p64 = bank.posteriors(sra.Ragged(ctx, X, lengths, dtype="float64"))
    p32, p64 = p32.cpu().numpy().astype(np.float64), p64.cpu().numpy()
    assert np.isfinite(p32).all() and p64[:, 1].max() > 0.999
    # (eps = 1e-36: a frame whose other state is 2^-135 of the rare one in density still has
    # 2e-5 of the posterior in float64, but that ratio is below the smallest normal float32 and
    # the float32 exponential returns zero for it: the range of the dtype, not of the division)
    np.testing.assert_allclose(p32, p64, rtol=0, atol=1e-5 if eps > 1e-35 else 1e-4)
    np.testing.assert_allclose(lp32.cpu().numpy(), lp64.cpu().numpy(), rtol=1e-5)


@pytest.mark.parametrize("dtype,path", [("float32", "auto"), ("float64", "auto"), ("float32", "fused"),
                                        ("float64", "fused")])
def test_estep_for_a_transitions_only_mstep(sra, ctx, dtype, path):
    """srx_estep_stats_select with SRX_STATS_TRANSITIONS: count, log-probability, start and transition sums
    are the ones of the full E-step bit for bit, the sums over the features complete or all
    zero (zero for a small job, which skips its pass over X), and the reference's M-step
    (start and transition probabilities only) gives the same parameters from either."""
    import torch
    rng = np.random.default_rng(21)
    M, S, D = 4, 3, 26
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    lengths = rng.integers(1, 80, size=60)
    um = rng.integers(0, M, size=len(lengths)).astype(np.int32)
    X = rng.standard_normal((int(lengths.sum()), D))
    bank = _bank(sra, ctx, prms)
    data = sra.Ragged(ctx, X, lengths, dtype=dtype)
    umd = ctx.tensor(um)
    ctx.set_estep_path(path)
    try:
        full = bank.estep(data, umd).clone()
        part = bank.estep(data, umd, stats="transitions").clone()
        again = bank.estep(data, umd).clone()          # the selection does not outlive its call
    finally:
        ctx.set_estep_path("auto")
    assert again[:, 2:].abs().sum() > 0 and torch.allclose(again, full, rtol=1e-12, atol=0)
    gf, gp = bank.unpack_stats(full), bank.unpack_stats(part)
    for m in range(M):
        assert gp[m]["nobs"] == gf[m]["nobs"]
        if path == "fused":         # (sums by float64 atomics: equal to the rounding of their order)
            assert abs(gp[m]["logprob"] - gf[m]["logprob"]) <= 1e-12 * abs(gf[m]["logprob"])
            np.testing.assert_allclose(gp[m]["start"], gf[m]["start"], rtol=1e-12, atol=1e-300)
            np.testing.assert_allclose(gp[m]["trans"], gf[m]["trans"], rtol=1e-12, atol=1e-300)
        else:
            assert gp[m]["logprob"] == gf[m]["logprob"]
            np.testing.assert_array_equal(gp[m]["start"], gf[m]["start"])
            np.testing.assert_array_equal(gp[m]["trans"], gf[m]["trans"])
        for k in ("post", "obs", "obs**2"):
            assert not gp[m][k].any() or np.array_equal(gp[m][k], gf[m][k])
    assert not gp[0]["obs"].any()                      # both kernels leave the feature sums out
    bank.mstep(part, mode="transitions_only")
    a1, p1 = bank.transmat.clone(), bank.startprob.clone()
    bank2 = _bank(sra, ctx, prms)
    bank2.mstep(full, mode="transitions_only")
    assert torch.allclose(a1, bank2.transmat, rtol=1e-12, atol=0) and torch.allclose(p1, bank2.startprob, rtol=1e-12, atol=0)


@pytest.mark.parametrize("M,S,D", [(1, 1, 1), (6, 3, 26), (5, 2, 9), (40, 4, 13), (7, 5, 48), (3, 6, 17),
                                   (9, 7, 33), (6, 8, 39)])
def test_small_job_estep_paths_agree_with_oracle(sra, ctx, M, S, D):
    """The lane-per-utterance E-step of small jobs (float32 up to 8 states, float64 up to 4 by
    default, any of them when asked for) on ragged batches with empty and one-frame
    utterances, a model without utterances, 130 utterances (three workgroups of the lane
    kernel, the last one partly empty): per-model statistics against the oracle, and against
    the other kernels for the same dtype."""
    rng = np.random.default_rng(1000 * M + 10 * S + D)
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    U = 130
    lengths = rng.integers(1, 90, size=U)
    lengths[[3, 64, 129]] = 0
    lengths[[5, 70]] = 1
    lengths[7] = 2
    um = rng.integers(0, M, size=U).astype(np.int32)
    if M > 2:
        um[um == 1] = 0                                   # model 1 trains on nothing
    X = rng.standard_normal((int(lengths.sum()), D))
    off = np.concatenate([[0], np.cumsum(lengths)])
    bank = _bank(sra, ctx, prms)
    want = []
    for m in range(M):
        sel = [u for u in range(U) if um[u] == m and lengths[u] > 0]
        if not sel:
            want.append(None)
            continue
        Xm = np.vstack([X[off[u]:off[u + 1]] for u in sel])
        st, lps = ho.estep(prms[m], Xm, [int(lengths[u]) for u in sel], loops=False)
        want.append((st, float(np.sum(lps)), len(sel)))
    umd = ctx.tensor(um)
    res = {}
    for dtype, tol in (("float32", 2e-5), ("float64", 1e-9)):
        data = sra.Ragged(ctx, X, lengths, dtype=dtype)
        for path in ("lane", "fused"):
            ctx.set_estep_path(path)
            try:
                raw = bank.estep(data, umd).cpu().numpy()
            finally:
                ctx.set_estep_path("auto")
            res[dtype, path] = raw
            got = bank.unpack_stats(ctx.tensor(raw))
            for m in range(M):
                if want[m] is None:
                    assert got[m]["nobs"] == 0 and got[m]["logprob"] == 0.0
                    continue
                st, lp, n = want[m]
                assert got[m]["nobs"] == n               # (empty utterances are not counted)
                assert abs(got[m]["logprob"] - lp) <= tol * abs(lp)
                for k in ("start", "trans", "post", "obs", "obs**2"):
                    np.testing.assert_allclose(got[m][k], st[k], rtol=tol, atol=tol * max(1.0, np.abs(st[k]).max()))
        scale = np.abs(res[dtype, "fused"]).max()
        assert np.abs(res[dtype, "lane"] - res[dtype, "fused"]).max() <= 2 * tol * scale


@pytest.mark.parametrize("path", ["wave", "lane", "fused", "generic"])
@pytest.mark.parametrize("dtype,tol", [("float32", 2e-5), ("float64", 1e-9)])
def test_estep_forced_path_through_unlikely_states(sra, ctx, dtype, tol, path):
    """A strict left-to-right chain of four states whose data sits on the last state from the
    first frame on: the chain has to walk through three states that each cost e^-60 per frame.
    Every frame still has a float32 scale factor (e^-60), so the scaled recursions are defined,
    but two things go out of float32's range on the way.  (1) A product of three frames'
    transfer matrices holds the row of start state 0 (the only one the chain can be in) at
    e^-120 of the row of start state 3: the wave kernel's matrices carry an exponent per row.
    (2) beta of the states the chain is NOT in grows like 1 / alpha^ and passes the largest
    float (the reference's recursion in float32 gives 0 x inf = NaN there): every kernel holds
    beta at the largest finite value, and the wave kernel scales it by sum alpha^ beta = 1
    like the sequential recursion, not by its largest entry."""
    S, D = 4, 13
    rng = np.random.default_rng(123)
    A = np.zeros((S, S))
    for i in range(S):
        A[i, i] = 0.6 if i + 1 < S else 1.0
        if i + 1 < S:
            A[i, i + 1] = 0.4
    pi = np.zeros(S); pi[0] = 1.0
    means = np.zeros((S, D))
    for i in range(S - 1):
        means[i] = (3.0 + 0.1 * i) * (1 if i % 2 == 0 else -1)      # 13 * 3^2 / 2 = 58 nats from the data
    prm = ho.Params(pi, A, means, np.ones((S, D)))
    # (600 and 1000 frames: beyond what the wave kernel keeps in LDS in float64 / float32)
    lengths = np.array([130, 65, 200, 320, 5, 64, 600, 1000])
    X = 0.3 * rng.standard_normal((int(lengths.sum()), D))
    bank = _bank(sra, ctx, [prm])
    data = sra.Ragged(ctx, X, lengths, dtype=dtype)
    generic = path == "generic"
    ctx.set_estep_path("auto" if generic else path)
    try:
        got = bank.unpack_stats(bank.estep(data, generic=generic))[0]
    finally:
        ctx.set_estep_path("auto")
    st, lps = ho.estep(prm, X, [int(t) for t in lengths], loops=False)
    lp = float(np.sum(lps))
    assert np.isfinite(got["logprob"]) and abs(got["logprob"] - lp) <= tol * abs(lp)
    for k in ("start", "trans", "post", "obs", "obs**2"):
        np.testing.assert_allclose(got[k], st[k], rtol=tol, atol=tol * max(1.0, np.abs(st[k]).max()), err_msg=k)
    if path == "wave":          # (once per dtype: the posteriors of the same batch)
        plp, gamma = bank.posteriors(data)
        gamma = gamma.cpu().numpy()
        assert np.isfinite(gamma).all()
        np.testing.assert_allclose(gamma.sum(axis=1), 1.0, atol=1e-5)
        np.testing.assert_allclose(plp.cpu().numpy(), lps, rtol=tol)
        _, ref = ho.score_posteriors(prm, X, [int(t) for t in lengths], loops=False)
        np.testing.assert_allclose(gamma, ref, atol=5e-5 if dtype == "float32" else 1e-9)


@pytest.mark.parametrize("S,D", [(3, 26), (2, 13)])
def test_wave_estep_left_to_right_model_on_foreign_data(sra, ctx, S, D):
    """A left-to-right model scored on frames it was not drawn from (float64; in float32 such
    frames have no density in any kernel).  The likelihood of a run of frames then differs
    between start states by hundreds of nats, most of them start states the chain is not in:
    the transfer-matrix products of the wave kernel have to carry every row on a scale of its
    own to keep the one that the forward vector selects."""
    rng = np.random.default_rng(5 * S + D)
    prm = ho.random_params(rng, S, D)
    A = np.triu(prm.A) + 1e-300
    A = np.triu(A / A.sum(axis=1, keepdims=True))
    pi = np.zeros(S); pi[0] = 1.0
    prm = ho.Params(pi, A / A.sum(axis=1, keepdims=True), 3.0 * prm.means, prm.covars)
    lengths = rng.integers(1, 500, size=24)
    lengths[:4] = [1, 2, 64, 499]
    X = 2.0 * rng.standard_normal((int(lengths.sum()), D))
    bank = _bank(sra, ctx, [prm])
    data = sra.Ragged(ctx, X, lengths, dtype="float64")
    ctx.set_estep_path("wave")
    try:
        got = bank.unpack_stats(bank.estep(data))[0]
    finally:
        ctx.set_estep_path("auto")
    st, lps = ho.estep(prm, X, [int(t) for t in lengths], loops=False)
    lp = float(np.sum(lps))
    assert np.isfinite(got["logprob"]) and abs(got["logprob"] - lp) <= 1e-9 * abs(lp)
    for k in ("start", "trans", "post", "obs", "obs**2"):
        np.testing.assert_allclose(got[k], st[k], rtol=1e-9, atol=1e-9 * max(1.0, np.abs(st[k]).max()), err_msg=k)


@pytest.mark.parametrize("M,S,D,tmax", [(1, 1, 1, 90), (6, 3, 26, 240), (5, 2, 9, 700), (4, 4, 39, 130),
                                        (2, 3, 100, 64)])
def test_wave_per_utterance_estep_against_oracle(sra, ctx, M, S, D, tmax):
    """The E-step of the smallest jobs (one wave per utterance, recursions cut into 64 pieces
    joined by scans of transfer-matrix products; what "auto" picks for at most 2048 utterances of
    models with at most 4 states): ragged batches with empty, one- and two-frame utterances,
    utterances shorter and much longer than 64 frames, a model without utterances, a left-to-right
    model with structural zeros, both dtypes; statistics against the oracle, the
    transitions-only selection against the full one, and the same call twice bit for bit."""
    rng = np.random.default_rng(77 * M + 10 * S + D)
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    if S >= 3:                                        # structural zeros: left to right, start in state 0
        A = np.triu(prms[0].A) + 1e-300
        A = np.triu(A / A.sum(axis=1, keepdims=True))
        pi = np.zeros(S); pi[0] = 1.0
        prms[0] = ho.Params(pi, A / A.sum(axis=1, keepdims=True), prms[0].means, prms[0].covars)
    U = 70
    lengths = rng.integers(1, tmax, size=U)
    lengths[[3, 64]] = 0
    lengths[[5, 69]] = 1
    lengths[7] = 2
    lengths[9], lengths[10], lengths[11] = 63, 64, 65
    um = rng.integers(0, M, size=U).astype(np.int32)
    if M > 2:
        um[um == 1] = 0
    X = rng.standard_normal((int(lengths.sum()), D))
    off = np.concatenate([[0], np.cumsum(lengths)])
    if S >= 3:
        # the left-to-right model's utterances are drawn from it (frames far from every state
        # the chain can be in have no float32 density at all: the dtype's range, not the kernel's)
        for u in np.flatnonzero(um == 0):
            st = 0
            for t in range(off[u], off[u + 1]):
                X[t] = prms[0].means[st] + rng.standard_normal(D) * np.sqrt(prms[0].covars[st])
                st = rng.choice(S, p=prms[0].A[st])
    bank = _bank(sra, ctx, prms)
    umd = ctx.tensor(um)
    for dtype, tol in (("float32", 2e-5), ("float64", 1e-9)):
        data = sra.Ragged(ctx, X, lengths, dtype=dtype)
        ctx.set_estep_path("wave")
        try:
            raw = bank.estep(data, umd).cpu().numpy()
            raw2 = bank.estep(data, umd).cpu().numpy()
            part = bank.estep(data, umd, stats="transitions").cpu().numpy()
        finally:
            ctx.set_estep_path("auto")
        auto = bank.estep(data, umd).cpu().numpy()
        assert np.array_equal(raw, raw2) and np.array_equal(raw, auto)      # "auto" takes this path here
        got, gp = bank.unpack_stats(ctx.tensor(raw)), bank.unpack_stats(ctx.tensor(part))
        for m in range(M):
            sel = [u for u in range(U) if um[u] == m and lengths[u] > 0]
            if not sel:
                assert got[m]["nobs"] == 0 and got[m]["logprob"] == 0.0 and not got[m]["obs"].any()
                continue
            Xm = np.vstack([X[off[u]:off[u + 1]] for u in sel])
            st, lps = ho.estep(prms[m], Xm, [int(lengths[u]) for u in sel], loops=False)
            lp = float(np.sum(lps))
            assert got[m]["nobs"] == len(sel)
            assert abs(got[m]["logprob"] - lp) <= tol * abs(lp)
            for k in ("start", "trans", "post", "obs", "obs**2"):
                np.testing.assert_allclose(got[m][k], st[k], rtol=tol, atol=tol * max(1.0, np.abs(st[k]).max()),
                                           err_msg=f"{dtype} model {m} {k}")
            assert gp[m]["logprob"] == got[m]["logprob"] and gp[m]["nobs"] == got[m]["nobs"]
            assert np.array_equal(gp[m]["trans"], got[m]["trans"]) and np.array_equal(gp[m]["start"], got[m]["start"])
            assert not gp[m]["obs"].any() and not gp[m]["post"].any()
    # structural zeros of the transition matrix stay exact zeros in the xi sums
    if S >= 3:
        assert np.all(got[0]["trans"][np.tril_indices(S, -1)] == 0.0)


def test_scoring_with_a_length_order_is_bit_identical(sra, ctx):
    """Ragged float32 batches are scored through srx_forward_score_ordered (utterances of
    similar length share a wave): bit for bit the scores of srx_forward_score, which takes
    them in memory order, and within the float32 bound of the oracle's."""
    import ctypes as C
    import torch
    from speech_recognition_amd import _lib
    rng = np.random.default_rng(21)
    M, S, D, U = 6, 8, 39, 333
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    lengths = rng.integers(0, 120, size=U)
    lengths[[0, 17, 332]] = [0, 1, 119]
    X = rng.standard_normal((int(lengths.sum()), D))
    bank = _bank(sra, ctx, prms)
    data = sra.Ragged(ctx, X, lengths, dtype="float32")
    order = data.length_order()
    assert order is not None and sorted(order.cpu().tolist()) == list(range(U))
    got = bank.score(data)
    plain = torch.empty_like(got)
    _lib.check(_lib.lib.srx_forward_score(ctx.handle, data.X.data_ptr(), data.code, data.offsets.data_ptr(),
                                          data.U, data.n_frames, C.byref(bank._struct), plain.data_ptr(),
                                          ctx.stream_ptr()))
    assert torch.equal(got, plain)
    off = np.concatenate([[0], np.cumsum(lengths)])
    for u in (1, 17, 100, 332):
        for m in (0, M - 1):
            want = float(np.sum(ho.score(prms[m], X[off[u]:off[u + 1]], loops=False)))
            assert abs(float(got[u, m]) - want) <= 2e-5 * abs(want)
    # equal lengths: no order is made
    assert sra.Ragged(ctx, X[:4000], [40] * 100, dtype="float32").length_order() is None


@pytest.mark.parametrize("S,path", [(3, "lane"), (3, "fused"), (16, "fused")])
def test_estep_long_utterances(sra, ctx, S, path):
    """Utterances of thousands of frames: the float32 accumulators of both E-step kernels are
    folded into float64 every few hundred frames, so the statistics keep the 2e-5 of short
    utterances."""
    rng = np.random.default_rng(300 + S)
    D, M = 13, 2
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    lengths = np.array([5000, 1, 2300, 700, 4100, 2049, 513, 64, 3000, 1025, 2048, 511, 512, 4096, 333, 1500,
                        2600, 40], dtype=np.int64)
    um = (np.arange(len(lengths)) % M).astype(np.int32)
    X = rng.standard_normal((int(lengths.sum()), D))
    off = np.concatenate([[0], np.cumsum(lengths)])
    bank = _bank(sra, ctx, prms)
    ctx.set_estep_path(path)
    try:
        got = bank.unpack_stats(bank.estep(sra.Ragged(ctx, X, lengths, dtype="float32"), ctx.tensor(um)))
    finally:
        ctx.set_estep_path("auto")
    for m in range(M):
        sel = [u for u in range(len(lengths)) if um[u] == m]
        Xm = np.vstack([X[off[u]:off[u + 1]] for u in sel])
        st, lps = ho.estep(prms[m], Xm, [int(lengths[u]) for u in sel], loops=False)
        assert got[m]["nobs"] == len(sel)
        assert abs(got[m]["logprob"] - float(np.sum(lps))) <= 1e-5 * abs(float(np.sum(lps)))
        for k in ("start", "trans", "post", "obs", "obs**2"):
            np.testing.assert_allclose(got[m][k], st[k], rtol=2e-5, atol=2e-5 * max(1.0, np.abs(st[k]).max()))


@pytest.mark.parametrize("S", [5, 12])
def test_viterbi_float32_long_utterances(sra, ctx, S):
    """Utterances of thousands of frames through the float32 Viterbi kernels (one lane per
    utterance up to 8 states, the sub-wave chain above): the lattice is kept relative to its
    maximum, so the float32 paths part from the float64 ones only at near-ties (1e-6 of the
    score here) and the log-probability keeps 1e-6 relative, whatever the length."""
    from conftest import assert_ties_only
    rng = np.random.default_rng(40 + S)
    D = 13
    prm = ho.random_params(rng, S, D)
    lengths = np.array([6000, 1, 2500, 17, 4097], dtype=np.int64)
    X = rng.standard_normal((int(lengths.sum()), D)).astype(np.float32).astype(np.float64)
    bank = _bank(sra, ctx, [prm])
    lp64, path64 = bank.viterbi(sra.Ragged(ctx, X, lengths, dtype="float64"))
    lp32, path32 = bank.viterbi(sra.Ragged(ctx, X, lengths, dtype="float32"))
    lp64, path64, lp32, path32 = (v.cpu().numpy() for v in (lp64, path64, lp32, path32))
    want_lp, want_path = ho.viterbi(prm, X[:6000])
    np.testing.assert_array_equal(path64[:6000], want_path)
    log_B = ho.log_density_diag(X, prm.means, prm.covars)
    n_diff, gap, short = assert_ties_only(prm.pi, prm.A, log_B, lengths, path32, path64, lp64, eps=1e-6)
    print(f"viterbi float32, {S} states, {int(lengths.sum())} frames: {n_diff} differ, largest tie gap {gap:.2e}")
    assert n_diff <= 20
    assert np.abs(lp32 - lp64).max() <= 1e-6 * np.abs(lp64).max()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_estep_without_utterances_or_frames(sra, ctx, dtype):
    """No utterance at all, and utterances that are all empty: the statistics are zeros (the
    dict of initialize_sufficient_statistics, hmm.py:224-235) in both dtypes."""
    rng = np.random.default_rng(4)
    prms = [ho.random_params(rng, 5, 7) for _ in range(3)]
    bank = _bank(sra, ctx, prms)
    for lengths in ([], [0, 0, 0]):
        data = sra.Ragged(ctx, np.zeros((0, 7)), lengths, dtype=dtype)
        um = np.zeros(len(lengths), dtype=np.int32)
        st = bank.estep(data, um).cpu().numpy()
        assert st.shape == (3, bank.stats_size)
        # every sum is zero, and an utterance without frames is not an observation
        assert np.all(st == 0.0), (dtype, lengths)


def test_random_front_end_configurations_against_oracle():
    """tools/fuzz_mfcc.py, 80 seeded cases: random sample rates, frame lengths, 512 / 1024
    point transforms, 8..64 filters, 3..16 cepstra, delta orders 0..2 with half-widths 1..3,
    with and without z-score, ragged batches with signals shorter than a frame, float32
    against the numpy oracle (1e-4 of the largest value un-normalised, 5e-3 absolute
    z-scored)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_mfcc.py"), "80", "11"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "80 cases ok" in r.stdout


@pytest.mark.parametrize("S,D", [(1, 5), (2, 13), (3, 26), (4, 39), (5, 7), (6, 13), (7, 24), (8, 26), (8, 39), (5, 48), (8, 40)])
def test_viterbi_fused_kernel_matches_oracle(sra, ctx, S, D):
    """The one-kernel float32 Viterbi (emission on the matrix cores, log-densities kept in LDS,
    four lanes per utterance in the max-plus recursion) on multi-model ragged batches that
    leave partial groups of 16 utterances and partial chunks of 16 frames, for every way the
    feature axis is cut into blocks of 16 and 8: paths equal the float64 oracle's
    (utils.py:243-266) except at near-ties, log-probabilities to 1e-5, and the two-kernel path
    gives the same states and log-probabilities bit for bit."""
    from conftest import assert_ties_only
    rng = np.random.default_rng(100 * S + D)
    M = 3
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    # a left-to-right model: structural zeros in the transition matrix and the start vector
    A = np.triu(prms[1].A) * (np.tril(np.ones((S, S)), 1))
    prms[1] = ho.Params(np.eye(S)[0], A / A.sum(axis=1, keepdims=True), prms[1].means, prms[1].covars)
    lengths = np.array([0, 1, 2, 15, 16, 17, 33, 200, 64, 31, 1, 48, 5, 129, 16, 0, 7, 96, 32, 3, 250,
                        18, 16, 160, 2, 80, 47, 49, 1, 100, 12, 65, 66, 15, 14, 201, 8, 300, 256, 257], dtype=np.int64)
    um = rng.integers(0, M, size=len(lengths)).astype(np.int32)
    X = rng.standard_normal((int(lengths.sum()), D)).astype(np.float32).astype(np.float64)
    bank = _bank(sra, ctx, prms)
    data = sra.Ragged(ctx, X, lengths, dtype="float32")
    got = {}
    for which in ("fused", "split"):
        ctx.set_viterbi_path(which)
        try:
            lp, path = bank.viterbi(data, um)
        finally:
            ctx.set_viterbi_path("auto")
        got[which] = (lp.cpu().numpy(), path.cpu().numpy())
    off = np.concatenate([[0], np.cumsum(lengths)])
    want_lp, want_path = np.zeros(len(lengths)), np.zeros(int(lengths.sum()), dtype=np.int64)
    log_B = np.zeros((int(lengths.sum()), S))
    for u, n in enumerate(lengths):
        if n == 0:
            continue
        a, b = off[u], off[u + 1]
        lp_u, want_path[a:b] = ho.viterbi(prms[um[u]], X[a:b])
        want_lp[u] = float(np.asarray(lp_u).reshape(-1)[0])
        log_B[a:b] = ho.log_density_diag(X[a:b], prms[um[u]].means, prms[um[u]].covars)
    for which, (lp, path) in got.items():
        for m in range(M):
            sel = np.nonzero(um == m)[0]
            # (assert_ties_only takes one model: check each model's utterances on their own)
            idx = np.concatenate([np.arange(off[u], off[u + 1]) for u in sel]) if len(sel) else np.zeros(0, int)
            n_diff, gap, short = assert_ties_only(prms[m].pi, prms[m].A, log_B[idx], lengths[sel], path[idx],
                                                  want_path[idx], want_lp[sel])
            assert n_diff <= 3, (which, m, n_diff)
        nz = lengths > 0
        assert np.abs(lp[nz] - want_lp[nz]).max() <= 1e-5 * np.abs(want_lp[nz]).max(), which
        assert np.all(lp[~nz] == 0.0)
    # the two paths do the same arithmetic term for term: which one a batch takes does not show
    np.testing.assert_array_equal(got["fused"][1], got["split"][1])
    np.testing.assert_array_equal(got["fused"][0], got["split"][0])
