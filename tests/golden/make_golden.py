"""Generate the golden vectors under tests/golden/ by running the reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py [mfcc] [hmm] [folds] [post] [endpoint] [harness]

The reference modules are imported as they are.  ``mfcc.py`` imports three
third-party modules that are absent here and that the code path never uses for
its result (``sounddevice.wait()``, ``seaborn``, and a ``librosa.feature.mfcc``
call whose result is discarded, mfcc.py:139); empty placeholder modules with
those names are put in ``sys.modules`` so the import succeeds.

Only data leaves this script: inputs and the reference's outputs, as .npz.
The reference's pickled models are NOT read (unpickling executes code).
"""
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
WORDS = ["odessa", "play_music", "stop_music", "turn_off_the_lights",
         "turn_on_the_lights", "what_time_is_it"]  # np.unique order


def _import_reference():
    sys.path.insert(0, REF)
    sd = types.ModuleType("sounddevice")
    sd.wait = lambda *a, **k: None
    sns = types.ModuleType("seaborn")
    lib = types.ModuleType("librosa")
    feat = types.ModuleType("librosa.feature")
    feat.mfcc = lambda **kw: None
    lib.feature = feat
    lib.load = lambda *a, **k: (None, None)
    sys.modules.update({"sounddevice": sd, "seaborn": sns, "librosa": lib,
                        "librosa.feature": feat})
    import matplotlib
    matplotlib.use("Agg")
    import mfcc as ref_mfcc
    import HMM_final as ref_final
    from my_hmmlearn import hmm as ref_hmm, utils as ref_utils
    return ref_mfcc, ref_final, ref_hmm, ref_utils


def _synthetic_signals():
    rng = np.random.default_rng(20240601)
    out = []
    # 16 kHz, 1.0 s: noise + three sinusoids
    t = np.arange(16000) / 16000.0
    s = 0.05 * rng.standard_normal(16000)
    for f, a in ((220.0, 0.1), (1330.0, 0.05), (3700.0, 0.03)):
        s += a * np.sin(2 * np.pi * f * t)
    out.append((s, 16000))
    # 22.05 kHz, 0.7 s chirp-ish + noise
    n = int(0.7 * 22050)
    t = np.arange(n) / 22050.0
    s = 0.1 * np.sin(2 * np.pi * (300.0 + 2000.0 * t) * t) + 0.02 * rng.standard_normal(n)
    out.append((s, 22050))
    # shortest input that still yields one output frame at 16 kHz: T = 5
    out.append((0.1 * rng.standard_normal(400 + 4 * 160), 16000))
    # silence then noise: exact-zero mel energies hit the eps floor
    s = np.concatenate([np.zeros(2000), 0.1 * rng.standard_normal(3000)])
    out.append((s, 16000))
    # all-zero signal: constant features, z-score keeps scale 1
    out.append((np.zeros(2400), 16000))
    return out


def make_mfcc(ref_mfcc, ref_final):
    from scipy.io import wavfile
    data = {}
    sigs = _synthetic_signals()
    real = ["odessa_5", "play_music_12", "turn_off_the_lights_3", "what_time_is_it_27"]
    for name in real:
        sr, s = wavfile.read(f"{REF}/cleaned_audios_3005/{name}.wav")
        sigs.append((np.asarray(s, dtype=np.float64), sr))
    with tempfile.TemporaryDirectory() as tmp:
        for i, (s, sr) in enumerate(sigs):
            s32 = np.asarray(s, dtype=np.float32)
            path = os.path.join(tmp, f"sig{i}.wav")
            wavfile.write(path, sr, s32)
            cep = ref_mfcc.compute_mfcc(wavfile.read(path)[1], sr)
            feat = ref_mfcc.calculate_mfcc_for_onerecording(False, path, sample_rate=sr)
            with np.errstate(all="ignore"):
                z = ref_final.extract_mfcc_calc(path, sr)
            data[f"sig{i}"] = s32
            data[f"sr{i}"] = np.int64(sr)
            data[f"cep{i}"] = cep            # (T, 13)
            data[f"feat{i}"] = feat          # (26, T-4)
            data[f"z{i}"] = z                # (T-4, 26)
    data["n"] = np.int64(len(sigs))
    data["fbank1024"] = ref_mfcc.mel_filterbank(1024)
    data["fbank512"] = ref_mfcc.mel_filterbank(512)
    np.savez_compressed(os.path.join(OUT, "mfcc_golden.npz"), **data)
    print("mfcc_golden.npz:", len(sigs), "signals")


def _ref_model(ref_hmm, pi, A, mu, var):
    m = ref_hmm.MyGaussianHMM(n_components=len(pi), implementation="scaling",
                              verbose=False)
    m.startprob_ = np.array(pi, dtype=float)
    m.transmat_ = np.array(A, dtype=float)
    m.means_ = np.array(mu, dtype=float)
    m._covars_ = np.array(var, dtype=float)
    m.n_features = m.means_.shape[1]
    return m


def _stats_arrays(prefix, st, d):
    d[prefix + "nobs"] = np.int64(st["nobs"])
    for k, n in (("start", "start"), ("trans", "trans"), ("post", "post"),
                 ("obs", "obs"), ("obs**2", "obs2")):
        d[prefix + n] = np.array(st[k])


def make_hmm(ref_hmm, ref_utils):
    rng = np.random.default_rng(7)
    cases = [
        ("s3d26", 3, 26, [57, 120, 33, 1, 2, 88], False),
        ("s8d39", 8, 39, [200, 200, 61, 5], False),
        ("s16d39", 16, 39, [200, 37, 140], False),
        ("s3d13", 3, 13, [90, 45, 1], False),
        ("s5d7lr", 5, 7, [64, 30, 12], True),   # left-to-right, zeros in pi/A
    ]
    data = {"cases": np.array([c[0] for c in cases])}
    for name, S, D, lengths, left_right in cases:
        pi = rng.dirichlet(np.ones(S))
        A = rng.dirichlet(np.ones(S), size=S)
        if left_right:
            pi = np.zeros(S)
            pi[0] = 1.0
            A = np.triu(A) * (np.tril(np.ones((S, S)), 1))
            A /= A.sum(axis=1, keepdims=True)
        mu = rng.standard_normal((S, D))
        var = rng.uniform(0.5, 1.5, (S, D))
        X = rng.standard_normal((sum(lengths), D))
        # make the data depend on the states a little so posteriors are not flat
        X += mu[rng.integers(0, S, size=len(X))] * 0.7
        m = _ref_model(ref_hmm, pi, A, mu, var)
        p = name + "_"
        data[p + "pi"], data[p + "A"], data[p + "mu"], data[p + "var"] = pi, A, mu, var
        data[p + "X"], data[p + "lengths"] = X, np.array(lengths)
        data[p + "loglik"] = m._log_multivariate_normal_density_diag(X, m.means_, m._covars_)
        subs = ref_utils.split_X_lengths(X, lengths)
        data[p + "utt_logprob"] = np.array([m.score(x) for x in subs])
        data[p + "score_all"] = np.float64(m.score(X, lengths))
        # lattices of the first utterance
        B, lp, g, al, be = m.fit_scaling(subs[0])
        data[p + "alpha0"], data[p + "beta0"], data[p + "gamma0"] = al, be, g
        data[p + "xi0"] = ref_utils.compute_scaling_xi_sum(al, m.transmat_, be, B)
        st, lp = m.do_estep(X, np.array(lengths))
        _stats_arrays(p + "st_", st, data)
        data[p + "logprob"] = np.float64(lp)
        # transitions-only M-step (what fit() calls, hmm.py:213)
        m1 = _ref_model(ref_hmm, pi, A, mu, var)
        m1.do_em_mstep(st)
        data[p + "em_pi"], data[p + "em_A"] = m1.startprob_, m1.transmat_
        # full M-step (do_mstep, hmm.py:165)
        m2 = _ref_model(ref_hmm, pi, A, mu, var)
        m2.do_mstep(st)
        data[p + "m_pi"], data[p + "m_A"] = m2.startprob_, m2.transmat_
        data[p + "m_mu"], data[p + "m_var"] = m2.means_, m2._covars_
        # second EM iteration from the re-estimated model
        st2, lp2 = m2.do_estep(X, np.array(lengths))
        _stats_arrays(p + "st2_", st2, data)
        data[p + "logprob2"] = np.float64(lp2)
        m2.do_mstep(st2)
        data[p + "m2_pi"], data[p + "m2_A"] = m2.startprob_, m2.transmat_
        data[p + "m2_mu"], data[p + "m2_var"] = m2.means_, m2._covars_
        # Viterbi through utils.viterbi called directly with log inputs
        with np.errstate(divide="ignore"):
            lpi, lA = np.log(pi), np.log(A)
        vlp, vpath = [], []
        for x in subs:
            ll = m._log_multivariate_normal_density_diag(x, m.means_, m._covars_)
            a, b = ref_utils.viterbi(lpi, lA, ll)
            vlp.append(a)
            vpath.append(b)
        data[p + "vit_logprob"] = np.array(vlp)
        data[p + "vit_path"] = np.concatenate(vpath).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "hmm_golden.npz"), **data)
    print("hmm_golden.npz:", [c[0] for c in cases])


def make_posteriors(ref_hmm):
    """score_scaling(compute_posteriors=True) of every hmm_golden case (its committed
    inputs are re-used)."""
    src = np.load(os.path.join(OUT, "hmm_golden.npz"))
    data = {}
    for name in src["cases"]:
        p = str(name) + "_"
        m = _ref_model(ref_hmm, src[p + "pi"], src[p + "A"], src[p + "mu"], src[p + "var"])
        lp, post = m.score_scaling(src[p + "X"], src[p + "lengths"], compute_posteriors=True)
        data[p + "logprob"], data[p + "posteriors"] = np.float64(lp), post
    # (sample() can not be pinned: the reference's covars_ getter raises a
    # TypeError, hmm.py:104-108, and generate_sample_from_state needs it)
    np.savez_compressed(os.path.join(OUT, "posteriors_golden.npz"), **data)
    print("posteriors_golden.npz:", sorted(data))


def _endpoint_chunks():
    """Synthetic 1 s chunks (quiet background, loud tone bursts, hiss around some of the
    bursts) plus 1 s chunks of two of the reference's raw recordings."""
    from scipy.io import wavfile
    rng = np.random.default_rng(515)
    out = []

    def chunk(fs, parts, seconds=1.0, floor=0.0005):
        n = int(seconds * fs)
        t = np.arange(n) / fs
        x = floor * rng.standard_normal(n)
        for kind, a, b, amp in parts:
            i, j = int(a * fs), int(b * fs)
            if kind == "tone":
                x[i:j] += amp * np.sin(2 * np.pi * 180.0 * t[i:j]) * (1 + 0.3 * np.sin(2 * np.pi * 7 * t[i:j]))
            else:   # hiss: many zero crossings, little energy
                x[i:j] += amp * rng.standard_normal(j - i)
        return x.astype(np.float32), fs

    out.append(chunk(22050, [("tone", 0.35, 0.65, 0.3)]))
    out.append(chunk(22050, [("tone", 0.15, 0.35, 0.3), ("tone", 0.6, 0.85, 0.25)]))
    out.append(chunk(22050, [("tone", 0.35, 0.65, 0.3)], floor=0.004))       # loud background
    out.append(chunk(22050, [("tone", 0.0, 0.2, 0.3)]))                       # starts in block 0
    out.append(chunk(22050, [("tone", 0.7, 1.0, 0.3)]))                       # reaches the end
    out.append(chunk(22050, []))                                              # background only
    out.append(chunk(22050, [("tone", 0.5, 0.54, 0.3)]))                      # too short a burst
    out.append(chunk(22050, [("hiss", 0.22, 0.4, 0.03), ("tone", 0.4, 0.6, 0.3), ("hiss", 0.6, 0.8, 0.03)]))
    out.append(chunk(16000, [("tone", 0.3, 0.5, 0.3), ("hiss", 0.5, 0.62, 0.04)]))
    out.append(chunk(16000, [("tone", 0.2, 0.5, 0.3), ("tone", 1.2, 1.9, 0.2), ("hiss", 1.9, 2.1, 0.03)], seconds=2.5))
    out.append(chunk(22050, [("tone", 0.0, 0.03, 0.3), ("tone", 0.4, 0.7, 0.3)]))   # short run in block 0
    for name in ("play_music_12", "what_time_is_it_27"):
        fs, x = wavfile.read(f"{REF}/Recordings_3005/{name}.wav")
        x = np.asarray(x, dtype=np.float32)
        for i in range(0, len(x), fs):
            out.append((x[i:i + fs], fs))
    return out


def make_endpoint():
    """AudioMonitor.start_saved_audio + combine_audio of new_speech_detect.py on each
    chunk.  The module imports audio / GUI / hmmlearn packages that are absent here and
    that this code path never calls: empty placeholder modules stand in for the
    import."""
    for name in ("pyautogui", "hmmlearn", "hmmlearn.hmm", "soundfile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["hmmlearn"].hmm = sys.modules["hmmlearn.hmm"]
    import new_speech_detect as nsd
    data = {}
    chunks = _endpoint_chunks()
    for i, (x, fs) in enumerate(chunks):
        mon = nsd.AudioMonitor(device=0, fs=fs)
        x64 = x.astype(np.float64)
        th = mon.start_saved_audio(x64)
        data[f"x{i}"], data[f"fs{i}"] = x, np.int64(fs)
        data[f"energy{i}"] = np.array(mon.energies, dtype=np.float64)
        data[f"zcr{i}"] = np.array(mon.zcrs, dtype=np.float64)
        speech = len(th) > 0 and th[0][0][-1] > 0
        segs = np.array(th[0], dtype=np.int64).reshape(-1, 4) if speech else np.zeros((0, 4), np.int64)
        data[f"segs{i}"] = segs
        data[f"thr{i}"] = np.array(th[1:4], dtype=np.float64) if speech else np.zeros(3)
        cut = mon.combine_audio(x64, th[0], fs) if speech else np.zeros(0)
        data[f"cut{i}"] = np.asarray(cut, dtype=np.float32)
        print(i, fs, len(x), "segments", segs.tolist(), "thr", data[f"thr{i}"], "cut", len(data[f"cut{i}"]))
    data["n"] = np.int64(len(chunks))
    np.savez_compressed(os.path.join(OUT, "endpoint_golden.npz"), **data)
    print("endpoint_golden.npz:", len(chunks), "chunks")


def fold_init(X, S):
    """Deterministic initial parameters for the fold fixtures (the reference's
    own init draws from unseeded global RNG state, hmm.py:120-127)."""
    pi = np.full(S, 1.0 / S)
    A = np.full((S, S), 0.2 / (S - 1))
    np.fill_diagonal(A, 0.8)
    idx = np.linspace(0, len(X) - 1, S + 2).astype(int)[1:-1]
    mu = X[idx].copy()
    var = np.tile(np.var(X, axis=0) + 1e-3, (S, 1))
    return pi, A, mu, var


def make_folds(ref_final, ref_hmm, n_iter=8):
    import contextlib
    import io
    names = sorted(f[:-4] for f in os.listdir(f"{REF}/cleaned_audios_3005")
                   if f.endswith(".wav") and not f.startswith("audio_"))
    feats, lengths = [], []
    for nme in names:
        with np.errstate(all="ignore"):
            z = ref_final.extract_mfcc_calc(f"{REF}/cleaned_audios_3005/{nme}.wav", 22050)
        feats.append(z.astype(np.float32))
        lengths.append(len(z))
    offs = np.concatenate([[0], np.cumsum(lengths)])
    index = {n: i for i, n in enumerate(names)}
    data = {"names": np.array(names), "feats": np.vstack(feats),
            "lengths": np.array(lengths), "words": np.array(WORDS),
            "n_iter": np.int64(n_iter)}

    def utt(n):
        i = index[n]
        return data["feats"][offs[i]:offs[i + 1]].astype(np.float64)

    def word_of(n):
        return n[: n.rfind("_")]

    for fold in range(1, 6):
        with open(f"{REF}/Folds/train {fold}.txt") as f:
            train = [l.strip()[:-4] for l in f if l.strip()]
        with open(f"{REF}/Folds/validation {fold}.txt") as f:
            val = [l.strip()[:-4] for l in f if l.strip()]
        data[f"f{fold}_train"] = np.array(train)
        data[f"f{fold}_val"] = np.array(val)
        models = {}
        for w in WORDS:
            xs = [utt(n) for n in train if word_of(n) == w]
            X = np.vstack(xs)
            L = np.array([len(x) for x in xs])
            pi, A, mu, var = fold_init(X, 3)
            m = _ref_model(ref_hmm, pi, A, mu, var)
            hist = []
            for _ in range(n_iter):
                st, lp = m.do_estep(X, L)
                m.do_mstep(st)
                hist.append(lp)
            models[w] = m
            p = f"f{fold}_{w}_"
            data[p + "pi"], data[p + "A"] = m.startprob_, m.transmat_
            data[p + "mu"], data[p + "var"] = m.means_, m._covars_
            data[p + "hist"] = np.array(hist)
        scores = np.array([[models[w].score(utt(n)) for w in WORDS] for n in val])
        pred = [max({w: s for w, s in zip(WORDS, row)}, key=dict(zip(WORDS, row)).get)
                for row in scores]
        right = sum(p == word_of(n) for p, n in zip(pred, val))
        data[f"f{fold}_scores"] = scores
        data[f"f{fold}_pred"] = np.array(pred)
        data[f"f{fold}_right"] = np.int64(right)
        print(f"fold {fold}: right {right}/{len(val)}", flush=True)
    np.savez_compressed(os.path.join(OUT, "folds_golden.npz"), **data)
    print("folds_golden.npz written")


HARNESS_SEED = 11          # random_state of every model train_hmm builds in make_harness
HARNESS_GLOBAL_SEED = 77   # numpy's global generator before train_hmm (Dirichlet draws)


def make_harness(ref_mfcc, ref_final, ref_hmm):
    """The HMM_final.py call surface end to end on the 24 fold-1 validation recordings,
    stored as float32 WAVs in a scratch directory the reference's functions are run in:

    * ``calculate_mfcc_for_onerecording`` / ``extract_mfcc_calc`` of every file;
    * ``recognize`` of every file and ``test_trained_model`` (its printed text) with the
      fold-1 models of folds_golden.npz, over six lists drawn from the 24 names;
    * ``train_hmm`` on the 24 files (4 per word).  Its models are built with
      ``random_state=None``, which re-seeds numpy from the clock (hmm.py:125); to get a
      run that can be repeated the class it instantiates is swapped for a subclass that
      differs only in passing ``random_state=HARNESS_SEED``, and numpy's global generator
      is seeded with HARNESS_GLOBAL_SEED first (the Dirichlet draws of hmm.py:120-124 come
      from it).  Everything else is the reference's own code: init, the EM loop of
      ``fit`` with ``do_em_mstep`` only, the monitor's stopping rule, the pickling."""
    import contextlib
    import io
    from scipy.io import wavfile
    folds = np.load(os.path.join(OUT, "folds_golden.npz"))
    names = [str(n) for n in folds["f1_val"]]
    sr = 22050
    data = {"names": np.array(names), "sr": np.int64(sr), "words": np.array(WORDS),
            "seed": np.int64(HARNESS_SEED), "global_seed": np.int64(HARNESS_GLOBAL_SEED)}
    rng = np.random.default_rng(99)
    lists = [names, names] + [[names[i] for i in rng.permutation(24)[:n]] for n in (24, 17, 12, 20)]
    list_files = ["validation 1.txt", "validation 1.txt", "validation 2.txt", "validation 3.txt",
                  "validation 4.txt", "validation 5.txt"]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "wav"))
        os.makedirs(os.path.join(tmp, "Folds"))
        os.makedirs(os.path.join(tmp, "SavedModels"))
        for fname, lst in zip(list_files[1:], lists[1:]):
            with open(os.path.join(tmp, "Folds", fname), "w") as f:
                f.write("".join(n + ".wav\n" for n in lst))
        for k, lst in enumerate(lists[1:]):
            data[f"list{k + 1}"] = np.array(lst)
        os.chdir(tmp)
        try:
            sigs, feats, zs = [], [], []
            for n in names:
                s = wavfile.read(f"{REF}/cleaned_audios_3005/{n}.wav")[1]
                s32 = np.asarray(s, dtype=np.float32)
                wavfile.write(f"wav/{n}.wav", sr, s32)
                sigs.append(s32)
                feats.append(ref_mfcc.calculate_mfcc_for_onerecording(False, f"wav/{n}.wav", sample_rate=sr))
                zs.append(ref_final.extract_mfcc_calc(f"wav/{n}.wav", sr))
            data["sig_lengths"] = np.array([len(x) for x in sigs])
            data["signals"] = np.concatenate(sigs)
            data["lengths"] = np.array([len(z) for z in zs])
            data["feat"] = np.concatenate([f.T for f in feats])      # (sum T, 26), rows = frames
            data["z"] = np.concatenate(zs)                           # (sum T, 26)
            models = {}
            for w in WORDS:
                p = f"f1_{w}_"
                models[w] = _ref_model(ref_hmm, folds[p + "pi"], folds[p + "A"], folds[p + "mu"], folds[p + "var"])
            data["labels"] = np.array([ref_final.recognize(models, f"wav/{n}.wav", sr) for n in names])
            data["scores"] = np.array([[models[w].score(z) for w in WORDS] for z in zs])
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ref_final.test_trained_model(models, "wav", sr)
            data["test_trained_model_stdout"] = np.array(buf.getvalue())
            print(buf.getvalue().count("right:"), "lists;", [l for l in buf.getvalue().splitlines() if l.startswith("right")])
            # seeded train_hmm
            plain = ref_hmm.MyGaussianHMM
            seeded = type("SeededHMM", (plain,), {
                "__init__": lambda self, **kw: plain.__init__(
                    self, **dict(kw, random_state=HARNESS_SEED, verbose=False)),
                "__module__": __name__, "__qualname__": "SeededHMM"})
            globals()["SeededHMM"] = seeded
            ref_hmm.MyGaussianHMM = seeded
            try:
                np.random.seed(HARNESS_GLOBAL_SEED)
                trained = {}
                files = [f"wav/{n}.wav" for n in names]
                labels = [n[: n.rfind("_")] for n in names]
                with contextlib.redirect_stdout(io.StringIO()) as tbuf:
                    ref_final.train_hmm(trained, files, labels, sr)
            finally:
                ref_hmm.MyGaussianHMM = plain
            data["train_stdout"] = np.array(tbuf.getvalue())
            data["train_files"] = np.array(sorted(os.listdir("SavedModels")))
            data["train_order"] = np.array(list(trained))
            for w, m in trained.items():
                p = f"train_{w}_"
                data[p + "pi"], data[p + "A"] = m.startprob_, m.transmat_
                data[p + "mu"], data[p + "var"] = m.means_, m._covars_
                data[p + "hist"] = np.array(m.monitor_.history)
                data[p + "iter"] = np.int64(m.monitor_.iter)
                data[p + "score"] = np.array([m.score(z) for z in zs])
                print(w, "iterations", m.monitor_.iter, "logprob", m.monitor_.history[-1])
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "harness_golden.npz"), **data)
    print("harness_golden.npz written")


if __name__ == "__main__":
    what = set(sys.argv[1:]) or {"mfcc", "hmm", "folds", "post", "endpoint", "harness"}
    ref_mfcc, ref_final, ref_hmm, ref_utils = _import_reference()
    if "mfcc" in what:
        make_mfcc(ref_mfcc, ref_final)
    if "hmm" in what:
        make_hmm(ref_hmm, ref_utils)
    if "post" in what:
        make_posteriors(ref_hmm)
    if "endpoint" in what:
        make_endpoint()
    if "folds" in what:
        make_folds(ref_final, ref_hmm)
    if "harness" in what:
        make_harness(ref_mfcc, ref_final, ref_hmm)
