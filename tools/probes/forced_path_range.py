"""Range of the recursions on a forced path: a strict left-to-right chain of S states whose data
sits on the last state (every other state costs about e^-60 per frame), S = 2 .. 20, both dtypes,
every E-step path, scoring, posteriors and Viterbi against the float64 oracle.

    python tools/probes/forced_path_range.py
"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import speech_recognition_amd as sra
from oracle import hmm_oracle as ho
ctx = sra.Context.default(0)
D = 13
for S in (2, 6, 8, 12, 16, 20):
    rng = np.random.default_rng(S)
    A = np.zeros((S, S))
    for i in range(S):
        A[i, i] = 0.6 if i + 1 < S else 1.0
        if i + 1 < S: A[i, i + 1] = 0.4
    pi = np.zeros(S); pi[0] = 1.0
    means = np.zeros((S, D))
    for i in range(S - 1): means[i] = (3.0 + 0.02 * i) * (1 if i % 2 == 0 else -1)
    prm = ho.Params(pi, A, means, np.ones((S, D)))
    lengths = np.array([130, 65, 200, 40, 64])
    X = 0.3 * rng.standard_normal((int(lengths.sum()), D))
    bank = sra.HMMBank(ctx, pi[None], A[None], means[None], np.ones((1, S, D)))
    st, lps = ho.estep(prm, X, [int(t) for t in lengths], loops=False)
    _, refpost = ho.score_posteriors(prm, X, [int(t) for t in lengths], loops=False)
    vlp_ref, vpath_ref = ho.viterbi(prm, X, [int(t) for t in lengths])
    for dtype in ("float32", "float64"):
        data = sra.Ragged(ctx, X, lengths, dtype=dtype)
        msgs = []
        for path in ("auto", "lane", "fused", "generic"):
            ctx.set_estep_path("auto" if path == "generic" else path)
            try:
                g = bank.unpack_stats(bank.estep(data, generic=(path == "generic")))[0]
            finally:
                ctx.set_estep_path("auto")
            err = max(float(np.abs(g[k] - st[k]).max() / max(1.0, np.abs(st[k]).max())) for k in ("start", "trans", "post", "obs", "obs**2"))
            msgs.append(f"{path}:{err:.1e}")
        sc = bank.score(data).cpu().numpy()[:, 0]
        plp, gam = bank.posteriors(data)
        gam = gam.cpu().numpy()
        vlp, vp = bank.viterbi(data)
        vp = vp.cpu().numpy()
        print(f"S={S:2d} {dtype} estep", " ".join(msgs), f"score {np.abs(sc - lps).max() / np.abs(lps).max():.1e}",
              f"post {np.abs(gam - refpost).max():.1e} finite {bool(np.isfinite(gam).all())}",
              f"viterbi paths differ in {int((vp != np.asarray(vpath_ref).ravel()).sum())} frames, lp {np.abs(vlp.cpu().numpy() - vlp_ref).max():.1e}")
