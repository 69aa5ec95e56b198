"""Random model / batch shapes through the float32 fast paths against the numpy oracle:
python tools/fuzz_shapes.py [n_cases] [seed] [float32|float64] [max utterances].  Prints the first failure
and exits 1.  float64 runs the f64 matrix-core scoring kernel (up to 16 states, 48 features) and the
shape-generic kernels (states up to 40) with tolerances a million times tighter."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
from oracle import hmm_oracle as ho
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import assert_ties_only

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dtype = sys.argv[3] if len(sys.argv) > 3 else "float32"
max_u = int(sys.argv[4]) if len(sys.argv) > 4 else 40      # more than 32 utterances: visiting orders are made
tight = 1e-6 if dtype == "float64" else 1.0
ctx = sra.Context.default(0)
if os.environ.get("SRX_VITERBI_PATH"):          # "fused" / "split": one of the two float32 Viterbi paths
    ctx.set_viterbi_path(os.environ["SRX_VITERBI_PATH"])
rng = np.random.default_rng(seed)
worst = {"score": 0.0, "post": 0.0, "stats": 0.0}
for case in range(n_cases):
    M = int(rng.integers(1, 9)) if rng.random() < 0.8 else int(rng.integers(9, 40))
    S = int(rng.integers(1, 17 if dtype == "float32" else 41))
    D = int(rng.integers(1, 49 if dtype == "float32" else 70))
    U = int(rng.integers(1, max_u))
    lengths = rng.integers(0, 60, size=U)
    lengths[rng.integers(0, U)] = int(rng.integers(1, 300))
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    X = rng.standard_normal((int(lengths.sum()), D))
    um = rng.integers(0, M, size=U).astype(np.int32)
    bank = sra.HMMBank(ctx, np.stack([p.pi for p in prms]), np.stack([p.A for p in prms]),
                       np.stack([p.means for p in prms]), np.stack([p.covars for p in prms]))
    data = sra.Ragged(ctx, X, lengths, dtype=dtype)
    tag = f"case {case}: M={M} S={S} D={D} U={U} lengths={lengths.tolist()}"
    try:
        sc = bank.score(data).cpu().numpy()
        lp_v, path = bank.viterbi(data, um)
        lp_p, gam = bank.posteriors(data, um)
        st = bank.unpack_stats(bank.estep(data, ctx.tensor(um), None))
        lp_v, path, lp_p, gam = lp_v.cpu().numpy(), path.cpu().numpy(), lp_p.cpu().numpy(), gam.cpu().numpy()
        off = np.concatenate([[0], np.cumsum(lengths)])
        for u in range(U):
            n = int(lengths[u])
            if n == 0:
                assert np.all(sc[u] == 0) and lp_p[u] == 0 and lp_v[u] == 0, "empty utterance"
                continue
            x = X[off[u]:off[u + 1]]
            ref = np.array([ho.score(p, x, [n], loops=False)[0] for p in prms])
            e = np.max(np.abs(sc[u] - ref) / np.abs(ref))
            worst["score"] = max(worst["score"], e)
            assert e < 1e-4 * tight, f"score rel err {e}"
            rl, rg = ho.score_posteriors(prms[um[u]], x, [n], loops=False)
            e = np.abs(gam[off[u]:off[u + 1]] - rg).max()
            worst["post"] = max(worst["post"], e)
            assert e < 5e-5 * tight, f"posterior abs err {e}"
            assert abs(lp_p[u] - rl[0]) < 1e-4 * tight * abs(rl[0]), "posterior logprob"
            vlp, vpath = ho.viterbi(prms[um[u]], x, [n])
            assert abs(lp_v[u] - vlp[0]) < 1e-4 * tight * abs(vlp[0]), "viterbi logprob"
            if dtype == "float64":
                assert np.array_equal(path[off[u]:off[u + 1]], vpath), "viterbi path"
            elif not np.array_equal(path[off[u]:off[u + 1]], vpath):
                # float32: the paths may part at near-ties of the float64 lattice only, and the
                # float32 path must reach the optimum under the float64 model
                pr = prms[um[u]]
                assert_ties_only(pr.pi, pr.A, ho.log_density_diag(x, pr.means, pr.covars), [n], path[off[u]:off[u + 1]], vpath,
                                 vlp[0], eps=2e-5)
        for m in range(M):
            sel = [u for u in range(U) if um[u] == m and lengths[u] > 0]
            if not sel:
                assert st[m]["nobs"] == 0
                continue
            Xm = np.vstack([X[off[u]:off[u + 1]] for u in sel])
            ref, lps = ho.estep(prms[m], Xm, [int(lengths[u]) for u in sel], loops=False)
            assert st[m]["nobs"] == len(sel)
            for k in ("start", "trans", "post", "obs", "obs**2"):
                scale = max(1.0, np.abs(ref[k]).max())
                e = np.abs(st[m][k] - ref[k]).max() / scale
                worst["stats"] = max(worst["stats"], e)
                assert e < 5e-5 * tight, f"stats {k} err {e}"
    except AssertionError as ex:
        print("FAIL", tag, "->", ex)
        sys.exit(1)
print(f"{n_cases} cases ok; worst score rel {worst['score']:.2e}, posterior abs {worst['post']:.2e}, stats {worst['stats']:.2e}")
