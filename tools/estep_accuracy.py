"""Float32 E-step statistics against the float64 ones on the same seeded batch: largest
relative difference per kind of statistic (relative to the largest entry of its row).
python tools/estep_accuracy.py [U] [S] [peaked|extreme]   (peaked: narrow variances, far-apart
means -- posteriors close to 0 and 1, scaled forward vectors with very small entries; extreme:
more of both, and sparse transition rows)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
ctx = sra.Context.default(0)
U = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
extreme = len(sys.argv) > 3 and sys.argv[3] == "extreme"
peaked = extreme or (len(sys.argv) > 3 and sys.argv[3] == "peaked")
D, M = 39, 6
rng = np.random.default_rng(11)
lengths = rng.integers(1, 340, size=U)
mu = rng.standard_normal((M, S, D)) * (8.0 if extreme else 3.0 if peaked else 1.0)
var = rng.uniform(0.01, 0.05, (M, S, D)) if extreme else rng.uniform(0.05, 0.2, (M, S, D)) if peaked \
    else rng.uniform(0.5, 1.5, (M, S, D))
A = rng.dirichlet(np.ones(S) * (0.05 if extreme else 0.2 if peaked else 1.0), size=(M, S))
um_host = (np.arange(U) % M).astype(np.int32)
# every utterance is a sample of its own model (a state path drawn from pi and A), so that
# its likelihood is not vanishing however sparse the transition rows are
pi = rng.dirichlet(np.ones(S), size=M)
cumA = np.cumsum(A, axis=2)
paths = np.zeros((U, int(lengths.max())), dtype=np.int64)
cur = np.array([rng.choice(S, p=pi[m]) for m in um_host])
for t in range(paths.shape[1]):
    paths[:, t] = cur
    r = rng.random(U)
    cur = np.minimum((r[:, None] > cumA[um_host, cur]).sum(axis=1), S - 1)
mask = np.arange(paths.shape[1])[None, :] < lengths[:, None]
st_of = paths[mask]
mod_of = np.repeat(um_host, lengths)
X = mu[mod_of, st_of] + rng.standard_normal((len(st_of), D)) * np.sqrt(var[mod_of, st_of])
bank = sra.HMMBank(ctx, pi, A, mu, var)
um = ctx.tensor(um_host)
s32 = bank.estep(sra.Ragged(ctx, X, lengths, dtype="float32"), um).cpu().numpy()
s64 = bank.estep(sra.Ragged(ctx, X, lengths, dtype="float64"), um).cpu().numpy()
d = np.abs(s32 - s64)
print(f"U={U} S={S} {'extreme' if extreme else 'peaked' if peaked else 'broad'}: finite {bool(np.isfinite(s32).all())}/{bool(np.isfinite(s64).all())} max |f32 - f64| / max |f64| per model row: "
      f"{(d.max(axis=1) / np.abs(s64).max(axis=1)).max():.3e};  "
      f"element-wise over entries above 1e-3 of their row's maximum: "
      f"{(d / np.maximum(np.abs(s64), 1e-300))[np.abs(s64) > 1e-3 * np.abs(s64).max(axis=1, keepdims=True)].max():.3e}")
