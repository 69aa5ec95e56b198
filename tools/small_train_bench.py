"""Per-iteration time of batched Baum-Welch at the reference's own size (6 word models x 3
states x 26 dims, 150 utterances of about 150 frames): launch-bound, not kernel-bound."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
ctx = sra.Context.default(0)
rng = np.random.default_rng(1)
M, S, D, U = 6, 3, 26, 150
lengths = rng.integers(60, 240, size=U)
X = rng.standard_normal((int(lengths.sum()), D))
um = (np.arange(U) % M).astype(np.int32)
for dtype in ("float32", "float64"):
    bank = sra.HMMBank(ctx, rng.dirichlet(np.ones(S), size=M), rng.dirichlet(np.ones(S), size=(M, S)),
                       rng.standard_normal((M, S, D)), rng.uniform(0.5, 1.5, (M, S, D)))
    data = sra.Ragged(ctx, X, lengths, dtype=dtype)
    tr = sra.parallel.BaumWelchTrainer(bank, data, um)
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        lp = tr.step()
        float(lp.sum().item())          # the convergence check of fit()
    dts = (time.perf_counter() - t0) / n
    print(f"{dtype}: {dt*1e6:.0f} us per EM iteration ({dts*1e6:.0f} us with a host read of the log-probability each)")
