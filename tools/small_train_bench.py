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
    # convergence checking the way MyGaussianHMM.fit does it: the log-probability of iteration k
    # goes to pinned host memory asynchronously and is looked at while iteration k + 1 runs
    logp = torch.empty((n,), dtype=torch.float64).pin_memory()
    ev = [torch.cuda.Event() for _ in range(2)]
    hist = []
    t0 = time.perf_counter()
    for k in range(n):
        lp = tr.step()
        logp[k:k + 1].copy_(lp.sum().reshape(1), non_blocking=True)
        ev[k % 2].record()
        if k >= 1:
            ev[(k - 1) % 2].synchronize()
            hist.append(float(logp[k - 1]))
    torch.cuda.synchronize()
    dtl = (time.perf_counter() - t0) / n
    print(f"{dtype}: {dt*1e6:.0f} us per EM iteration; {dts*1e6:.0f} us with a blocking host read of the "
          f"log-probability each; {dtl*1e6:.0f} us with the check one iteration behind")
# the class itself (one model, float64 and float32), convergence checking on, no early stop
Xs = X[:int(lengths[:25].sum())]
for dtype in ("float32", "float64"):
    m = sra.MyGaussianHMM(n_components=S, n_iter=60, tol=-1e300, verbose=False, dtype=dtype, random_state=0)
    np.random.seed(0)
    m.init(Xs, lengths[:25])
    m.fit_from_current(Xs, lengths[:25])
    t0 = time.perf_counter()
    m.fit_from_current(Xs, lengths[:25])
    dt = (time.perf_counter() - t0) / m.monitor_.iter
    print(f"MyGaussianHMM.fit_from_current {dtype}: {dt*1e6:.0f} us per EM iteration over {m.monitor_.iter} iterations "
          f"(25 utterances, convergence checked every iteration)")
