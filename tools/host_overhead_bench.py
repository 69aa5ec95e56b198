"""Host time per call (enqueue only, no synchronisation) of the entry points on a tiny
batch: what a caller that handles one utterance at a time pays before any kernel runs."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
ctx = sra.Context.default(0)
rng = np.random.default_rng(1)
M, S, D, U = 6, 3, 26, 8
lengths = rng.integers(60, 240, size=U)
X = rng.standard_normal((int(lengths.sum()), D))
um = ctx.tensor((np.arange(U) % M).astype(np.int32))
bank = sra.HMMBank(ctx, rng.dirichlet(np.ones(S), size=M), rng.dirichlet(np.ones(S), size=(M, S)),
                   rng.standard_normal((M, S, D)), rng.uniform(0.5, 1.5, (M, S, D)))
audio = (0.1 * rng.standard_normal(22050)).astype(np.float32)
plan = sra.MfccPlan(ctx, 22050, nfft=1024, n_mel=26, delta_order=1, zscore=True)
a_dev = ctx.tensor(audio)
for dtype in ("float32", "float64"):
    data = sra.Ragged(ctx, X, lengths, dtype=dtype)
    stats = bank.estep(data, um)
    calls = {
        "score": lambda: bank.score(data),
        "estep": lambda: bank.estep(data, um, None, out=stats),
        "viterbi": lambda: bank.viterbi(data, um),
        "posteriors": lambda: bank.posteriors(data, um),
        "mfcc": lambda: plan.run(a_dev.double() if dtype == "float64" else a_dev, [0, len(audio)], dtype=dtype),
    }
    for name, fn in calls.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{dtype:8s} {name:10s} host {1e6*(t1-t0)/n:7.0f} us/call   with completion {1e6*(t2-t0)/n:7.0f} us/call")
