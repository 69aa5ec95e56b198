"""Float64 E-step (validation dtype): python tools/estep_f64_bench.py [U] [S]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
ctx = sra.Context.default(0)
U = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
T, D, M = 200, 39, 6
X = torch.randn((U * T, D), device='cuda', dtype=torch.float64)
off = torch.arange(0, (U + 1) * T, T, device='cuda', dtype=torch.int64)
data = sra.Ragged(ctx, X, offsets=off, dtype="float64")
rng = np.random.default_rng(8)
bank = sra.HMMBank(ctx, rng.dirichlet(np.ones(S), size=M), rng.dirichlet(np.ones(S), size=(M, S)),
                   rng.standard_normal((M, S, D)), rng.uniform(0.5, 1.5, (M, S, D)))
tr = sra.parallel.BaumWelchTrainer(bank, data, np.arange(U) % M)
tr.step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    lp = tr.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("float64 ms/iter", dt * 1e3, "frames/s", U * T / dt, "logprob0", float(lp[0]))
