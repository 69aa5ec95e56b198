"""Float64 emission log-densities of all state columns (srx_emission_loglik):
python tools/emission_f64_bench.py [U]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
U = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T, D, M, S = 200, 39, 6, 8
ctx = sra.Context.default(0)
X = torch.randn((U * T, D), device='cuda', dtype=torch.float64)
data = sra.Ragged(ctx, X, offsets=torch.arange(0, (U + 1) * T, T, device='cuda', dtype=torch.int64), dtype="float64")
rng = np.random.default_rng(8)
bank = sra.HMMBank(ctx, rng.dirichlet(np.ones(S), size=M), rng.dirichlet(np.ones(S), size=(M, S)),
                   rng.standard_normal((M, S, D)), rng.uniform(0.5, 1.5, (M, S, D)))
ll = bank.emission_loglik(data)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(3):
    ll = bank.emission_loglik(data)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 3
# reference values for a few frames, on the host
idx = torch.tensor([0, 1, U * T // 2, U * T - 1], device='cuda')
x = X[idx].cpu().numpy()
_, _, mu, var = bank.params_host()
want = -0.5 * (D * np.log(2 * np.pi) + np.log(var).sum(-1)[None] + (((x[:, None, None, :] - mu[None]) ** 2) / var[None]).sum(-1))
got = ll[idx].cpu().numpy().reshape(4, M, S)
print(f"float64 emission {U*T} frames x {M*S} columns: {ms:.3f} ms ({U*T/ms/1e6:.2f} Gframes/s), max rel err {np.abs(got/want-1).max():.2e}")
