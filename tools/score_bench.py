"""Time the scoring kernel alone: python tools/score_bench.py [M] [S] [U] [T] [D]."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
M = int(sys.argv[1]) if len(sys.argv) > 1 else 6
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
U = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
T = int(sys.argv[4]) if len(sys.argv) > 4 else 200
D = int(sys.argv[5]) if len(sys.argv) > 5 else 39
ctx = sra.Context.default(0)
X = torch.randn((U * T, D), device='cuda', dtype=torch.float32)
off = torch.arange(0, (U + 1) * T, T, device='cuda', dtype=torch.int64)
data = sra.Ragged(ctx, X, offsets=off, dtype="float32")
rng = np.random.default_rng(8)
bank = sra.HMMBank(ctx, rng.dirichlet(np.ones(S), size=M), rng.dirichlet(np.ones(S), size=(M, S)),
                   rng.standard_normal((M, S, D)), rng.uniform(0.5, 1.5, (M, S, D)))
out = bank.score(data)
torch.cuda.synchronize()
n = 20
for rep in range(8):          # the clock settles over the first tens of ms of load: watch it
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n):
        bank.score(data, out=out)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / n
    print(f"  group {rep}: {ms:.4f} ms")
print(f"M={M} S={S} U={U} T={T} D={D}: {ms:.4f} ms  {U*T/ms/1e6:.2f} Gframes/s  {U*T*D*4/ms/1e9:.3f} TB/s X-stream  finite={bool(torch.isfinite(out).all())}")
