"""Float64 scoring: matrix-core kernel against the shape-generic one.
python tools/score_f64_bench.py [M] [S] [U] [T] [D]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
M = int(sys.argv[1]) if len(sys.argv) > 1 else 6
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
U = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
T = int(sys.argv[4]) if len(sys.argv) > 4 else 200
D = int(sys.argv[5]) if len(sys.argv) > 5 else 39
ctx = sra.Context.default(0)
X = torch.randn((U * T, D), device='cuda', dtype=torch.float64)
off = torch.arange(0, (U + 1) * T, T, device='cuda', dtype=torch.int64)
data = sra.Ragged(ctx, X, offsets=off, dtype="float64")
rng = np.random.default_rng(8)
bank = sra.HMMBank(ctx, rng.dirichlet(np.ones(S), size=M), rng.dirichlet(np.ones(S), size=(M, S)),
                   rng.standard_normal((M, S, D)), rng.uniform(0.5, 1.5, (M, S, D)))
res = {}
for generic in (False, True):
    out = bank.score(data, generic=generic)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 3
    ev0.record()
    for _ in range(n):
        bank.score(data, out=out, generic=generic)
    ev1.record()
    torch.cuda.synchronize()
    res[generic] = (ev0.elapsed_time(ev1) / n, out.clone())
ms_f, o_f = res[False]
ms_g, o_g = res[True]
rel = ((o_f - o_g).abs() / o_g.abs()).max().item()
print(f"M={M} S={S} U={U} T={T} D={D} float64: matrix-core {ms_f:.3f} ms ({U*T/ms_f/1e6:.2f} Gframes/s), "
      f"generic {ms_g:.3f} ms; max relative difference {rel:.2e}")
