"""Time batched Viterbi decoding: python tools/viterbi_bench.py [S] [U] [T] [dtype]."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
U = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dtype = sys.argv[4] if len(sys.argv) > 4 else "float32"
D, M = int(os.environ.get("VB_D", "39")), 6
ctx = sra.Context.default(0)
if os.environ.get("SRX_VITERBI_PATH"):
    ctx.set_viterbi_path(os.environ["SRX_VITERBI_PATH"])
X = torch.randn((U * T, D), device='cuda', dtype=torch.float32)
off = torch.arange(0, (U + 1) * T, T, device='cuda', dtype=torch.int64)
data = sra.Ragged(ctx, X.double() if dtype == "float64" else X, offsets=off, dtype=dtype)
rng = np.random.default_rng(8)
bank = sra.HMMBank(ctx, rng.dirichlet(np.ones(S), size=M), rng.dirichlet(np.ones(S), size=(M, S)),
                   rng.standard_normal((M, S, D)), rng.uniform(0.5, 1.5, (M, S, D)))
um = ctx.tensor((np.arange(U) % M).astype(np.int32))
lp, path = bank.viterbi(data, um)
torch.cuda.synchronize()
from _warm import warm
warm(lambda: bank.viterbi(data, um))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
ev0.record()
for _ in range(n):
    lp, path = bank.viterbi(data, um)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / n
print(f"S={S} U={U} T={T} {dtype}: {ms:.3f} ms  {U*T/ms/1e6:.3f} Gframes/s")
