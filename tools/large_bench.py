"""Large model (1024 states x 16-component diagonal GMMs x 39 dims): emission
log-densities and dense Viterbi.  python tools/large_bench.py [U] [T] [S] [MX]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
U = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
MX = int(sys.argv[4]) if len(sys.argv) > 4 else 16
D = 39
ctx = sra.Context.default(0)
rng = np.random.default_rng(5)
em = sra.GmmEmission(ctx, rng.dirichlet(np.ones(MX), size=S), rng.standard_normal((S, MX, D)),
                     rng.uniform(0.5, 1.5, (S, MX, D)))
X = torch.randn((U * T, D), device='cuda', dtype=torch.float32)
off = torch.arange(0, (U + 1) * T, T, device='cuda', dtype=torch.int64)
data = sra.Ragged(ctx, X, offsets=off, dtype="float32")
ll = em.loglik(data)
torch.cuda.synchronize()
from _warm import warm
warm(lambda: em.loglik(data, out=ll))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
n = 3
ev[0].record()
for _ in range(n):
    em.loglik(data, out=ll)
ev[1].record()
torch.cuda.synchronize()
ms_e = ev[0].elapsed_time(ev[1]) / n
flop = 2.0 * 2 * D * S * MX * U * T
print(f"emission S={S} MX={MX}: {ms_e:.3f} ms  {U*T/ms_e/1e3:.3f} Mframes/s  {flop/ms_e/1e9:.1f} TFLOP/s (algorithmic 2*78*S*MX per frame)")
# spot check against numpy on 8 frames x 16 states
idx = torch.randint(0, U * T, (8,), device='cuda')
w, mu, cv = em.weights.cpu().numpy(), em.means.cpu().numpy(), em.covars.cpu().numpy()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hmm_oracle as ho
ref = ho.gmm_log_density(X[idx].double().cpu().numpy(), w[:16], mu[:16], cv[:16])
print("emission max rel err", float(np.max(np.abs(ll[idx][:, :16].cpu().numpy() - ref) / np.abs(ref))))
A = rng.dirichlet(np.ones(S), size=S)
pi = rng.dirichlet(np.ones(S))
spd, Ad = ctx.tensor(pi), ctx.tensor(A)
lp, path = sra.viterbi_dense(ctx, ll, off, spd, Ad)
torch.cuda.synchronize()
ev[2].record()
for _ in range(n):
    lp, path = sra.viterbi_dense(ctx, ll, off, spd, Ad)
ev[3].record()
torch.cuda.synchronize()
ms_v = ev[2].elapsed_time(ev[3]) / n
print(f"viterbi dense S={S}: {ms_v:.3f} ms  {U*T/ms_v/1e3:.3f} Mframes/s  {U*T*float(S)*S/ms_v/1e9:.2f} T(max,+)/s")
r_lp, r_path = ho.viterbi_dense(pi, A, ll[:T].double().cpu().numpy(), [T])
print("viterbi utt0: path agreement", float(np.mean(path[:T].cpu().numpy() == r_path)), "lp rel err", abs(float(lp[0]) - r_lp[0]) / abs(r_lp[0]))
