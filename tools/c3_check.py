"""configs[2] size on one GPU: 1,048,576 utterances x 200 frames x 39 dims (31 GB of
features).  Times a Baum-Welch iteration and checks that addressing beyond 4 GiB is
sound: scores and statistics of slices taken from the far end of the buffer equal
those of the same slices processed on their own."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
U = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T, D, M, S = 200, 39, 6, 16
ctx = sra.Context.default(0)
g = torch.Generator(device='cuda'); g.manual_seed(1)
X = torch.empty((U * T, D), device='cuda', dtype=torch.float32)
step = 1 << 24
for i in range(0, U * T, step):
    X[i:i + step].normal_(generator=g)
off = torch.arange(0, (U + 1) * T, T, device='cuda', dtype=torch.int64)
data = sra.Ragged(ctx, X, offsets=off, dtype="float32")
rng = np.random.default_rng(8)
bank = sra.HMMBank(ctx, rng.dirichlet(np.ones(S), size=M), rng.dirichlet(np.ones(S), size=(M, S)),
                   rng.standard_normal((M, S, D)), rng.uniform(0.5, 1.5, (M, S, D)))
um = (np.arange(U) % M).astype(np.int32)
umd = ctx.tensor(um)
order = ctx.tensor(np.argsort(um, kind="stable").astype(np.int64))
stats = bank.estep(data, umd, order).clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    bank.estep(data, umd, order, out=stats)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"E-step U={U}: {dt*1e3:.2f} ms  {U*T/dt/1e9:.2f} Gframes/s  workspace {ctx.workspace_bytes()/2**30:.1f} GiB")
st = bank.unpack_stats(stats)
for m in range(M):
    n_fr = float((um == m).sum()) * T
    assert st[m]["nobs"] == (um == m).sum()
    assert abs(st[m]["post"].sum() - n_fr) < 2e-5 * n_fr, (m, st[m]["post"].sum(), n_fr)
    assert abs(st[m]["trans"].sum() - (n_fr - (um == m).sum())) < 2e-5 * n_fr
# the far end on its own: last 4096 utterances (beyond byte 4 GiB of X when U is large)
n = 4096
tail = sra.Ragged(ctx, X[(U - n) * T:], offsets=off[:n + 1], dtype="float32")
bank8 = sra.HMMBank(ctx, rng.dirichlet(np.ones(8), size=M), rng.dirichlet(np.ones(8), size=(M, 8)),
                    rng.standard_normal((M, 8, D)), rng.uniform(0.5, 1.5, (M, 8, D)))
sc_all = bank8.score(data)
torch.cuda.synchronize()
t0 = time.perf_counter()
sc_all = bank8.score(data)
torch.cuda.synchronize()
print(f"scoring U={U}: {(time.perf_counter()-t0)*1e3:.2f} ms")
sc_tail = bank8.score(tail)
assert torch.equal(sc_all[U - n:], sc_tail), "scores of the far end differ"
# E-step of the far end: statistics of (all) minus (all but the tail) equal the tail's
head = sra.Ragged(ctx, X[:(U - n) * T], offsets=off[:U - n + 1], dtype="float32")
s_head = bank.estep(head, ctx.tensor(um[:U - n]), None).cpu().numpy()
s_tail = bank.estep(tail, ctx.tensor(um[U - n:]), None).cpu().numpy()
s_all = bank.estep(data, umd, None).cpu().numpy()
err = np.abs(s_all - (s_head + s_tail)).max() / np.abs(s_all).max()
print("E-step additivity over the 4 GiB boundary: max rel err", err)
assert err < 1e-6
print("ok")
