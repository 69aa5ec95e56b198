"""Time of the fused E-step alone (float32, 131072 utterances x 200 frames x 39 features, 6 models
x 16 states) after the clock has ramped up: best and median of 5 groups of 4 calls."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
ctx = sra.Context.default(0)
U, T, D, M, S = int(os.environ.get("ES_U", 131072)), 200, 39, 6, 16
# ES_RAGGED=1: utterances of 60 to 340 frames (mean 200) instead of 200 each
# ES_ORDER=memory: the caller's stable sort by model instead of the library's own order
if os.environ.get("ES_RAGGED"):
    lengths = np.random.default_rng(3).integers(60, 341, size=U)
    off = torch.as_tensor(np.concatenate([[0], np.cumsum(lengths)]), dtype=torch.int64, device='cuda')
    X = torch.randn((int(lengths.sum()), D), device='cuda', dtype=torch.float32)
else:
    X = torch.randn((U * T, D), device='cuda', dtype=torch.float32)
    off = torch.arange(0, (U + 1) * T, T, device='cuda', dtype=torch.int64)
data = sra.Ragged(ctx, X, offsets=off, dtype="float32")
rng = np.random.default_rng(8)
bank = sra.HMMBank(ctx, rng.dirichlet(np.ones(S), size=M), rng.dirichlet(np.ones(S), size=(M, S)),
                   rng.standard_normal((M, S, D)), rng.uniform(0.5, 1.5, (M, S, D)))
# ES_UM=block: each model's utterances contiguous in memory; default: round robin over the models
um_host = np.arange(U) * M // U if os.environ.get("ES_UM") == "block" else np.arange(U) % M
um = torch.as_tensor(um_host, dtype=torch.int32, device='cuda')
order = None
if os.environ.get("ES_ORDER") == "memory":
    order = torch.as_tensor(np.argsort(um_host, kind="stable").astype(np.int64), device='cuda')
which = os.environ.get("ES_STATS", "all")   # "transitions": the E-step of a transitions-only M-step
out = None
for _ in range(6):
    out = bank.estep(data, um, order, out=out, stats=which)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        out = bank.estep(data, um, order, out=out, stats=which)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 4)
st = out.cpu().numpy()
print(f"estep ms/call best {min(ts):.3f} median {sorted(ts)[2]:.3f}  frames/s {X.shape[0]/min(ts)*1e3:.3e}  "
      f"checksum {float(np.abs(st).sum()):.10e} finite {bool(np.isfinite(st).all())}")
