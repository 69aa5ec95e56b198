"""Time the end-pointing pre-pass: python tools/endpoint_bench.py [n_rec] [seconds]."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sec = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
fs = 16000
n = int(sec * fs)
ctx = sra.Context.default(0)
audio = 0.001 * torch.randn((R * n,), device='cuda', dtype=torch.float32)
# a loud second in the middle of every recording
audio.view(R, n)[:, n // 3:2 * n // 3] *= 200.0
so = np.arange(0, (R + 1) * n, n, dtype=np.int64)
lay = sra.endpoint.BatchLayout(so, fs, ctx=ctx)      # offsets on the device once
res = sra.endpoint.detect_batch(audio, lay, fs, ctx=ctx)
torch.cuda.synchronize()
from _warm import warm
warm(lambda: sra.endpoint.detect_batch(audio, lay, fs, ctx=ctx))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
k = 5
ev0.record()
for _ in range(k):
    res = sra.endpoint.detect_batch(audio, lay, fs, ctx=ctx)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / k
print(f"R={R} {sec}s: {ms:.3f} ms  {R*n/ms/1e6:.2f} Gsamples/s  hbm frac {R*n*4/ms/1e6/8000:.3f}  segments {int(res.counts.sum())}")
# yardstick: a plain reduction over the same samples
ev0.record()
for _ in range(k):
    s = audio.sum()
ev1.record()
torch.cuda.synchronize()
ms2 = ev0.elapsed_time(ev1) / k
print(f"torch.sum over the same buffer: {ms2:.3f} ms  hbm frac {R*n*4/ms2/1e6/8000:.3f}")
