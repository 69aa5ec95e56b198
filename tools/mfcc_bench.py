import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
ctx = sra.Context.default(0)
n_utt, n_samp = 8192, 48000
audio = 0.2 * torch.randn((n_utt * n_samp,), device='cuda', dtype=torch.float32)
so = np.arange(0, (n_utt + 1) * n_samp, n_samp, dtype=np.int64)
order = int(sys.argv[1]) if len(sys.argv) > 1 else 2
zs = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
plan = sra.MfccPlan(ctx, 16000, nfft=512, n_mel=40, delta_order=order, zscore=zs)
feats, fo = plan.run(audio, so)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    plan.run(audio, so, out=feats)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
frames = n_utt * ((n_samp - 400) // 160 + 1)
print("ms", dt * 1e3, "frames/s", frames / dt, "hbm frac", frames * 796 / dt / 8e12)
