import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
ctx = sra.Context.default(0)
n_utt, n_samp = 8192, 48000
audio = 0.2 * torch.randn((n_utt * n_samp,), device='cuda', dtype=torch.float32)
so = np.arange(0, (n_utt + 1) * n_samp, n_samp, dtype=np.int64)
order = int(sys.argv[1]) if len(sys.argv) > 1 else 2
zs = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
ref_cfg = len(sys.argv) > 3 and sys.argv[3] in ("ref", "librosa")   # the reference's own configurations
lib_cfg = len(sys.argv) > 3 and sys.argv[3] == "librosa"   # extract_mfcc (HMM_lib.py:11-15)
if lib_cfg:
    n_samp = 66150
    audio = 0.2 * torch.randn((n_utt * n_samp,), device='cuda', dtype=torch.float32)
    so = np.arange(0, (n_utt + 1) * n_samp, n_samp, dtype=np.int64)
    plan = sra.MfccPlan.librosa(ctx, 22050)
elif ref_cfg:
    n_samp = 66150                                          # 3 s at 22.05 kHz
    audio = 0.2 * torch.randn((n_utt * n_samp,), device='cuda', dtype=torch.float32)
    so = np.arange(0, (n_utt + 1) * n_samp, n_samp, dtype=np.int64)
    plan = sra.MfccPlan(ctx, 22050, nfft=1024, n_mel=26, delta_order=1, zscore=True)
else:
    plan = sra.MfccPlan(ctx, 16000, nfft=512, n_mel=40, delta_order=order, zscore=zs)
feats, fo = plan.run(audio, so)
torch.cuda.synchronize()
from _warm import warm
warm(lambda: plan.run(audio, so, out=feats))
t0 = time.perf_counter()
for _ in range(5):
    plan.run(audio, so, out=feats)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
L, hop = (551, 220) if ref_cfg else (400, 160)
frames = n_utt * ((n_samp - L) // hop + 1)
if lib_cfg:
    hop, frames = 512, n_utt * (n_samp // 512 + 1)
print("ms", dt * 1e3, "frames/s", frames / dt, "hbm frac", frames * (hop * 4 + plan.n_feat * 4) / dt / 8e12)
