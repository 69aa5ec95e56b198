"""Phase times of cepstra_fast_kernel from a library built with -DSRX_MFCC_STAMPS:
    tools/build_variant.sh stamps mfcc_fast.hip -DSRX_MFCC_STAMPS
    SRX_LIB=build_variants/libsrx_stamps.so python tools/mfcc_stamps.py [ref]
Prints the cycles of one wave's step (4 frames) per phase, averaged over all waves.  A wave's
cycles include the time its SIMD spent on its co-resident waves."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
from speech_recognition_amd._lib import lib
ctx = sra.Context.default(0)
ref = len(sys.argv) > 1 and sys.argv[1] == "ref"
n_utt, n_samp = 8192, (66150 if ref else 48000)
audio = 0.2 * torch.randn((n_utt * n_samp,), device='cuda', dtype=torch.float32)
so = np.arange(0, (n_utt + 1) * n_samp, n_samp, dtype=np.int64)
plan = sra.MfccPlan(ctx, 22050, nfft=1024, n_mel=26, delta_order=1, zscore=True) if ref else \
    sra.MfccPlan(ctx, 16000, nfft=512, n_mel=40, delta_order=2, zscore=True)
feats, fo = plan.run(audio, so)
for _ in range(60):
    plan.run(audio, so, out=feats)
buf = (C.c_ulonglong * 16)()
lib.srx_debug_mfcc_stamps.argtypes = [C.c_void_p]
lib.srx_debug_mfcc_stamps(buf)
for _ in range(20):
    plan.run(audio, so, out=feats)
lib.srx_debug_mfcc_stamps(buf)
v = np.array(list(buf), dtype=np.float64)
steps = v[15]
names = ["loop top + fetch issue", "window + first register FFT", "twiddle, transpose through LDS",
         "second register FFT", "upper-half exchange, unpack, power -> LDS", "mel products + partial sums out",
         "partial sums in, log, log-mel row out", "DCT, row sums, store"]
tot = v[:8].sum()
for n, x in zip(names, v[:8]):
    print(f"{n:45s} {x / steps:9.1f} cycles per wave-step  {100 * x / tot:5.1f} %")
print(f"{'total':45s} {tot / steps:9.1f}   ({int(steps)} wave-steps)")
