"""Scoring of a ragged batch (65,536 utterances of 100 to 300 frames, mean 200, 39 features, 6 x 8
states, float32): in memory order and with the length order Ragged.length_order() makes."""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
from speech_recognition_amd import _lib
from _warm import warm
M, S, U, D = 6, 8, 65536, 39
ctx = sra.Context.default(0)
lengths = np.random.default_rng(4).integers(100, 301, size=U)
X = torch.randn((int(lengths.sum()), D), device='cuda', dtype=torch.float32)
data = sra.Ragged(ctx, X, lengths, dtype="float32")
rng = np.random.default_rng(8)
bank = sra.HMMBank(ctx, rng.dirichlet(np.ones(S), size=M), rng.dirichlet(np.ones(S), size=(M, S)),
                   rng.standard_normal((M, S, D)), rng.uniform(0.5, 1.5, (M, S, D)))
out = bank.score(data)


def plain():
    _lib.check(_lib.lib.srx_forward_score(ctx.handle, data.X.data_ptr(), data.code, data.offsets.data_ptr(),
                                          data.U, data.n_frames, C.byref(bank._struct), out.data_ptr(),
                                          ctx.stream_ptr()))


for name, fn in (("memory order", plain), ("length order", lambda: bank.score(data, out=out))):
    warm(fn)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}: {ms:.4f} ms  {data.n_frames/ms/1e6:.2f} Gframes/s  {data.n_frames*D*4/ms/1e9:.3f} TB/s X-stream")
