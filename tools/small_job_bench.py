"""The reference's own job sizes: time per EM iteration of MyGaussianHMM.fit (one word model,
25 utterances; and a 150-utterance job) with and without graph replay, and the latency of
recognising one recording with models that are already on the device.

    python tools/small_job_bench.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                              # noqa: E402
import speech_recognition_amd as sra                      # noqa: E402
from speech_recognition_amd import recognizer as rec      # noqa: E402

rng = np.random.default_rng(0)
S, D = 3, 26


def fit_time(n_utts, dtype, use_graph, mstep, n_iter=120):
    lengths = rng.integers(60, 240, size=n_utts)
    X = rng.standard_normal((int(lengths.sum()), D))
    m = sra.MyGaussianHMM(n_components=S, n_iter=n_iter, tol=-np.inf, verbose=False, random_state=1,
                          dtype=dtype, mstep=mstep, use_graph=use_graph)
    np.random.seed(3)
    m.fit(X, lengths)                        # warm-up: workspaces, plans
    best = 1e9
    for _ in range(3):
        np.random.seed(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.fit(X, lengths)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / n_iter * 1e3, m.monitor_.iter


for n_utts in (25, 150):
    for dtype in ("float32", "float64"):
        for mstep in ("transitions_only", "full"):
            eager, it = fit_time(n_utts, dtype, False, mstep)
            graph, it2 = fit_time(n_utts, dtype, True, mstep)
            print(f"fit {n_utts:4d} utterances {dtype} {mstep:17s}: {eager:.4f} ms/iteration launch by launch, "
                  f"{graph:.4f} with graph replay ({it} iterations)", flush=True)

# one recording -> label, models resident (what recognize does after the WAV read)
models = {}
for w in rec.WORDS:
    m = sra.MyGaussianHMM(n_components=S, verbose=False, dtype="float32")
    m.startprob_ = rng.dirichlet(np.ones(S))
    m.transmat_ = rng.dirichlet(np.ones(S), size=S)
    m.means_ = rng.standard_normal((S, D))
    m._covars_ = rng.uniform(0.5, 1.5, (S, D))
    m.n_features = D
    models[w] = m
sig = (0.1 * rng.standard_normal(22050 * 2)).astype(np.float32)
sr = 22050


def one(dtype):
    data = sra.mfcc.extract_ragged([sig], sr, dtype=dtype)
    labels, scores = rec.classify(models, data)
    return int(labels.cpu()[0])


for dtype in ("float32", "float64"):
    for _ in range(20):
        one(dtype)
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        one(dtype)
        ts.append(time.perf_counter() - t0)
    ts = np.sort(ts) * 1e3
    print(f"recognize one 2 s recording ({dtype}, models resident, label on the host): "
          f"median {ts[100]:.3f} ms, best {ts[0]:.3f} ms, 90 % {ts[180]:.3f} ms", flush=True)
