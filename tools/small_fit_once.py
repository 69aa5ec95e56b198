"""One 200-iteration fit of a 3-state model on 25 utterances (the reference's per-word job), for
kernel traces: python tools/small_fit_once.py [float64]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
rng = np.random.default_rng(0)
lengths = rng.integers(60, 240, size=25)
X = rng.standard_normal((int(lengths.sum()), 26))
dtype = sys.argv[1] if len(sys.argv) > 1 else "float32"
m = sra.MyGaussianHMM(n_components=3, n_iter=200, tol=-np.inf, verbose=False, random_state=1, dtype=dtype)
np.random.seed(3)
m.fit(X, lengths)
