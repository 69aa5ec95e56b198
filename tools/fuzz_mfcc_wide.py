"""Random librosa-convention configurations at 2048 points (the float32 register-transform
kernel) against the numpy oracle: python tools/fuzz_mfcc_wide.py [n_cases] [seed].  Hops from 2
samples up, ragged batches with many tiny utterances, batches barely long enough for the kernel.
Prints the first failure and exits 1."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
from oracle import mfcc_oracle as mo

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
worst = 0.0
for case in range(n_cases):
    sr = int(rng.choice([8000, 16000, 22050, 44100]))
    hop = int(rng.choice([2, 3, 64, 160, 441, 512, 1000, 2048])) if rng.random() < 0.7 else int(rng.integers(2, 2049))
    n_mels = int(rng.integers(4, 129))
    n_mfcc = int(rng.integers(1, min(32, n_mels) + 1))
    pad_mode = str(rng.choice(["constant", "reflect"]))
    top_db = float(rng.choice([0.0, 20.0, 80.0]))
    lo = 1025 if pad_mode == "reflect" else 0
    n_sig = int(rng.integers(1, 40))
    cap = 6000 if hop >= 64 else 2600      # (small hops: many frames per sample)
    lens = [int(rng.integers(lo, cap)) if rng.random() < 0.7 else int(rng.integers(lo, lo + 40)) for _ in range(n_sig)]
    sigs = []
    for n in lens:
        t = np.arange(n) / sr
        s = 0.05 * rng.standard_normal(n) + 0.2 * np.sin(2 * np.pi * rng.uniform(80, 3000) * t)
        if n > 2000 and rng.random() < 0.5:
            s[n // 4: n // 2] = 0.0
        sigs.append(s.astype(np.float32))
    tag = f"case {case}: sr={sr} hop={hop} n_mels={n_mels} n_mfcc={n_mfcc} {pad_mode} top_db={top_db} lens={lens}"
    cfg = mo.LibrosaConfig(sample_rate=sr, n_mfcc=n_mfcc, n_fft=2048, hop_length=hop, n_mels=n_mels,
                           pad_mode=pad_mode, top_db=top_db)
    try:
        f, ln = sra.mfcc.librosa_mfcc_batch(sigs, sr, n_mfcc, dtype="float32", n_fft=2048, hop_length=hop,
                                            n_mels=n_mels, pad_mode=pad_mode, top_db=top_db)
        assert list(ln) == [cfg.n_frames(n) for n in lens]
        off = np.concatenate([[0], np.cumsum(ln)])
        for k, s in enumerate(sigs):
            ref = mo.librosa_mfcc(s.astype(np.float64), cfg)
            got = f[off[k]:off[k + 1]]
            assert got.shape == ref.shape
            if ref.size == 0:
                continue
            err = float(np.max(np.abs(got - ref)) / max(np.max(np.abs(ref)), 1e-30))
            worst = max(worst, err)
            assert np.isfinite(got).all() and err <= 2e-4, (k, err)
    except Exception as e:      # noqa: BLE001
        print("FAILED", tag, repr(e))
        sys.exit(1)
print(f"{n_cases} cases, largest error / largest cepstrum {worst:.2e}")
