"""MI355X-native MFCC + diagonal-Gaussian HMM hot path (scoring, Baum-Welch,
Viterbi) behind the call surface of Tejashreenk/Speech-Recognition.

    from speech_recognition_amd import hmm, mfcc, recognizer, parallel
    model = hmm.MyGaussianHMM(n_components=3, tol=1e-4, n_iter=2000)
    model.fit(X, lengths); model.score(X)

All arithmetic runs in libsrx.so (hand-written HIP for gfx950, C ABI in
include/srx.h); importing the package fails if that library is not built.
"""
from . import _lib                      # noqa: F401  (raises if libsrx.so is missing)
from . import runtime, hmm, mfcc, endpoint, recognizer, hmm_lib, parallel, SpeechUtil   # noqa: F401
from .hmm import MyGaussianHMM, GaussianHMM, ConvergenceMonitor        # noqa: F401
from .runtime import (Context, HMMBank, Ragged, MfccPlan, GmmEmission, viterbi_dense,   # noqa: F401
                      argmax_scores, column_moments)

HMM_final = recognizer   # the reference's module name for the same functions
HMM_lib = hmm_lib         # likewise (librosa-convention features, hmmlearn-style models)
new_speech_detect = endpoint   # likewise, for AudioMonitor's saved-audio methods

__version__ = "0.1.0"
