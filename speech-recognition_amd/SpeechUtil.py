"""WAV I/O with the reference's names (SpeechUtil.py:16-40).  Playback and
recording (sounddevice) are outside the path and not provided."""
import numpy as np
from scipy.io import wavfile


def read_recording(filename, Fs=16000):
    """Samples of ``filename + '.wav'`` as stored (no resampling; Fs is
    trusted, as in SpeechUtil.py:34-40)."""
    return wavfile.read(f"{filename}.wav")[1]


def save_signal(signal, filename, Fs=16000):
    """SpeechUtil.py:16-18."""
    wavfile.write(f"{filename}.wav", Fs, np.asarray(signal))
