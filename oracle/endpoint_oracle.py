"""CPU oracle (test infrastructure) for the end-pointing step in front of the MFCC
path: 10 ms block energies and zero-crossing counts, the Rabiner-Sambur style
thresholds and the run search of the reference's ``AudioMonitor``
(new_speech_detect.py:30-61, :80-89, :197-343, :447-470).

fp64 numpy / plain Python restatement; each function names the reference lines it
follows.  Checked against the imported reference through
``tests/golden/endpoint_golden.npz`` (parity pinned; the reference sums float32
blocks in float32, so energies agree to float32 rounding and the thresholds with
them -- the fixtures keep clear of ties).
"""
import numpy as np

NOISE_PEAK = 4.5       # new_speech_detect.py:84, :208
ZCR_CAP = 25           # new_speech_detect.py:220 (IF)


def block_size(fs, block_ms=10):
    """new_speech_detect.py:34."""
    return int(fs * block_ms / 1000)


def zero_crossings(x):
    """compute_zero_crossing_rate_sum, new_speech_detect.py:52-61: sign changes inside
    the block, zero counted as positive."""
    x = np.asarray(x)
    if x.size < 2:
        return 0.0
    pos = x >= 0
    return float(np.count_nonzero(pos[1:] != pos[:-1]))


def block_features(signal, block):
    """start_saved_audio, new_speech_detect.py:320-329 with :44-47 and :52-61: one
    (sum of |x|, zero crossings) pair per block; the last block may be short."""
    signal = np.asarray(signal, dtype=np.float64)
    starts = range(0, len(signal), block)
    energy = np.array([np.abs(signal[i:i + block]).sum() for i in starts])
    zcr = np.array([zero_crossings(signal[i:i + block]) for i in starts])
    return energy, zcr


def thresholds(energy, signal, fs, block):
    """(ITU, ITL, IZCT), new_speech_detect.py:205-221; None when the peak energy is
    under the noise floor."""
    imx, imn = float(np.max(energy)), float(np.min(energy))
    if imx < NOISE_PEAK:
        return None
    itl = min(0.03 * (imx - imn) + imn, 4 * imn)
    itu = 5 * itl
    first = int(0.1 * fs)
    z0 = np.array([zero_crossings(signal[i:i + block]) for i in range(0, first, block)])
    izct = min(ZCR_CAP, float(z0.mean() + 2 * z0.std()))
    return itu, itl, izct


def runs(energy, itu, itl):
    """The run search of new_speech_detect.py:223-270.  Returns the accepted runs above
    ITL as (start, length): a run above ITU is accepted when it is longer than 5
    blocks, starts in block 0 or reaches the last block; a run above ITL is accepted
    when it is longer than 5 blocks (or reaches the last block) while an accepted upper
    run without a partner starts at or after it."""
    n = len(energy)
    upper, lower = [], []
    u0, c, l0, d = -1, 0, -1, 0
    for i, e in enumerate(energy):
        if e > itu:
            if u0 == -1:
                u0 = i
            c += 1
            if i == n - 1:
                upper.append((u0, c))
                u0, c = -1, 0
        else:
            if c > 5 or u0 == 0:
                upper.append((u0, c))
            u0, c = -1, 0
        if e > itl:
            if l0 == -1:
                l0 = i
            d += 1
            if i == n - 1 and len(upper) > len(lower) and l0 <= upper[-1][0]:
                lower.append((l0, d))
                l0, d = -1, 0
        else:
            if d > 5 and len(upper) > len(lower) and l0 <= upper[-1][0]:
                lower.append((l0, d))
            l0, d = -1, 0
    return lower


def widen(n1, n2, izct, zcr):
    """calc_N1_prime / calc_N2_prime and the 10-block margin,
    new_speech_detect.py:276-279, :297-318."""
    n = len(zcr)
    before = [i for i in range(max(0, n1 - 25), n1 + 1) if zcr[i] > izct]
    n1p = before[0] if len(before) >= 3 else n1
    after = [i for i in range(n2 + 1, min(n2 + 25, n)) if zcr[i] > izct]
    n2p = after[-1] if len(after) >= 3 else n2
    return max(n1p - 10, 0), n2p


def detect(signal, fs, block_ms=10):
    """start_saved_audio + detect_if_noise + detect_speech_start_2,
    new_speech_detect.py:320-334, :80-89, :197-295.  Returns
    (segments [[N1', N1, N2, N2'], ...] in blocks, ITU, ITL, IZCT); no speech gives
    ([], peak energy, 0, 0) -- the reference's two noise outcomes ([] and
    [[[0, 0, 0, 0]], IMX, 0, 0]) both mean "nothing to cut"."""
    block = block_size(fs, block_ms)
    energy, zcr = block_features(signal, block)
    th = thresholds(energy, np.asarray(signal, dtype=np.float64), fs, block)
    if th is None:
        return [], float(np.max(energy)), 0.0, 0.0
    itu, itl, izct = th
    segs = []
    for n1, d in runs(energy, itu, itl):
        n2 = n1 + d
        n1p, n2p = widen(n1, n2, izct, zcr)
        segs.append([n1p, n1, n2, n2p])
    if not segs:
        return [], float(np.max(energy)), 0.0, 0.0
    return segs, itu, itl, izct


def cut_ranges(segments, fs):
    """combine_audio, new_speech_detect.py:447-470: sample ranges [start, end) of the
    pieces that are concatenated (a segment that begins inside the previous one extends
    it)."""
    out = []
    start = end = 0
    i = 0
    while i < len(segments):
        if end != segments[i][3]:
            start, end = segments[i][0], segments[i][3]
        if i + 1 < len(segments) and segments[i + 1][0] < end:
            end = segments[i + 1][3]
        else:
            out.append((int(start * fs * 0.01), int(end * fs * 0.01)))
        i += 1
    return out


def combine(signal, segments, fs):
    """The concatenated pieces (combine_audio's return value)."""
    signal = np.asarray(signal)
    if not segments:
        return signal[:0]
    return np.concatenate([signal[a:b] for a, b in cut_ranges(segments, fs)])
