"""End-pointing with the reference's call surface (``AudioMonitor`` of
new_speech_detect.py), computed on the GPU.

``AudioMonitor.start_saved_audio`` / ``combine_audio`` keep the reference's names,
arguments and return formats (new_speech_detect.py:320-334, :447-470);
``detect_batch`` / ``cut_batch`` are the additive entry points for many recordings
per launch, whose output feeds ``mfcc.extract_batch``.  The block features, thresholds
and the run search are libsrx kernels (``srx_endpoint_detect``); there is no numpy
path.  Only the arithmetic-free glue is host code: merging overlapping segments into
sample ranges as ``combine_audio`` does.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .runtime import Context, dtype_info, ptr


class EndpointResult:
    """Device tensors of one detect_batch call: energy / zcr per block, block offsets,
    segments [R, max_segments, 4] (N1', N1, N2, N2' in blocks), counts [R],
    thresholds [R, 3] (ITU, ITL, IZCT; or peak energy, 0, 0 when nothing was found)."""

    def __init__(self, energy, zcr, block_offsets, segments, counts, thresholds):
        self.energy, self.zcr, self.block_offsets = energy, zcr, block_offsets
        self.segments, self.counts, self.thresholds = segments, counts, thresholds

    def segments_host(self):
        """Per recording, the list of [N1', N1, N2, N2'] lists."""
        seg, cnt = self.segments.cpu().numpy(), self.counts.cpu().numpy()
        if np.any(cnt > seg.shape[1]):
            raise ValueError(f"a recording has {int(cnt.max())} segments; raise max_segments "
                             f"(now {seg.shape[1]})")
        return [seg[r, :cnt[r]].tolist() for r in range(len(cnt))]


class BatchLayout:
    """The offsets of one batch layout on the host and on the device: sample offsets of the
    recordings and the block offsets that follow from them.  detect_batch builds one per call
    from plain offsets; a caller that runs many batches of one layout builds it once
    (`BatchLayout(sample_offsets, fs, ctx=ctx)`) and passes it as sample_offsets."""

    def __init__(self, sample_offsets, fs, *, block_duration_ms=10, ctx=None):
        self.ctx = ctx or Context.default()
        self.fs, self.block_duration_ms = int(fs), int(block_duration_ms)
        self.so = np.asarray(sample_offsets, dtype=np.int64)
        self.block = int(fs * block_duration_ms / 1000)   # new_speech_detect.py:34
        if self.block < 1:
            raise ValueError("block shorter than one sample")
        lens = np.diff(self.so)
        self.bo = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum((lens + self.block - 1) // self.block, out=self.bo[1:])
        self.so_d, self.bo_d = self.ctx.tensor(self.so), self.ctx.tensor(self.bo)


def detect_batch(audio, sample_offsets, fs, *, block_duration_ms=10, noise_peak=4.5, zcr_cap=25,
                 max_segments=16, dtype="float32", ctx=None):
    """Block features and speech segments of many recordings in one launch.
    audio: packed 1-D samples (host array or device tensor), sample_offsets: R+1 host ints or
    a BatchLayout made from them."""
    ctx = ctx or Context.default()
    tdt, code, _ = dtype_info(dtype)
    if isinstance(sample_offsets, BatchLayout):
        lay = sample_offsets
        if lay.ctx is not ctx or lay.fs != int(fs) or lay.block_duration_ms != int(block_duration_ms):
            raise ValueError("the BatchLayout was made for another context, rate or block length")
    else:
        lay = BatchLayout(sample_offsets, fs, block_duration_ms=block_duration_ms, ctx=ctx)
    so, bo = lay.so, lay.bo
    a = ctx.tensor(audio, tdt).reshape(-1)
    if so[-1] != a.numel():
        raise ValueError(f"sample offsets end at {so[-1]} but audio has {a.numel()} samples")
    cfg = _lib.EndpointCfg(int(fs), int(block_duration_ms), int(zcr_cap), 0, float(noise_peak))
    R, nb = len(so) - 1, int(bo[-1])
    dev = ctx.device
    energy = torch.empty((nb,), dtype=torch.float64, device=dev)
    zcr = torch.empty((nb,), dtype=torch.int32, device=dev)
    segs = torch.zeros((R, max_segments, 4), dtype=torch.int32, device=dev)
    counts = torch.zeros((R,), dtype=torch.int32, device=dev)
    thr = torch.zeros((R, 3), dtype=torch.float64, device=dev)
    so_d, bo_d = lay.so_d, lay.bo_d
    check(lib.srx_endpoint_detect(ctx.handle, ptr(a), code, ptr(so_d), R, int(so[-1]), C.byref(cfg),
                                  ptr(bo_d), nb, ptr(energy), ptr(zcr), ptr(segs), ptr(counts),
                                  ptr(thr), int(max_segments), ctx.stream_ptr()),
          "srx_endpoint_detect")
    res = EndpointResult(energy, zcr, bo, segs, counts, thr)
    res._keep = (a, so_d, bo_d)                          # outlive the asynchronous kernels
    return res


def cut_ranges(segments, sample_rate):
    """Sample ranges [start, end) that combine_audio concatenates
    (new_speech_detect.py:447-470): a segment that begins inside the previous one
    extends it."""
    out = []
    start = end = 0
    i = 0
    while i < len(segments):
        if end != segments[i][3]:
            start, end = segments[i][0], segments[i][3]
        if i + 1 < len(segments) and segments[i + 1][0] < end:
            end = segments[i + 1][3]
        else:
            out.append((int(start * sample_rate * 0.01), int(end * sample_rate * 0.01)))
        i += 1
    return out


def cut_batch(audio, sample_offsets, fs, *, dtype="float32", ctx=None, **kw):
    """detect_batch, then the speech pieces of every recording concatenated on the
    device: returns (packed samples, new sample offsets (host), EndpointResult).
    Recordings without speech get length 0."""
    ctx = ctx or Context.default()
    tdt, _, _ = dtype_info(dtype)
    so = np.asarray(sample_offsets, dtype=np.int64)
    a = ctx.tensor(audio, tdt).reshape(-1)
    res = detect_batch(a, so, fs, dtype=dtype, ctx=ctx, **kw)
    pieces, new_so = [], [0]
    for r, segs in enumerate(res.segments_host()):
        n = 0
        for s, e in cut_ranges(segs, fs):
            s, e = min(s, so[r + 1] - so[r]), min(e, so[r + 1] - so[r])
            if e > s:
                pieces.append(a[so[r] + s:so[r] + e])
                n += int(e - s)
        new_so.append(new_so[-1] + n)
    packed = torch.cat(pieces) if pieces else a[:0]
    return packed, np.asarray(new_so, dtype=np.int64), res


def clean_recordings(signals, fs, *, chunk_seconds=1, dtype="float32", ctx=None, to_host=True, **kw):
    """The offline cleaning pass of the reference for many recordings at once
    (process_recorded_audio / process_chunk / get_cleaned_audio,
    new_speech_detect.py:509-517, :555-606): every recording is cut into chunks of
    ``chunk_seconds``, each chunk is end-pointed on its own, and the speech pieces of
    a recording's chunks are concatenated.  All chunks of all recordings go through
    one srx_endpoint_detect launch.  Returns the cleaned recordings (numpy arrays,
    or device tensors with ``to_host=False``); a recording without speech comes back
    empty."""
    ctx = ctx or Context.default()
    tdt, _, _ = dtype_info(dtype)
    chunk = int(fs * chunk_seconds)
    lens = np.array([len(s) for s in signals], dtype=np.int64)
    npdt = np.float64 if dtype == "float64" else np.float32
    packed = np.concatenate([np.asarray(s, dtype=npdt).reshape(-1) for s in signals]) \
        if len(signals) else np.zeros(0, dtype=npdt)
    # chunk boundaries inside every recording
    bounds, owner = [0], []
    base = 0
    for r, n in enumerate(lens):
        for a in range(0, int(n), chunk):
            bounds.append(base + min(a + chunk, int(n)))
            owner.append(r)
        base += int(n)
    cut, new_so, _ = cut_batch(packed, np.asarray(bounds, dtype=np.int64), fs, dtype=dtype, ctx=ctx, **kw)
    out = []
    owner = np.asarray(owner, dtype=np.int64)
    for r in range(len(signals)):
        idx = np.nonzero(owner == r)[0]
        piece = cut[new_so[idx[0]]:new_so[idx[-1] + 1]] if len(idx) else cut[:0]
        out.append(piece.cpu().numpy() if to_host else piece)
    return out


class AudioMonitor:
    """The reference's class for saved audio (new_speech_detect.py:30-43, :320-334,
    :447-470).  ``device`` is accepted for signature compatibility."""

    def __init__(self, device=0, fs=16000, block_duration_ms=10, dtype="float64", ctx=None):
        self.device = device
        self.fs = fs
        self.block_duration_ms = block_duration_ms
        self.block_size = int(fs * block_duration_ms / 1000)
        self.energies = []
        self.zcrs = []
        self.threshold_values = []
        self.dtype = dtype
        self._ctx = ctx

    def start_saved_audio(self, signal):
        """[segments, ITU, ITL, IZCT]; [] for a recording under the noise peak and
        [[[0, 0, 0, 0]], IMX, 0, 0] when no run qualifies, as the reference returns."""
        signal = np.asarray(signal).reshape(-1)
        res = detect_batch(signal, [0, len(signal)], self.fs, block_duration_ms=self.block_duration_ms,
                           dtype=self.dtype, ctx=self._ctx, max_segments=64)
        self.energies = res.energy.cpu().numpy().tolist()
        self.zcrs = res.zcr.cpu().numpy().astype(np.float64).tolist()
        segs = res.segments_host()[0]
        thr = res.thresholds.cpu().numpy()[0]
        if segs:
            self.threshold_values = [segs, float(thr[0]), float(thr[1]), float(thr[2])]
        elif len(self.energies) and thr[0] >= 4.5:
            self.threshold_values = [[[0, 0, 0, 0]], float(thr[0]), 0, 0]
        return self.threshold_values

    def combine_audio(self, signal, thresholds, sample_rate=16000):
        """The speech pieces of ``signal`` concatenated; [] for the noise marker."""
        if len(thresholds) and min(thresholds[0]) == max(thresholds[0]) == 0:
            return []
        signal = np.asarray(signal)
        return np.concatenate([signal[a:b] for a, b in cut_ranges(thresholds, sample_rate)])
