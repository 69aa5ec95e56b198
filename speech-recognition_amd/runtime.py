"""Device-side plumbing: one libsrx context per GPU, torch tensors as buffers.

torch is used for device memory, streams and (in parallel.py) the process
group only; every arithmetic step of the path is a libsrx kernel.
"""
import ctypes as C
import threading

import numpy as np
import torch

from . import _lib
from ._lib import lib, check

_DTYPES = {"float32": (torch.float32, _lib.SRX_F32, np.float32),
           "float64": (torch.float64, _lib.SRX_F64, np.float64)}


def dtype_info(dtype):
    key = np.dtype(dtype).name if not isinstance(dtype, str) else dtype
    if key not in _DTYPES:
        raise ValueError(f"dtype must be float32 or float64, got {dtype!r}")
    return _DTYPES[key]


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError(
            "speech_recognition_amd needs a visible MI355X: torch.cuda.is_available() is "
            "False and there is no CPU fallback")


class Context:
    """One per GPU / rank (srx_create)."""

    _lock = threading.Lock()
    _default = {}

    def __init__(self, device=None):
        _require_gpu()
        if device is None:
            device = torch.cuda.current_device()
        self.index = int(torch.device("cuda", device).index if not isinstance(device, int) else device)
        self.device = torch.device("cuda", self.index)
        h = C.c_void_p()
        check(lib.srx_create(self.index, C.byref(h)), "srx_create")
        self._h = h

    @classmethod
    def default(cls, device=None):
        """The process-wide context of (device, current torch stream): work queued on two
        streams must not share one context's scratch buffers, so each stream gets its own."""
        _require_gpu()
        idx = torch.cuda.current_device() if device is None else int(device)
        key = (idx, int(torch.cuda.current_stream(idx).cuda_stream))
        with cls._lock:
            if key not in cls._default:
                cls._default[key] = cls(idx)
            return cls._default[key]

    @property
    def handle(self):
        return self._h

    def stream_ptr(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_exact_f32(self, on):
        """True: float32 scoring stays on exact-f32 matrix instructions; False
        (default): split-bf16 emission contraction."""
        check(lib.srx_set_option(self._h, _lib.SRX_OPT_EXACT_F32, int(bool(on))), "srx_set_option")

    def set_estep_path(self, path):
        """Kernels of the E-step: "auto" (one wave per utterance for the smallest jobs -- at most
        2048 utterances and 4 states, either dtype --, one lane per utterance for small jobs of
        at most 8 states, the fused matrix-core kernel otherwise), "fused", "lane" or "wave"."""
        code = {"auto": _lib.SRX_ESTEP_AUTO, "fused": _lib.SRX_ESTEP_FUSED, "lane": _lib.SRX_ESTEP_LANE,
                "wave": _lib.SRX_ESTEP_WAVE}[path]
        check(lib.srx_set_option(self._h, _lib.SRX_OPT_ESTEP_PATH, code), "srx_set_option")

    def set_viterbi_path(self, path):
        """Kernels of float32 Viterbi decoding for models of at most 8 states: "auto" / "fused"
        (one kernel, log-densities kept on the chip) or "split" (an emission kernel, then a
        recursion kernel)."""
        code = {"auto": _lib.SRX_VITERBI_AUTO, "fused": _lib.SRX_VITERBI_FUSED,
                "split": _lib.SRX_VITERBI_SPLIT}[path]
        check(lib.srx_set_option(self._h, _lib.SRX_OPT_VITERBI_PATH, code), "srx_set_option")

    def workspace_bytes(self):
        return int(lib.srx_workspace_bytes(self._h))

    def tensor(self, array, dtype=None):
        """Host array / tensor -> contiguous device tensor."""
        if isinstance(array, torch.Tensor):
            t = array.to(self.device)
            if dtype is not None:
                t = t.to(dtype)
            return t.contiguous()
        a = np.ascontiguousarray(array)
        t = torch.from_numpy(a).to(self.device)
        if dtype is not None:
            t = t.to(dtype)
        return t

    def close(self):
        if self._h:
            from . import mfcc as _fe
            _fe.drop_plans(self)
            lib.srx_destroy(self._h)
            self._h = C.c_void_p()
            with Context._lock:
                for k in [k for k, v in Context._default.items() if v is self]:
                    del Context._default[k]


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()


def offsets_from_lengths(lengths, total=None):
    """lengths -> int64 offsets[U+1]; the reference's length check
    (my_hmmlearn/utils.py:284-286)."""
    lengths = np.asarray(lengths, dtype=np.int64).reshape(-1)
    if (lengths < 0).any():
        raise ValueError("negative utterance length")
    off = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(lengths, out=off[1:])
    if total is not None and off[-1] != total:
        raise ValueError(f"lengths array {lengths} doesn't sum to {total} samples")
    return off


class Ragged:
    """Packed frames [n_frames, D] on the device plus utterance offsets."""

    def __init__(self, ctx, X, lengths=None, dtype="float32", offsets=None, offsets_host=None):
        """X: host array or device tensor (a device tensor of the right dtype is taken as it
        is, no copy).  lengths: frames per utterance (host); or offsets: int64 device tensor
        [U+1], with offsets_host its host copy if the caller has one."""
        tdt, self.code, _ = dtype_info(dtype)
        self.ctx = ctx
        if isinstance(X, torch.Tensor):
            if X.dim() != 2:
                raise ValueError("X must be 2-D (n_frames, n_features)")
            self.X = X.to(ctx.device, tdt).contiguous()
        else:
            X = np.asarray(X)
            if X.ndim != 2:
                raise ValueError("X must be 2-D (n_frames, n_features)")
            self.X = ctx.tensor(X, tdt)
        self.n_frames, self.D = int(self.X.shape[0]), int(self.X.shape[1])
        if offsets is not None:
            self.offsets_host = None if offsets_host is None else np.asarray(offsets_host, dtype=np.int64)
            self.offsets = offsets.to(ctx.device, torch.int64).contiguous()
            self.U = int(self.offsets.numel()) - 1
        else:
            if lengths is None:
                lengths = [self.n_frames]
            off = offsets_from_lengths(lengths, self.n_frames)
            self.offsets_host = off
            self.offsets = ctx.tensor(off)
            self.U = len(off) - 1
        self._length_order = False           # not computed yet

    def length_order(self):
        """Visiting order for the batched kernels: utterances by length in 32 descending
        buckets (stable inside a bucket, so neighbours in memory stay neighbours), as an int64
        device tensor; None when all utterances have the same length or there are too few to
        matter.  Computed once per object."""
        if self._length_order is False:
            order = None
            if self.U <= 32:                     # two batches at most: nothing to gain
                lengths = np.zeros(0, dtype=np.int64)
            elif self.offsets_host is not None:
                lengths = np.diff(np.asarray(self.offsets_host, dtype=np.int64))
            else:
                lengths = (self.offsets[1:] - self.offsets[:-1]).cpu().numpy()
            if lengths.size and lengths.min() != lengths.max():
                bucket = lengths * 32 // (int(lengths.max()) + 1)
                order = self.ctx.tensor(np.argsort(-bucket, kind="stable").astype(np.int64))
            self._length_order = order
        return self._length_order


def argmax_scores(ctx, scores, out=None):
    """max(scores, key=scores.get) of recognize (HMM_final.py:51-52) for every row of a
    (U, M) float64 score matrix on the device: int32 index of the first best column."""
    if scores.dim() != 2 or scores.dtype != torch.float64 or not scores.is_contiguous():
        raise ValueError("scores must be a contiguous (U, M) float64 device tensor")
    U, M = int(scores.shape[0]), int(scores.shape[1])
    if out is None:
        out = torch.empty((U,), dtype=torch.int32, device=ctx.device)
    check(lib.srx_argmax_scores(ctx.handle, ptr(scores), U, M, ptr(out), ctx.stream_ptr()),
          "srx_argmax_scores")
    return out


def column_moments(ctx, X):
    """(mean, variance with ddof = 1) of every column of a (n, D) device tensor, float64
    device tensors: the diagonal of np.cov(X.T) that MyGaussianHMM.init uses
    (my_hmmlearn/hmm.py:131-132)."""
    code = {torch.float32: _lib.SRX_F32, torch.float64: _lib.SRX_F64}[X.dtype]
    X = X.contiguous()
    n, D = int(X.shape[0]), int(X.shape[1])
    mean = torch.empty((D,), dtype=torch.float64, device=ctx.device)
    var = torch.empty((D,), dtype=torch.float64, device=ctx.device)
    check(lib.srx_column_moments(ctx.handle, ptr(X), code, n, D, ptr(mean), ptr(var), ctx.stream_ptr()),
          "srx_column_moments")
    return mean, var


class HMMBank:
    """n_models diagonal-Gaussian HMMs with common (S, D), parameters resident
    on the device as float64 (srx_models_t)."""

    def __init__(self, ctx, startprob, transmat, means, covars):
        self.ctx = ctx
        sp = np.asarray(startprob, dtype=np.float64)
        A = np.asarray(transmat, dtype=np.float64)
        mu = np.asarray(means, dtype=np.float64)
        cv = np.asarray(covars, dtype=np.float64)
        if sp.ndim == 1:
            sp, A, mu, cv = sp[None], A[None], mu[None], cv[None]
        M, S = sp.shape
        if A.shape != (M, S, S) or mu.ndim != 3 or mu.shape[:2] != (M, S) or cv.shape != mu.shape:
            raise ValueError("Shape mismatch")
        if cv.ndim != 3:
            raise ValueError("'diag' covars must have shape (n_components, n_dim)")
        if np.any(cv <= 0):
            raise ValueError("'diag' covars must be positive")
        self.M, self.S, self.D = M, S, int(mu.shape[2])
        # one allocation for the four arrays, so that a snapshot of the whole bank is one copy
        flat = np.concatenate([sp.ravel(), A.ravel(), mu.ravel(), cv.ravel()])
        self._flat = ctx.tensor(flat)
        n = [0, sp.size, sp.size + A.size, sp.size + A.size + mu.size, flat.size]
        self.startprob = self._flat[n[0]:n[1]].view(sp.shape)
        self.transmat = self._flat[n[1]:n[2]].view(A.shape)
        self.means = self._flat[n[2]:n[3]].view(mu.shape)
        self.covars = self._flat[n[3]:n[4]].view(cv.shape)
        self._struct = _lib.Models(M, S, self.D, self.startprob.data_ptr(),
                                   self.transmat.data_ptr(), self.means.data_ptr(),
                                   self.covars.data_ptr())
        self.stats_size = int(lib.srx_stats_size(S, self.D))

    @classmethod
    def from_models(cls, ctx, models):
        """models: sequence of objects with startprob_/transmat_/means_/_covars_."""
        return cls(ctx, np.stack([m.startprob_ for m in models]),
                   np.stack([m.transmat_ for m in models]),
                   np.stack([m.means_ for m in models]),
                   np.stack([m._covars_ for m in models]))

    def conditioning(self):
        """max of mean^2 / variance over the bank (synchronises).  The matrix-core paths
        evaluate the Gaussian exponent in expanded form, cst + x mean/var - x^2/(2 var), whose
        rounding error per frame is eps * sum_d (|x mean|/var + x^2/(2 var)) with eps = 2^-16
        for float32 scoring (split-bf16 operands), 2^-23 for the exact-f32 mode and the fused
        E-step, 2^-52 in float64: harmless for z-scored features such as the front-end's
        (this number is then of order 10), not for raw features far from zero in units of
        their spread (1e6 and more); use float64 for those, or shift the features."""
        return float((self.means * self.means / self.covars).max().item())

    def _check(self, data):
        if data.D != self.D:
            raise ValueError(
                f"Unexpected number of dimensions, got {data.D} but expected {self.D}")

    def _utt_model(self, data, utt_model):
        """Model index per utterance as an int32 device tensor; host arrays are checked
        (the kernels index their tables with these numbers), device tensors are taken
        as they are."""
        if utt_model is None:
            return None
        if isinstance(utt_model, torch.Tensor):
            if utt_model.dtype != torch.int32 or utt_model.numel() != data.U:
                raise ValueError("utt_model must be int32 with one entry per utterance")
            return utt_model
        um = np.asarray(utt_model, dtype=np.int32)
        if um.shape != (data.U,):
            raise ValueError("utt_model must have one entry per utterance")
        if um.size and (um.min() < 0 or um.max() >= self.M):
            raise ValueError("utt_model entry outside [0, n_models)")
        return self.ctx.tensor(um)

    # -- kernels ---------------------------------------------------------
    def emission_loglik(self, data):
        self._check(data)
        out = torch.empty((data.n_frames, self.M * self.S), dtype=data.X.dtype, device=self.ctx.device)
        check(lib.srx_emission_loglik(self.ctx.handle, ptr(data.X), data.code, data.n_frames,
                                      C.byref(self._struct), ptr(out), self.ctx.stream_ptr()),
              "srx_emission_loglik")
        return out

    def score(self, data, out=None, generic=False):
        """(U, M) forward log-probabilities, float64, on the device."""
        self._check(data)
        if out is None:
            out = torch.empty((data.U, self.M), dtype=torch.float64, device=self.ctx.device)
        # ragged float32 batches: utterances of similar length share a wave
        order = data.length_order() if not generic and data.X.dtype == torch.float32 else None
        if order is not None:
            check(lib.srx_forward_score_ordered(self.ctx.handle, ptr(data.X), data.code, ptr(data.offsets),
                                                data.U, data.n_frames, ptr(order), C.byref(self._struct),
                                                ptr(out), self.ctx.stream_ptr()), "srx_forward_score_ordered")
            return out
        fn = lib.srx_forward_score_generic if generic else lib.srx_forward_score
        check(fn(self.ctx.handle, ptr(data.X), data.code, ptr(data.offsets), data.U, data.n_frames,
                 C.byref(self._struct), ptr(out), self.ctx.stream_ptr()), "srx_forward_score")
        return out

    def classify(self, data, scores=None, labels=None):
        """(labels (U,) int32, scores (U, M) float64), both on the device: every utterance
        scored against all models and given the index of its first best one -- recognize
        (HMM_final.py:49-52) for a batch, nothing leaves the GPU."""
        scores = self.score(data, out=scores)
        return argmax_scores(self.ctx, scores, out=labels), scores

    def score_static(self, data, out):
        """``score`` through srx_forward_score_static: float32 only, no device memory is
        allocated (``out`` must be the caller's (U, M) float64 tensor)."""
        self._check(data)
        if out is None or tuple(out.shape) != (data.U, self.M) or out.dtype != torch.float64:
            raise ValueError("out must be a (U, M) float64 device tensor")
        check(lib.srx_forward_score_static(self.ctx.handle, ptr(data.X), data.code, ptr(data.offsets),
                                           data.U, data.n_frames, C.byref(self._struct), ptr(out),
                                           self.ctx.stream_ptr()), "srx_forward_score_static")
        return out

    def viterbi(self, data, utt_model=None):
        self._check(data)
        path = torch.empty((data.n_frames,), dtype=torch.int32, device=self.ctx.device)
        lp = torch.empty((data.U,), dtype=torch.float64, device=self.ctx.device)
        um = self._utt_model(data, utt_model)
        check(lib.srx_viterbi(self.ctx.handle, ptr(data.X), data.code, ptr(data.offsets), data.U,
                              data.n_frames, ptr(um), C.byref(self._struct), ptr(path), ptr(lp),
                              self.ctx.stream_ptr()), "srx_viterbi")
        return lp, path

    def posteriors(self, data, utt_model=None, out=None):
        """(per-utterance log-probability (U,) float64, posteriors (n_frames, S) in
        the data's dtype): score_scaling with compute_posteriors=True,
        my_hmmlearn/hmm.py:308-328, each utterance under its own model.  out: a contiguous
        (n_frames, S) tensor of the data's dtype to write the posteriors into."""
        self._check(data)
        if out is None:
            out = torch.empty((data.n_frames, self.S), dtype=data.X.dtype, device=self.ctx.device)
        if out.shape != (data.n_frames, self.S) or out.dtype != data.X.dtype or \
                not out.is_contiguous() or out.device != data.X.device:
            raise ValueError("out must be a contiguous (n_frames, n_states) tensor of the data's "
                             "dtype on the data's device")
        gamma = out
        lp = torch.empty((data.U,), dtype=torch.float64, device=self.ctx.device)
        um = self._utt_model(data, utt_model)
        check(lib.srx_posteriors(self.ctx.handle, ptr(data.X), data.code, ptr(data.offsets), data.U,
                                 data.n_frames, ptr(um), C.byref(self._struct), ptr(gamma), ptr(lp),
                                 self.ctx.stream_ptr()), "srx_posteriors")
        return lp, gamma

    def estep(self, data, utt_model=None, utt_order=None, out=None, generic=False, stats="all"):
        """Packed statistics (M, stats_size) float64 on the device.  stats: what the caller
        will use of them -- "all", or "transitions" (count, log-probability, start and
        transition sums: what the reference's own fit() re-estimates from,
        my_hmmlearn/hmm.py:435-451); with "transitions" the sums over the features come back
        zero, the kernels leave out their pass over X."""
        self._check(data)
        if out is None:
            out = torch.empty((self.M, self.stats_size), dtype=torch.float64, device=self.ctx.device)
        which = {"all": _lib.SRX_STATS_ALL, "transitions": _lib.SRX_STATS_TRANSITIONS}[stats]
        utt_model = self._utt_model(data, utt_model)
        if generic:
            if which != _lib.SRX_STATS_ALL:
                raise ValueError("the shape-generic E-step computes all statistics")
            check(lib.srx_estep_stats_generic(self.ctx.handle, ptr(data.X), data.code, ptr(data.offsets),
                                              data.U, data.n_frames, ptr(utt_model), ptr(utt_order),
                                              C.byref(self._struct), ptr(out), self.ctx.stream_ptr()),
                  "srx_estep_stats_generic")
        else:
            check(lib.srx_estep_stats_select(self.ctx.handle, ptr(data.X), data.code, ptr(data.offsets),
                                             data.U, data.n_frames, ptr(utt_model), ptr(utt_order),
                                             C.byref(self._struct), which, ptr(out),
                                             self.ctx.stream_ptr()), "srx_estep_stats")
        return out

    def mstep(self, stats, priors=None, mode="full"):
        pr = priors if priors is not None else _lib.Priors(1.0, 1.0, 0.0, 0.0, 1e-2, 1.0)
        code = {"full": _lib.SRX_MSTEP_FULL,
                "transitions_only": _lib.SRX_MSTEP_TRANSITIONS_ONLY}[mode]
        check(lib.srx_mstep(self.ctx.handle, ptr(stats), C.byref(pr), code, C.byref(self._struct),
                            self.ctx.stream_ptr()), "srx_mstep")

    def snapshot(self, out=None):
        """Device copy of all parameters (one launch); ``restore`` puts it back."""
        if out is None:
            return self._flat.clone()
        out.copy_(self._flat)
        return out

    def restore(self, snap):
        self._flat.copy_(snap)

    # -- host views --------------------------------------------------------
    def params_host(self):
        return (self.startprob.cpu().numpy(), self.transmat.cpu().numpy(),
                self.means.cpu().numpy(), self.covars.cpu().numpy())

    def unpack_stats(self, stats):
        """Packed record(s) -> list of dicts with the reference's keys
        (initialize_sufficient_statistics, my_hmmlearn/hmm.py:224-235) plus
        'logprob'."""
        s = stats.detach().cpu().numpy().reshape(self.M, -1)
        S, D = self.S, self.D
        out = []
        for r in s:
            o = 2
            d = {"nobs": int(round(r[0])), "logprob": float(r[1])}
            d["start"] = r[o:o + S].copy(); o += S
            d["trans"] = r[o:o + S * S].reshape(S, S).copy(); o += S * S
            d["post"] = r[o:o + S].copy(); o += S
            d["obs"] = r[o:o + S * D].reshape(S, D).copy(); o += S * D
            d["obs**2"] = r[o:o + S * D].reshape(S, D).copy()
            out.append(d)
        return out


class GmmEmission:
    """Large emission model: S states, each a mixture of n_mix diagonal Gaussians
    (srx_gmm_t).  ``loglik(data)`` -> (n_frames, S) log-densities; with n_mix = 1
    and weight 1 this is the reference's _log_multivariate_normal_density_diag
    (my_hmmlearn/hmm.py:154-163)."""

    def __init__(self, ctx, weights, means, covars):
        self.ctx = ctx
        w = np.asarray(weights, dtype=np.float64)
        mu = np.asarray(means, dtype=np.float64)
        cv = np.asarray(covars, dtype=np.float64)
        if mu.ndim != 3 or cv.shape != mu.shape or w.shape != mu.shape[:2]:
            raise ValueError("Shape mismatch")
        if np.any(cv <= 0):
            raise ValueError("'diag' covars must be positive")
        if np.any(w < 0):
            raise ValueError("mixture weights must be non-negative")
        self.S, self.n_mix, self.D = (int(v) for v in mu.shape)
        self.weights, self.means, self.covars = ctx.tensor(w), ctx.tensor(mu), ctx.tensor(cv)
        self._struct = _lib.Gmm(self.S, self.n_mix, self.D, self.weights.data_ptr(),
                                self.means.data_ptr(), self.covars.data_ptr())

    def loglik(self, data, out=None):
        if data.D != self.D:
            raise ValueError(
                f"Unexpected number of dimensions, got {data.D} but expected {self.D}")
        if out is None:
            out = torch.empty((data.n_frames, self.S), dtype=data.X.dtype, device=self.ctx.device)
        check(lib.srx_gmm_emission_loglik(self.ctx.handle, ptr(data.X), data.code, data.n_frames,
                                          C.byref(self._struct), ptr(out), self.ctx.stream_ptr()),
              "srx_gmm_emission_loglik")
        return out


def viterbi_dense(ctx, loglik, offsets, startprob, transmat):
    """utils.viterbi (my_hmmlearn/utils.py:243-266) for every utterance of a packed
    (n_frames, S) log-density tensor over one dense model (S <= 1024).
    offsets: device int64 [U+1].  Returns (logprob (U,) float64, path (n_frames,) int32)."""
    S = int(loglik.shape[1])
    sp = startprob if isinstance(startprob, torch.Tensor) else ctx.tensor(np.asarray(startprob, dtype=np.float64))
    A = transmat if isinstance(transmat, torch.Tensor) else ctx.tensor(np.asarray(transmat, dtype=np.float64))
    if tuple(sp.shape) != (S,) or tuple(A.shape) != (S, S):
        raise ValueError("Shape mismatch")
    code = {torch.float32: _lib.SRX_F32, torch.float64: _lib.SRX_F64}[loglik.dtype]
    U = int(offsets.numel()) - 1
    n_frames = int(loglik.shape[0])
    path = torch.empty((n_frames,), dtype=torch.int32, device=ctx.device)
    lp = torch.empty((U,), dtype=torch.float64, device=ctx.device)
    check(lib.srx_viterbi_dense(ctx.handle, ptr(loglik), code, ptr(offsets), U, n_frames, S, ptr(sp),
                                ptr(A), ptr(path), ptr(lp), ctx.stream_ptr()), "srx_viterbi_dense")
    return lp, path


class MfccPlan:
    """Front-end tables on the device for one configuration (srx_mfcc_plan_*)."""

    def __init__(self, ctx, sample_rate, frame_size=0.025, frame_stride=0.01, nfft=1024, n_mel=26,
                 mel_sample_rate=16000, n_ceps=13, delta_order=1, delta_n=2, log_mult=20.0,
                 zscore=True, frame_len=None, hop=None, window="hamming", center=None,
                 mel_style="htk", power_norm=False, log_floor=0.0, top_db=0.0):
        self.ctx = ctx
        # mfcc.py:18-19: float products truncated where they are used
        if frame_len is None:
            frame_len = int(frame_size * sample_rate)
        if hop is None:
            hop = int(frame_stride * sample_rate)
        try:
            codes = ({"hamming": 0, "hann": 1}[window], {None: 0, "constant": 1, "reflect": 2}[center],
                     {"htk": 0, "slaney": 1}[mel_style])
        except KeyError as e:
            raise ValueError(f"unknown front-end convention {e.args[0]!r}") from None
        self.cfg = _lib.MfccCfg(int(sample_rate), int(frame_len), int(hop), int(nfft), int(n_mel),
                                int(mel_sample_rate), int(n_ceps), int(delta_order), int(delta_n),
                                int(bool(zscore)), float(log_mult), codes[0], codes[1], codes[2],
                                int(bool(power_norm)), float(log_floor), float(top_db))
        self.n_feat = int(n_ceps) * (1 + int(delta_order))
        p = C.c_void_p()
        check(lib.srx_mfcc_plan_create(ctx.handle, C.byref(self.cfg), C.byref(p)),
              "srx_mfcc_plan_create")
        self._p = p

    @classmethod
    def librosa(cls, ctx, sample_rate, n_mfcc=13, n_fft=2048, hop_length=512, n_mels=128,
                pad_mode="constant", amin=1e-10, top_db=80.0, zscore=True):
        """The conventions of ``librosa.feature.mfcc(y=, sr=, n_mfcc=)`` as extract_mfcc
        calls it (HMM_lib.py:11-15): centred frames of n_fft samples every hop_length,
        periodic Hann, |rfft|^2, Slaney mel filters up to sr/2, 10*log10 floored at amin,
        raised to (utterance maximum - top_db), DCT-II 'ortho'; no deltas."""
        return cls(ctx, sample_rate, nfft=n_fft, n_mel=n_mels, mel_sample_rate=sample_rate,
                   n_ceps=n_mfcc, delta_order=0, log_mult=10.0, zscore=zscore, frame_len=n_fft,
                   hop=hop_length, window="hann", center=pad_mode, mel_style="slaney",
                   power_norm=True, log_floor=amin, top_db=top_db or 0.0)

    def out_frames(self, n_samples):
        return int(lib.srx_mfcc_out_frames(C.byref(self.cfg), int(n_samples)))

    def prepare(self, sample_offsets_host):
        """Everything a batch needs besides the samples, from its sample offsets: frames per
        utterance, the offsets on the device, the slices the C calls take.  ``run`` keeps the
        last one, so repeated batches of the same shape (a stream of equal chunks) upload
        nothing but samples."""
        so = np.array(sample_offsets_host, dtype=np.int64)
        lens = np.diff(so)
        L, hop, dn = self.cfg.frame_len, self.cfg.hop, self.cfg.delta_n
        if self.cfg.center:
            raw = np.where(lens > 0, lens // hop + 1, 0)
            if self.cfg.center == 2 and (lens <= self.cfg.nfft // 2).any():
                raise ValueError("reflected padding needs more than nfft/2 samples per utterance")
        else:
            raw = np.where(lens >= L, (lens - L) // hop + 1, 0)
        outf = raw if self.cfg.delta_order == 0 else np.maximum(raw - 2 * dn, 0)
        fo = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(outf, out=fo[1:])
        # The fast kernels address one call's samples with 31 bits: longer buffers go in
        # slices of whole utterances (an utterance that is itself longer runs alone,
        # through the generic kernels).
        limit = (1 << 31) - 512
        slices = []
        u0 = 0
        while u0 < len(lens):
            u1 = int(np.searchsorted(so, so[u0] + limit, side="right")) - 1
            u1 = max(u1, u0 + 1)
            if fo[u1] > fo[u0]:
                slices.append((u0, u1, self.ctx.tensor(so[u0:u1 + 1] - so[u0]),
                               self.ctx.tensor(fo[u0:u1 + 1] - fo[u0])))
            u0 = u1
        # (one slice that starts at zero: its frame offsets are the batch's)
        fo_dev = slices[0][3] if len(slices) == 1 and slices[0][0] == 0 and slices[0][1] == len(lens) \
            else self.ctx.tensor(fo)
        return {"so": so, "fo": fo, "slices": slices, "fo_dev": fo_dev}

    def run(self, audio, sample_offsets_host, dtype="float32", out=None):
        """audio: packed 1-D device tensor (or host array); returns
        (features [sum T, n_feat] on the device, frame offsets host int64)."""
        tdt, code, _ = dtype_info(dtype)
        so_in = np.asarray(sample_offsets_host, dtype=np.int64)
        prep = getattr(self, "_prepared", None)
        if prep is None or prep["so"].shape != so_in.shape or not np.array_equal(prep["so"], so_in):
            prep = self._prepared = self.prepare(so_in)
        so, fo = prep["so"], prep["fo"]
        a = self.ctx.tensor(audio, tdt).reshape(-1)
        if so[-1] != a.numel():
            raise ValueError(f"sample offsets end at {so[-1]} but audio has {a.numel()} samples")
        if out is None:
            out = torch.empty((int(fo[-1]), self.n_feat), dtype=tdt, device=self.ctx.device)
        for u0, u1, so_d, fo_d in prep["slices"]:
            check(lib.srx_mfcc_batch(self.ctx.handle, self._p, ptr(a[so[u0]:so[u1]]), code,
                                     ptr(so_d), u1 - u0, int(so[u1] - so[u0]),
                                     ptr(out[fo[u0]:fo[u1]]), ptr(fo_d), self.ctx.stream_ptr()),
                  "srx_mfcc_batch")
        self._keep = a          # the samples must outlive the asynchronous kernels
        return out, fo

    def run_ragged(self, audio, sample_offsets_host, dtype="float32", out=None):
        """``run`` with the result as a ``Ragged`` batch on the device (features and frame
        offsets stay where the kernels wrote them): what scoring, decoding and the E-step take."""
        feats, fo = self.run(audio, sample_offsets_host, dtype=dtype, out=out)
        return Ragged(self.ctx, feats, dtype=dtype, offsets=self._prepared["fo_dev"], offsets_host=fo)

    def close(self):
        if self._p:
            lib.srx_mfcc_plan_destroy(self._p)
            self._p = C.c_void_p()


def filterbank_host(nfft, mel_sample_rate=16000, n_mel=26, mel_style="htk"):
    """Dense filter matrix as libsrx builds it (no GPU needed)."""
    cfg = _lib.MfccCfg(16000, 400, 160, int(nfft), int(n_mel), int(mel_sample_rate),
                       min(13, int(n_mel)), 1, 2, 1, 20.0, 0, 0, {"htk": 0, "slaney": 1}[mel_style])
    out = np.zeros((n_mel, nfft // 2 + 1), dtype=np.float64)
    check(lib.srx_mfcc_filterbank_host(C.byref(cfg), out.ctypes.data_as(C.c_void_p)),
          "srx_mfcc_filterbank_host")
    return out
