"""ctypes binding of libsrx.so (the C ABI declared in include/srx.h).

There is no CPU fallback: if the library is missing or was not built, importing
this module raises, and every compute call needs a visible MI355X.
"""
import ctypes as C
import os

# torch ships its own HIP runtime (soname libamdhip64.so.7).  It must be loaded
# first so that libsrx.so binds to that same runtime instead of pulling a second
# copy from /opt/rocm into the process (two runtimes cannot share the GPU).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRX_LIB") or os.path.join(_HERE, "libsrx.so")

SRX_F32, SRX_F64 = 0, 1
SRX_MSTEP_FULL, SRX_MSTEP_TRANSITIONS_ONLY = 0, 1
SRX_OPT_EXACT_F32 = 1
SRX_OPT_ESTEP_PATH = 2
SRX_OPT_VITERBI_PATH = 3
SRX_VITERBI_AUTO, SRX_VITERBI_FUSED, SRX_VITERBI_SPLIT = 0, 1, 2
SRX_STATS_ALL, SRX_STATS_TRANSITIONS = 0, 1
SRX_ESTEP_AUTO, SRX_ESTEP_FUSED, SRX_ESTEP_LANE, SRX_ESTEP_WAVE = 0, 1, 2, 3
SRX_EINVAL, SRX_EHIP, SRX_ENOMEM, SRX_EUNSUPPORTED = -1, -2, -3, -4


class SrxError(RuntimeError):
    pass


class Models(C.Structure):
    _fields_ = [("n_models", C.c_int32), ("n_states", C.c_int32), ("n_features", C.c_int32),
                ("startprob", C.c_void_p), ("transmat", C.c_void_p),
                ("means", C.c_void_p), ("covars", C.c_void_p)]


class Gmm(C.Structure):
    _fields_ = [("n_states", C.c_int32), ("n_mix", C.c_int32), ("n_features", C.c_int32),
                ("weights", C.c_void_p), ("means", C.c_void_p), ("covars", C.c_void_p)]


class Priors(C.Structure):
    _fields_ = [("startprob_prior", C.c_double), ("transmat_prior", C.c_double),
                ("means_prior", C.c_double), ("means_weight", C.c_double),
                ("covars_prior", C.c_double), ("covars_weight", C.c_double)]


class MfccCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("frame_len", C.c_int32), ("hop", C.c_int32),
                ("nfft", C.c_int32), ("n_mel", C.c_int32), ("mel_sample_rate", C.c_int32),
                ("n_ceps", C.c_int32), ("delta_order", C.c_int32), ("delta_n", C.c_int32),
                ("zscore", C.c_int32), ("log_mult", C.c_double),
                # librosa-style conventions (all zero = the reference's mfcc.py)
                ("window", C.c_int32), ("center", C.c_int32), ("mel_style", C.c_int32),
                ("power_norm", C.c_int32), ("log_floor", C.c_double), ("top_db", C.c_double)]


class EndpointCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("block_ms", C.c_int32), ("zcr_cap", C.c_int32),
                ("reserved", C.c_int32), ("noise_peak", C.c_double)]


_vp, _i64, _i32 = C.c_void_p, C.c_int64, C.c_int32

# name -> (restype, argtypes); must list every symbol include/srx.h declares.
SIGNATURES = {
    "srx_version": (C.c_int, []),
    "srx_last_error": (C.c_char_p, []),
    "srx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "srx_destroy": (C.c_int, [_vp]),
    "srx_set_option": (C.c_int, [_vp, C.c_int, C.c_int]),
    "srx_workspace_bytes": (_i64, [_vp]),
    "srx_stats_size": (_i64, [_i32, _i32]),
    "srx_mfcc_plan_create": (C.c_int, [_vp, C.POINTER(MfccCfg), C.POINTER(_vp)]),
    "srx_mfcc_plan_destroy": (C.c_int, [_vp]),
    "srx_mfcc_out_frames": (_i64, [C.POINTER(MfccCfg), _i64]),
    "srx_mfcc_batch": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _i64, _i64, _vp, _vp, _vp]),
    "srx_mfcc_filterbank_host": (C.c_int, [C.POINTER(MfccCfg), _vp]),
    "srx_endpoint_blocks": (_i64, [C.POINTER(EndpointCfg), _i64]),
    "srx_endpoint_detect": (C.c_int, [_vp, _vp, C.c_int, _vp, _i64, _i64, C.POINTER(EndpointCfg), _vp,
                                      _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "srx_emission_loglik": (C.c_int, [_vp, _vp, C.c_int, _i64, C.POINTER(Models), _vp, _vp]),
    "srx_forward_score": (C.c_int, [_vp, _vp, C.c_int, _vp, _i64, _i64, C.POINTER(Models), _vp, _vp]),
    "srx_forward_score_ordered": (C.c_int, [_vp, _vp, C.c_int, _vp, _i64, _i64, _vp, C.POINTER(Models), _vp, _vp]),
    "srx_forward_score_generic": (C.c_int, [_vp, _vp, C.c_int, _vp, _i64, _i64, C.POINTER(Models), _vp, _vp]),
    "srx_forward_score_static": (C.c_int, [_vp, _vp, C.c_int, _vp, _i64, _i64, C.POINTER(Models), _vp, _vp]),
    "srx_static_scratch_bytes": (_i64, []),
    "srx_viterbi": (C.c_int, [_vp, _vp, C.c_int, _vp, _i64, _i64, _vp, C.POINTER(Models), _vp, _vp, _vp]),
    "srx_estep_stats": (C.c_int, [_vp, _vp, C.c_int, _vp, _i64, _i64, _vp, _vp, C.POINTER(Models), _vp, _vp]),
    "srx_estep_stats_select": (C.c_int, [_vp, _vp, C.c_int, _vp, _i64, _i64, _vp, _vp, C.POINTER(Models), C.c_int, _vp, _vp]),
    "srx_estep_stats_generic": (C.c_int, [_vp, _vp, C.c_int, _vp, _i64, _i64, _vp, _vp, C.POINTER(Models), _vp, _vp]),
    "srx_allreduce_stats": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "srx_argmax_scores": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "srx_column_moments": (C.c_int, [_vp, _vp, C.c_int, _i64, _i32, _vp, _vp, _vp]),
    "srx_mstep": (C.c_int, [_vp, _vp, C.POINTER(Priors), C.c_int, C.POINTER(Models), _vp]),
    "srx_gmm_emission_loglik": (C.c_int, [_vp, _vp, C.c_int, _i64, C.POINTER(Gmm), _vp, _vp]),
    "srx_posteriors": (C.c_int, [_vp, _vp, C.c_int, _vp, _i64, _i64, _vp, C.POINTER(Models), _vp, _vp, _vp]),
    "srx_viterbi_dense": (C.c_int, [_vp, _vp, C.c_int, _vp, _i64, _i64, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python speech-recognition_amd/build.py` "
        "(hipcc, gfx950). There is no CPU fallback.")

lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc, what=""):
    """Map a negative status to the exception the reference raises for the
    same mistake (ValueError for shape errors, utils.py:134-135)."""
    if rc == 0:
        return
    msg = lib.srx_last_error().decode("utf-8", "replace")
    if rc == SRX_EINVAL:
        raise ValueError(msg)
    if rc == SRX_ENOMEM:
        raise MemoryError(msg)
    raise SrxError(f"{what or 'libsrx'} failed ({rc}): {msg}")
