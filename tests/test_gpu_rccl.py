"""srx_allreduce_stats through the C ABI on a one-rank RCCL communicator (a GPU box has one
card; RCCL refuses two ranks on one device).  What this pins: the library finds
ncclAllReduce in the RCCL copy already in the process, passes communicator, buffer, count
and stream in the right places (RCCL rejects anything else), works in place and corrupts
nothing.  A sum over one rank is the identity for any element type and operator, so the
dtype / operator codes are NOT checked here: they are the enumerators of rccl/rccl.h at
build time.  The two-rank arithmetic is covered on CPU (gloo) in tests/test_host_cpu.py."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]       # ncclUniqueId, nccl.h


def test_allreduce_stats_one_rank_identity():
    import torch
    import speech_recognition_amd as sra
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    ctx = sra.Context.default()
    torch.cuda.set_device(ctx.index)
    uid = _UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        M, S, D = 6, 16, 39
        size = int(sra._lib.lib.srx_stats_size(S, D))
        rng = np.random.default_rng(2)
        host = rng.standard_normal((M, size))
        stats = ctx.tensor(host)
        out = sra.parallel.allreduce_stats_rccl(ctx, comm.value, stats)
        torch.cuda.synchronize()
        assert out.data_ptr() == stats.data_ptr()
        np.testing.assert_array_equal(stats.cpu().numpy(), host)
        # argument checks: no communicator -> ValueError, empty message is a no-op
        with pytest.raises(ValueError):
            sra.parallel.allreduce_stats_rccl(ctx, 0, stats)
        sra.parallel.allreduce_stats_rccl(ctx, comm.value, stats[:0])
    finally:
        rccl.ncclCommDestroy(comm)
