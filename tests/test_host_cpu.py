"""CPU-only checks: the C-ABI library loads and exports what include/srx.h
declares, host-side logic (offsets, sharding, error mapping), and the
multi-rank statistics path over gloo with world_size 2."""
import ctypes
import os
import re
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "srx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import speech_recognition_amd as sra
    names = _declared_symbols()
    assert len(names) >= 17
    lib = ctypes.CDLL(sra._lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/srx.h but not exported"
    assert set(names) == set(sra._lib.SIGNATURES), "ctypes table and header disagree"
    assert sra._lib.lib.srx_version() >= 100
    assert sra._lib.lib.srx_stats_size(3, 26) == 2 + 3 + 9 + 3 + 2 * 78


def test_host_only_entry_points():
    import speech_recognition_amd as sra
    cfg = sra._lib.MfccCfg(22050, 551, 220, 1024, 26, 16000, 13, 1, 2, 1, 20.0)
    f = sra._lib.lib.srx_mfcc_out_frames
    assert f(ctypes.byref(cfg), 20727) == 92 - 4          # odessa_5.wav: 92 raw frames
    assert f(ctypes.byref(cfg), 550) == 0
    assert f(ctypes.byref(cfg), 551 + 3 * 220) == 0       # 4 raw frames -> nothing left after trim
    assert f(ctypes.byref(cfg), 551 + 4 * 220) == 1
    fb = sra.mfcc.mel_filterbank(1024)
    assert fb.shape == (26, 513) and np.count_nonzero(fb) == 944
    bad = sra._lib.MfccCfg(16000, 400, 160, 1000, 26, 16000, 13, 1, 2, 1, 20.0)
    out = np.zeros((26, 501))
    rc = sra._lib.lib.srx_mfcc_filterbank_host(ctypes.byref(bad), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == sra._lib.SRX_EINVAL
    assert b"power of two" in sra._lib.lib.srx_last_error()
    with pytest.raises(ValueError):
        sra._lib.check(rc)


def test_no_gpu_fails_loudly():
    import torch
    import speech_recognition_amd as sra
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sra.Context.default()
    m = sra.MyGaussianHMM(n_components=2, verbose=False)
    m.startprob_ = np.array([0.5, 0.5]); m.transmat_ = np.full((2, 2), 0.5)
    m.means_ = np.zeros((2, 3)); m._covars_ = np.ones((2, 3))
    with pytest.raises(RuntimeError):
        m.score(np.zeros((4, 3)))


def test_offsets_and_shards():
    from speech_recognition_amd.runtime import offsets_from_lengths
    from speech_recognition_amd.parallel import shard_bounds, shard_utterances
    np.testing.assert_array_equal(offsets_from_lengths([3, 0, 2]), [0, 3, 3, 5])
    with pytest.raises(ValueError, match="doesn't sum"):
        offsets_from_lengths([3, 2], total=6)
    rng = np.random.default_rng(0)
    L = rng.integers(1, 300, size=1000)
    for w in (1, 2, 3, 8):
        b = shard_bounds(L, w)
        assert b[0] == 0 and b[-1] == len(L) and np.all(np.diff(b) >= 0)
        frames = [L[b[r]:b[r + 1]].sum() for r in range(w)]
        assert max(frames) - min(frames) <= 2 * L.max()
        assert sum(frames) == L.sum()
    assert shard_utterances([5, 5, 5, 5], 1, 2) == (2, 4)
    assert shard_bounds([7], 4)[-1] == 1


def test_monitor_semantics(capsys):
    from speech_recognition_amd.hmm import ConvergenceMonitor
    m = ConvergenceMonitor(tol=1e-2, n_iter=5, verbose=True)
    m.report(-10.0)
    assert not m.converged
    m.report(-9.0)
    assert not m.converged
    m.report(-8.995)
    assert m.converged and m.iter == 3
    err = capsys.readouterr().err.splitlines()
    assert err[1].split() == ["2", "-9.00000000", "+1.00000000"]
    m2 = ConvergenceMonitor(tol=0.0, n_iter=2, verbose=False)
    m2.report(-1.0); m2.report(-0.5)
    assert m2.converged        # iteration budget used up


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, ret):
    """Each rank computes Baum-Welch statistics of its utterance shard (with the
    CPU oracle standing in for the kernel) and the package's collective sums
    them; the replicated M-step must then agree with the single-process one."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import hmm_oracle as ho
    from speech_recognition_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(3)
    S, D = 4, 6
    prm = ho.random_params(rng, S, D)
    lengths = rng.integers(1, 40, size=23)
    X = rng.standard_normal((int(lengths.sum()), D))
    lo, hi = parallel.shard_utterances(lengths, rank, world)
    off = np.concatenate([[0], np.cumsum(lengths)])
    st, lps = ho.estep(prm, X[off[lo]:off[hi]], lengths[lo:hi], loops=False)
    rec = np.concatenate([[st["nobs"], lps.sum()], st["start"], st["trans"].ravel(), st["post"],
                          st["obs"].ravel(), st["obs**2"].ravel()])
    t = torch.from_numpy(rec[None].copy())
    parallel.allreduce_stats(t)
    full, lp_full = ho.estep(prm, X, lengths, loops=False)
    ref = np.concatenate([[full["nobs"], lp_full.sum()], full["start"], full["trans"].ravel(),
                          full["post"], full["obs"].ravel(), full["obs**2"].ravel()])
    ok = np.allclose(t.numpy()[0], ref, rtol=1e-10, atol=1e-12) and parallel.world() == (rank, world)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_two_rank_stats_allreduce_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1)
