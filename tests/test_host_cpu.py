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


def test_bench_launcher_starts_one_rank_per_gpu():
    """`bench.py --gpus 2` without RANK in the environment launches two ranks before any
    GPU call (checked with --dry-run-launch: the children print their rank environment
    and never import torch.cuda)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--dry-run-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    recs = sorted((json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")),
                  key=lambda d: d["RANK"])
    assert [d["RANK"] for d in recs] == ["0", "1"]
    assert [d["LOCAL_RANK"] for d in recs] == ["0", "1"]
    assert all(d["WORLD_SIZE"] == "2" and d["MASTER_ADDR"] == "127.0.0.1" for d in recs)
    assert recs[0]["MASTER_PORT"] == recs[1]["MASTER_PORT"]
    # under torch.distributed.run the ranks exist already: nothing more is launched
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--dry-run-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1


def test_two_rank_bench_record():
    """The kept line of `bench.py --gpus 2 --share-gpu --backend gloo` (two ranks sharing the
    one GPU of a test box, profiles/r3_two_rank_rehearsal_gloo.json): the N > 1 branch ran, its
    line carries the training metric with the rank count an all-reduce returned, and the
    all-reduced statistics of the two shards equal one rank's E-step over all utterances."""
    import json
    with open(os.path.join(ROOT, "profiles", "r3_two_rank_rehearsal_gloo.json")) as f:
        d = json.load(f)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["collective_backend"] == "gloo"
    assert d["metric"] == "hmm_scored_frames_per_sec_6_models" and d["scaling"] == "weak"
    for key in ("baum_welch_iter_per_s", "allreduce_ms_per_iter", "value", "ms_per_step"):
        assert isinstance(d[key], float) and d[key] > 0, key
    assert d["extra"]["baum_welch"]["rccl_ranks"] == 2
    assert 0 <= d["sharded_vs_single_rank_stats_max_rel_err"] < 1e-5


class _OracleBank:
    """Stand-in for HMMBank on a machine without a GPU: the same methods the trainer calls,
    with the CPU oracle in place of the kernels and CPU tensors in place of device memory."""

    class _Ctx:
        def __init__(self):
            import torch
            self.device = torch.device("cpu")

        def tensor(self, a, dtype=None):
            import torch
            return torch.from_numpy(np.ascontiguousarray(a))

    def __init__(self, params, X, lengths):
        from oracle import hmm_oracle as ho
        self.ho, self.ctx = ho, self._Ctx()
        self.params = list(params)
        self.M = len(params)
        self.S, self.D = params[0].means.shape
        self.stats_size = 2 + self.S + self.S * self.S + self.S + 2 * self.S * self.D
        self.X, self.lengths = X, np.asarray(lengths)

    def estep(self, data, utt_model, utt_order, out=None, stats="all"):
        assert stats in ("all", "transitions")     # (the oracle-backed E-step computes everything)
        ho = self.ho
        off = np.concatenate([[0], np.cumsum(self.lengths)])
        um = utt_model.numpy()
        for m in range(self.M):
            idx = [u for u in range(len(self.lengths)) if um[u] == m]
            rec = np.zeros(self.stats_size)
            if idx:
                Xm = np.vstack([self.X[off[u]:off[u + 1]] for u in idx])
                st, lps = ho.estep(self.params[m], Xm, self.lengths[idx], loops=False)
                rec = np.concatenate([[st["nobs"], lps.sum()], st["start"], st["trans"].ravel(),
                                      st["post"], st["obs"].ravel(), st["obs**2"].ravel()])
            out[m] = __import__("torch").from_numpy(rec)
        return out

    def unpack(self, rec):
        S, D = self.S, self.D
        o = 2
        d = {"nobs": rec[0]}
        d["start"] = rec[o:o + S]; o += S
        d["trans"] = rec[o:o + S * S].reshape(S, S); o += S * S
        d["post"] = rec[o:o + S]; o += S
        d["obs"] = rec[o:o + S * D].reshape(S, D); o += S * D
        d["obs**2"] = rec[o:o + S * D].reshape(S, D)
        return d

    def mstep(self, stats, priors, mode):
        f = self.ho.mstep_full if mode == "full" else self.ho.mstep_transitions
        self.params = [f(p, self.unpack(stats[m].numpy())) for m, p in enumerate(self.params)]


class _Shard:
    def __init__(self, X, lengths):
        import torch
        self.X, self.U = torch.from_numpy(X), len(lengths)


def _trainer_rank_main(rank, world, port, ret):
    """Three EM iterations of BaumWelchTrainer.step on this rank's shard: E-step, the
    package's all-reduce, replicated M-step.  Every rank must end with the parameters of a
    single-process run over all utterances, and report the global log-probabilities."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import hmm_oracle as ho
    from speech_recognition_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(11)
    S, D, M = 3, 5, 2
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    lengths = rng.integers(1, 30, size=31)
    um = (np.arange(len(lengths)) % M).astype(np.int32)
    X = rng.standard_normal((int(lengths.sum()), D))
    off = np.concatenate([[0], np.cumsum(lengths)])
    lo, hi = parallel.shard_utterances(lengths, rank, world)
    Xs, Ls = X[off[lo]:off[hi]], lengths[lo:hi]
    bank = _OracleBank(prms, Xs, Ls)
    tr = parallel.BaumWelchTrainer(bank, _Shard(Xs, Ls), um[lo:hi], mstep="full")
    single = _OracleBank(prms, X, lengths)
    ok = True
    for it in range(3):
        lp = tr.step().numpy()
        import torch
        st = torch.zeros((M, single.stats_size), dtype=torch.float64)
        single.estep(None, torch.from_numpy(um), None, out=st)
        lp1 = st[:, 1].numpy().copy()
        single.mstep(st, None, "full")
        ok = ok and np.allclose(lp, lp1, rtol=1e-10)
        for a, b in zip(bank.params, single.params):
            ok = ok and np.allclose(a.A, b.A, rtol=1e-9, atol=1e-12) \
                and np.allclose(a.means, b.means, rtol=1e-9, atol=1e-12) \
                and np.allclose(a.covars, b.covars, rtol=1e-9, atol=1e-12) \
                and np.allclose(a.pi, b.pi, rtol=1e-9, atol=1e-12)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_two_rank_trainer_step_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_rank_main, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1)


def test_hmm_lib_resamples_on_load(tmp_path):
    """HMM_lib.py:12 loads with librosa.load(path, sr=sample_rate): a file stored at another rate
    comes back at the requested one, ceil(n * ratio) samples long, with its tone where it was."""
    from scipy.io import wavfile
    import speech_recognition_amd as sra
    n, sr = 16000, 16000
    t = np.arange(n) / sr
    y = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
    wavfile.write(str(tmp_path / "tone.wav"), sr, y)
    out = sra.hmm_lib._load(str(tmp_path / "tone"), 22050)
    assert out.dtype == np.float32 and len(out) == -(-n * 22050 // sr)
    spec = np.abs(np.fft.rfft(out * np.hanning(len(out))))
    assert abs(np.argmax(spec) * 22050.0 / len(out) - 1000.0) < 2.0
    assert abs(np.abs(out[2000:-2000]).max() - 0.5) < 5e-3
    same = sra.hmm_lib._load(str(tmp_path / "tone"), 16000)
    assert np.array_equal(same, y)
