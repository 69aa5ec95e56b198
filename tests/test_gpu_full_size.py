"""The hot path at the benchmark's full sizes, checked through properties that do
not need the oracle to run the whole workload:

* scoring 65,536 x 200 x 39 against 6 x 8 states: a random sample of utterances
  agrees with the numpy oracle; permuting the utterances permutes the scores
  bit for bit; the matrix-core kernel agrees with the shape-generic kernel;
* Baum-Welch statistics of 32,768 x 200 x 39 with 6 x 16 states: the counting
  identities of the statistics record hold (sum of posteriors = frames,
  start counts = utterances, transition counts = frames - utterances, obs sums
  consistent with posteriors), a sample of the per-model sums agrees with the
  oracle run on that model's utterances of a small prefix, and two ranks' worth
  of shards add up to the whole (the data-parallel identity).

Tolerances are the float32 ones of the parity tests (1e-4 relative on
log-likelihoods, 2e-5 on statistics)."""
import numpy as np
import pytest

from conftest import assert_ties_only
from oracle import hmm_oracle as ho

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sra():
    import speech_recognition_amd as s
    return s


@pytest.fixture(scope="module")
def ctx(sra):
    return sra.Context.default()


def _models(rng, M, S, D):
    prms = [ho.random_params(rng, S, D) for _ in range(M)]
    return prms


def _bank(sra, ctx, prms):
    return sra.HMMBank(ctx, np.stack([p.pi for p in prms]), np.stack([p.A for p in prms]),
                       np.stack([p.means for p in prms]), np.stack([p.covars for p in prms]))


def test_scoring_full_size_properties(sra, ctx):
    import torch
    U, T, D, M, S = 65536, 200, 39, 6, 8
    rng = np.random.default_rng(2024)
    prms = _models(rng, M, S, D)
    bank = _bank(sra, ctx, prms)
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    X = torch.randn((U * T, D), device="cuda", dtype=torch.float32, generator=g)
    off = torch.arange(0, (U + 1) * T, T, device="cuda", dtype=torch.int64)
    data = sra.Ragged(ctx, X, offsets=off, dtype="float32")
    sc = bank.score(data).clone()
    assert sc.shape == (U, M) and bool(torch.isfinite(sc).all())
    # (1) a sample against the oracle
    pick = rng.choice(U, size=24, replace=False)
    for u in pick:
        xu = X[u * T:(u + 1) * T].cpu().numpy().astype(np.float64)
        ref = np.array([ho.score(p, xu, [T], loops=False)[0] for p in prms])
        got = sc[u].cpu().numpy()
        assert np.max(np.abs(got - ref) / np.abs(ref)) < 2e-5
    # (2) permuting utterances permutes scores, bit for bit
    perm = torch.randperm(U, device="cuda", generator=g)
    Xp = X.view(U, T, D)[perm].reshape(U * T, D).contiguous()
    scp = bank.score(sra.Ragged(ctx, Xp, offsets=off, dtype="float32"))
    assert torch.equal(scp, sc[perm])
    del Xp
    # (3) matrix-core kernel against the shape-generic kernel on a slice
    n = 2048
    part = sra.Ragged(ctx, X[:n * T], offsets=off[:n + 1], dtype="float32")
    a = bank.score(part).cpu().numpy()
    b = bank.score(part, generic=True).cpu().numpy()
    assert np.max(np.abs(a - b) / np.abs(b)) < 2e-5
    assert np.array_equal(a, sc[:n].cpu().numpy())          # batch position does not matter


def test_estep_full_size_identities(sra, ctx):
    import torch
    U, T, D, M, S = 32768, 200, 39, 6, 16
    rng = np.random.default_rng(77)
    prms = _models(rng, M, S, D)
    bank = _bank(sra, ctx, prms)
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    X = torch.randn((U * T, D), device="cuda", dtype=torch.float32, generator=g)
    lengths = rng.integers(150, T + 1, size=U)               # ragged inside fixed slots
    starts = np.arange(U, dtype=np.int64) * T
    # pack the ragged utterances: keep the first lengths[u] frames of every slot
    keep = torch.from_numpy(np.concatenate([np.arange(s, s + n) for s, n in zip(starts, lengths)])).cuda()
    Xr = X[keep].contiguous()
    del X, keep
    data = sra.Ragged(ctx, Xr, lengths, dtype="float32")
    um = (np.arange(U) % M).astype(np.int32)
    tr = sra.parallel.BaumWelchTrainer(bank, data, um)
    tr.bank.estep(tr.data, tr.utt_model, tr.utt_order, out=tr.stats)
    st = bank.unpack_stats(tr.stats)
    n_frames = np.array([lengths[um == m].sum() for m in range(M)], dtype=np.float64)
    n_utts = np.array([(um == m).sum() for m in range(M)], dtype=np.float64)
    for m in range(M):
        s = st[m]
        assert s["nobs"] == n_utts[m]
        assert np.isfinite(s["logprob"])
        assert abs(s["post"].sum() - n_frames[m]) < 2e-5 * n_frames[m]
        assert abs(s["start"].sum() - n_utts[m]) < 2e-5 * n_utts[m]
        assert abs(s["trans"].sum() - (n_frames[m] - n_utts[m])) < 2e-5 * n_frames[m]
        # rows of the xi sums: out-flow of state i = its posterior mass minus last frames
        assert np.all(s["trans"].sum(axis=1) <= s["post"] * (1 + 1e-4) + 1e-3)
        # weighted second moments dominate squared first moments (Cauchy-Schwarz)
        assert np.all(s["obs**2"] * s["post"][:, None] >= s["obs"] ** 2 * (1 - 1e-4) - 1e-3)
    # a prefix against the oracle: the first 96 utterances on their own
    n = 96
    offs = np.concatenate([[0], np.cumsum(lengths[:n])])
    Xs = Xr[:offs[-1]].cpu().numpy().astype(np.float64)
    small = sra.Ragged(ctx, Xr[:offs[-1]], lengths[:n], dtype="float32")
    got = bank.unpack_stats(bank.estep(small, ctx.tensor(um[:n]), None))
    for m in range(M):
        sel = [u for u in range(n) if um[u] == m]
        Xm = np.vstack([Xs[offs[u]:offs[u + 1]] for u in sel])
        ref, lps = ho.estep(prms[m], Xm, [int(lengths[u]) for u in sel], loops=False)
        np.testing.assert_allclose(got[m]["logprob"], lps.sum(), rtol=1e-5)
        for k in ("start", "trans", "post", "obs", "obs**2"):
            np.testing.assert_allclose(got[m][k], ref[k], rtol=2e-5,
                                       atol=2e-5 * max(1.0, np.abs(ref[k]).max()))
    # data-parallel identity: two shards' statistics add up to the whole
    lo, hi = sra.parallel.shard_utterances(lengths, 0, 2), sra.parallel.shard_utterances(lengths, 1, 2)
    total = np.zeros_like(tr.stats.cpu().numpy())
    off_all = np.concatenate([[0], np.cumsum(lengths)])
    for a, b in (lo, hi):
        shard = sra.Ragged(ctx, Xr[off_all[a]:off_all[b]], lengths[a:b], dtype="float32")
        total += bank.estep(shard, ctx.tensor(um[a:b]), None).cpu().numpy()
    whole = tr.stats.cpu().numpy()
    np.testing.assert_allclose(total, whole, rtol=2e-6, atol=2e-6 * np.abs(whole).max())


def test_viterbi_full_size_properties(sra, ctx):
    """65,536 x 200 x 39, one 8-state model per utterance: the best path can not
    beat the sum over paths; a sample of utterances agrees with the oracle (a
    differing frame must be a near-tie: the path's own log-probability then agrees)."""
    import torch
    U, T, D, M, S = 65536, 200, 39, 6, 8
    rng = np.random.default_rng(31)
    prms = _models(rng, M, S, D)
    bank = _bank(sra, ctx, prms)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    X = torch.randn((U * T, D), device="cuda", dtype=torch.float32, generator=g)
    off = torch.arange(0, (U + 1) * T, T, device="cuda", dtype=torch.int64)
    data = sra.Ragged(ctx, X, offsets=off, dtype="float32")
    um = (np.arange(U) % M).astype(np.int32)
    lp, path = bank.viterbi(data, ctx.tensor(um))
    sc = bank.score(data)
    own = sc[torch.arange(U, device="cuda"), torch.from_numpy(um.astype(np.int64)).cuda()]
    assert bool(torch.isfinite(lp).all())
    assert bool((lp <= own + 1e-6 * own.abs()).all())
    assert int(path.min()) >= 0 and int(path.max()) < S
    pick = rng.choice(U, size=16, replace=False)
    n_diff = 0
    for u in pick:
        xu = X[u * T:(u + 1) * T].cpu().numpy().astype(np.float64)
        rlp, rpath = ho.viterbi(prms[um[u]], xu, [T])
        got = path[u * T:(u + 1) * T].cpu().numpy()
        assert abs(float(lp[u]) - rlp[0]) < 1e-5 * abs(rlp[0])
        pu = prms[um[u]]
        nd, _, _ = assert_ties_only(pu.pi, pu.A, ho.log_density_diag(xu, pu.means, pu.covars), [T],
                                    got, rpath, rlp)
        n_diff += nd
    print(f"full-size viterbi float32: {n_diff} of {16 * T} sampled frames differ, all at ties")


def test_posteriors_full_size_properties(sra, ctx):
    """65,536 x 200 x 39, one 16-state model per utterance: rows of the posteriors sum to
    one, their sums per model are the E-step's occupation counts, the log-probabilities are
    the scores, and a sample of utterances agrees with the oracle."""
    import torch
    U, T, D, M, S = 65536, 200, 39, 6, 16
    rng = np.random.default_rng(61)
    prms = _models(rng, M, S, D)
    bank = _bank(sra, ctx, prms)
    g = torch.Generator(device="cuda")
    g.manual_seed(13)
    X = torch.randn((U * T, D), device="cuda", dtype=torch.float32, generator=g)
    off = torch.arange(0, (U + 1) * T, T, device="cuda", dtype=torch.int64)
    data = sra.Ragged(ctx, X, offsets=off, dtype="float32")
    um = (np.arange(U) % M).astype(np.int32)
    um_d = ctx.tensor(um)
    lp, gam = bank.posteriors(data, um_d)
    assert bool(torch.isfinite(gam).all()) and float(gam.min()) >= 0.0
    assert float((gam.sum(dim=1) - 1).abs().max()) < 1e-5
    sc = bank.score(data)
    own = sc[torch.arange(U, device="cuda"), torch.from_numpy(um.astype(np.int64)).cuda()]
    assert float(((lp - own).abs() / own.abs()).max()) < 2e-5
    st = bank.unpack_stats(bank.estep(data, um_d, None))
    occ = gam.view(U, T, S).sum(dim=1, dtype=torch.float64).cpu().numpy()
    for m in range(M):
        np.testing.assert_allclose(occ[um == m].sum(axis=0), st[m]["post"], rtol=2e-5)
    for u in rng.choice(U, size=8, replace=False):
        xu = X[u * T:(u + 1) * T].cpu().numpy().astype(np.float64)
        rl, rg = ho.score_posteriors(prms[um[u]], xu, [T], loops=False)
        np.testing.assert_allclose(gam[u * T:(u + 1) * T].cpu().numpy(), rg, rtol=0, atol=2e-5)


def test_endpoint_full_size_properties(sra, ctx):
    """8,192 recordings x 3 s at 16 kHz: a loud stretch in every recording is found where it
    was put, permuting the recordings permutes the results, and a recording processed alone
    gives the same blocks and segments as inside the batch."""
    import torch
    R, fs = 8192, 16000
    n = 3 * fs
    g = torch.Generator(device="cuda")
    g.manual_seed(17)
    audio = 0.0005 * torch.randn((R, n), device="cuda", dtype=torch.float32, generator=g)
    starts = torch.randint(20, 150, (R,), device="cuda", generator=g)          # in blocks of 160
    lens = torch.randint(30, 120, (R,), device="cuda", generator=g)
    blk = torch.arange(n, device="cuda") // 160
    loud = (blk[None, :] >= starts[:, None]) & (blk[None, :] < (starts + lens)[:, None])
    audio = torch.where(loud, audio * 400.0, audio)
    so = np.arange(0, (R + 1) * n, n, dtype=np.int64)
    res = sra.endpoint.detect_batch(audio.reshape(-1), so, fs, ctx=ctx)
    cnt = res.counts.cpu().numpy()
    seg = res.segments.cpu().numpy()
    assert np.all(cnt == 1)
    np.testing.assert_array_equal(seg[:, 0, 1], starts.cpu().numpy())
    np.testing.assert_array_equal(seg[:, 0, 2], (starts + lens).cpu().numpy())
    perm = torch.randperm(R, device="cuda", generator=g)
    res_p = sra.endpoint.detect_batch(audio[perm].reshape(-1), so, fs, ctx=ctx)
    # a layout made once gives what plain offsets give
    lay = sra.endpoint.BatchLayout(so, fs, ctx=ctx)
    res_l = sra.endpoint.detect_batch(audio.reshape(-1), lay, fs, ctx=ctx)
    assert torch.equal(res_l.segments, res.segments) and torch.equal(res_l.energy, res.energy)
    with pytest.raises(ValueError):
        sra.endpoint.detect_batch(audio.reshape(-1), lay, 8000, ctx=ctx)
    assert torch.equal(res_p.segments, res.segments[perm])
    assert torch.equal(res_p.energy.view(R, -1), res.energy.view(R, -1)[perm])
    assert torch.equal(res_p.zcr.view(R, -1), res.zcr.view(R, -1)[perm])
    for r in (0, 4097, R - 1):
        one = sra.endpoint.detect_batch(audio[r], [0, n], fs, ctx=ctx)
        assert torch.equal(one.energy, res.energy.view(R, -1)[r])
        assert torch.equal(one.segments[0], res.segments[r])


def test_front_end_beyond_2g_samples(sra, ctx):
    """46,000 recordings x 3 s = 2.2e9 samples in one call (more than the fast kernels
    address at once: the host layer cuts the call into slices of whole utterances):
    recordings either side of the cut give the features they give on their own."""
    import torch
    R, n = 46000, 48000
    audio = torch.empty((R * n,), device="cuda", dtype=torch.float32)
    g = torch.Generator(device="cuda")
    g.manual_seed(23)
    step = 1 << 26
    for i in range(0, R * n, step):
        audio[i:i + step].normal_(generator=g)
    audio *= 0.1
    so = np.arange(0, (R + 1) * n, n, dtype=np.int64)
    plan = sra.MfccPlan(ctx, 16000, nfft=512, n_mel=40, delta_order=2, zscore=True)
    feats, fo = plan.run(audio, so)
    assert feats.shape == (R * 294, 39) and bool(torch.isfinite(feats[::1000]).all())
    cut = int(np.searchsorted(so, (1 << 31) - 512, side="right")) - 1      # first recording of slice 2
    for r in (0, cut - 1, cut, R - 1):
        one, _ = plan.run(audio[r * n:(r + 1) * n], [0, n])
        assert torch.equal(one, feats[fo[r]:fo[r + 1]])
    del feats, audio


def test_front_end_ragged_batch_of_thousands(sra, ctx):
    """3,000 recordings of 0 to 6 s in one float32 call: every workgroup of the delta / z-score
    kernel walks several utterances, with the next one's cepstra requested ahead, through every
    mixture of empty, short (one tile, rows kept in registers) and long (several tiles)
    recordings.  Each checked recording gives exactly the features it gives on its own."""
    import torch
    rng = np.random.default_rng(77)
    R = 3000
    n = rng.integers(0, 6 * 16000, size=R)
    n[rng.integers(0, R, size=40)] = rng.integers(0, 500, size=40)      # shorter than a frame or two
    so = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    audio = torch.empty((int(so[-1]),), device="cuda", dtype=torch.float32).normal_(generator=g) * 0.1
    for order, zscore in ((2, True), (1, True), (0, False)):
        plan = sra.MfccPlan(ctx, 16000, nfft=512, n_mel=40, delta_order=order, zscore=zscore)
        feats, fo = plan.run(audio, so)
        assert bool(torch.isfinite(feats).all())
        picks = list(rng.integers(0, R, size=60)) + [0, 1, R - 2, R - 1]
        for r in picks:
            one, fo1 = plan.run(audio[so[r]:so[r + 1]], [0, int(n[r])])
            assert fo[r + 1] - fo[r] == fo1[1]
            assert torch.equal(one, feats[fo[r]:fo[r + 1]]), (order, zscore, r, int(n[r]))


def test_addressing_beyond_4gib(sra, ctx):
    """262,144 utterances x 200 x 39 floats = 7.6 GiB of features: results for
    utterances past the 4 GiB mark equal those of the same data processed on its own
    (scores bit for bit, statistics to rounding of the float64 sums)."""
    import torch
    U, T, D, M = 1 << 18, 200, 39, 6
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    X = torch.empty((U * T, D), device="cuda", dtype=torch.float32)
    step = 1 << 24
    for i in range(0, U * T, step):
        X[i:i + step].normal_(generator=g)
    off = torch.arange(0, (U + 1) * T, T, device="cuda", dtype=torch.int64)
    data = sra.Ragged(ctx, X, offsets=off, dtype="float32")
    rng = np.random.default_rng(12)
    bank8 = _bank(sra, ctx, _models(rng, M, 8, D))
    bank16 = _bank(sra, ctx, _models(rng, M, 16, D))
    n = 2048
    tail = sra.Ragged(ctx, X[(U - n) * T:], offsets=off[:n + 1], dtype="float32")
    head = sra.Ragged(ctx, X[:(U - n) * T], offsets=off[:U - n + 1], dtype="float32")
    sc = bank8.score(data)
    assert torch.equal(sc[U - n:], bank8.score(tail))
    um = (np.arange(U) % M).astype(np.int32)
    s_all = bank16.estep(data, ctx.tensor(um), None).cpu().numpy()
    s_head = bank16.estep(head, ctx.tensor(um[:U - n]), None).cpu().numpy()
    s_tail = bank16.estep(tail, ctx.tensor(um[U - n:]), None).cpu().numpy()
    assert np.abs(s_all - (s_head + s_tail)).max() < 1e-9 * np.abs(s_all).max()
    lp, path = bank8.viterbi(data, ctx.tensor(um))
    lp_t, path_t = bank8.viterbi(tail, ctx.tensor(um[U - n:]))
    assert torch.equal(lp[U - n:], lp_t) and torch.equal(path[(U - n) * T:], path_t)
    ll = bank8.emission_loglik(tail)
    ll_all_tail = bank8.emission_loglik(data)[(U - n) * T:]
    assert torch.equal(ll, ll_all_tail)


def test_score_batch_spanning_4gib(sra, ctx):
    """A batch of 16 consecutive utterances whose frames span more than 4 GiB (13 of
    1.75 M frames x 48 floats and 3 short ones) is out of the matrix-core kernel's reach and
    takes the far-batch kernel: its short utterances agree with the oracle, two long ones
    with the same data scored on its own (a batch of one, matrix-core kernel), and the
    ordinary batch that follows in the same call with its own call."""
    import torch
    D, M, S, L = 48, 2, 5, 1_750_000
    lengths = np.array([200, L, L, L, L, 150, L, L, L, L, L, 1, L, L, L, L] + [300, 250, 1, 40, 77], dtype=np.int64)
    assert int(lengths[:16].sum()) * D * 4 >= 1 << 32
    n = int(lengths.sum())
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    X = torch.empty((n, D), device="cuda", dtype=torch.float32)
    for i in range(0, n, 1 << 22):
        X[i:i + (1 << 22)].normal_(generator=g)
    rng = np.random.default_rng(31)
    prms = _models(rng, M, S, D)
    bank = _bank(sra, ctx, prms)
    off = np.concatenate([[0], np.cumsum(lengths)])
    sc = bank.score(sra.Ragged(ctx, X, lengths, dtype="float32")).cpu().numpy()
    assert sc.shape == (len(lengths), M) and np.isfinite(sc).all()
    for u in (0, 5, 11):                                   # short utterances of the far batch: oracle
        x = X[off[u]:off[u + 1]].cpu().numpy().astype(np.float64)
        for m in range(M):
            want = ho.score(prms[m], x)
            assert abs(sc[u, m] - want) <= 1e-9 * abs(want) + 1e-9
    for u in (1, 15):                                      # long ones: the same frames in a batch of one
        alone = bank.score(sra.Ragged(ctx, X[off[u]:off[u + 1]], [int(lengths[u])], dtype="float32")).cpu().numpy()
        assert np.abs(sc[u] - alone[0]).max() <= 1e-5 * np.abs(alone).max()
    tail = bank.score(sra.Ragged(ctx, X[off[16]:], lengths[16:], dtype="float32")).cpu().numpy()
    assert np.array_equal(sc[16:], tail)
    del X


def test_training_config_full_size(sra, ctx):
    """The training configuration at its full size on one GPU: 1,048,576 utterances x 200
    frames x 39 dims (31 GB of features), 6 models x 16 states.

    * counting identities of every model's statistics record;
    * the first 96 utterances agree with the oracle's E-step (per model, 2e-5);
    * addressing beyond 4 GiB: scores of the last 4096 utterances equal those of the same
      utterances scored on their own, bit for bit, and the statistics of (everything but the
      tail) + (the tail) add up to the statistics of the whole (the data-parallel identity
      across the 4 GiB boundary)."""
    import torch
    U, T, D, M, S = 1 << 20, 200, 39, 6, 16
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2 ** 30:
        pytest.skip("needs 60 GiB of free device memory")
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    X = torch.empty((U * T, D), device="cuda", dtype=torch.float32)
    for i in range(0, U * T, 1 << 24):
        X[i:i + (1 << 24)].normal_(generator=g)
    off = torch.arange(0, (U + 1) * T, T, device="cuda", dtype=torch.int64)
    data = sra.Ragged(ctx, X, offsets=off, dtype="float32")
    rng = np.random.default_rng(8)
    prms = _models(rng, M, S, D)
    bank = _bank(sra, ctx, prms)
    um = (np.arange(U) % M).astype(np.int32)
    umd = ctx.tensor(um)
    order = ctx.tensor(np.argsort(um, kind="stable").astype(np.int64))
    s_all = bank.estep(data, umd, order).clone()
    st = bank.unpack_stats(s_all)
    for m in range(M):
        n_u = int((um == m).sum())
        n_fr = float(n_u) * T
        assert st[m]["nobs"] == n_u
        assert abs(st[m]["start"].sum() - n_u) < 2e-5 * n_u
        assert abs(st[m]["post"].sum() - n_fr) < 2e-5 * n_fr
        assert abs(st[m]["trans"].sum() - (n_fr - n_u)) < 2e-5 * n_fr
        assert np.isfinite(st[m]["logprob"])
    # oracle prefix
    n0 = 96
    Xh = X[:n0 * T].cpu().numpy().astype(np.float64)
    head = sra.Ragged(ctx, X[:n0 * T], offsets=off[:n0 + 1], dtype="float32")
    sh = bank.unpack_stats(bank.estep(head, ctx.tensor(um[:n0]), None))
    for m in range(M):
        idx = [u for u in range(n0) if um[u] == m]
        Xm = np.vstack([Xh[u * T:(u + 1) * T] for u in idx])
        ref, lps = ho.estep(prms[m], Xm, [T] * len(idx), loops=False)
        assert abs(sh[m]["logprob"] - lps.sum()) < 1e-4 * abs(lps.sum())
        for key in ("start", "trans", "post", "obs", "obs**2"):
            scale = np.abs(ref[key]).max()
            assert np.max(np.abs(sh[m][key] - ref[key])) < 2e-5 * scale, (m, key)
    # the far end of the buffer (beyond byte 4 GiB)
    n = 4096
    tail = sra.Ragged(ctx, X[(U - n) * T:], offsets=off[:n + 1], dtype="float32")
    rest = sra.Ragged(ctx, X[:(U - n) * T], offsets=off[:U - n + 1], dtype="float32")
    prm8 = _models(rng, M, 8, D)
    bank8 = _bank(sra, ctx, prm8)
    sc_all = bank8.score(data)
    sc_tail = bank8.score(tail)
    assert bool(torch.isfinite(sc_all).all())
    assert torch.equal(sc_all[U - n:], sc_tail), "scores of the far end differ"
    ref_sc = np.stack([ho.score(p, X[(U - 3) * T:].cpu().numpy().astype(np.float64), [T] * 3, loops=False)
                       for p in prm8], axis=1)
    assert np.max(np.abs(sc_all[U - 3:].cpu().numpy() - ref_sc) / np.abs(ref_sc)) < 1e-4
    s_rest = bank.estep(rest, ctx.tensor(um[:U - n]), None).cpu().numpy()
    s_tail = bank.estep(tail, ctx.tensor(um[U - n:]), None).cpu().numpy()
    s_all2 = bank.estep(data, umd, None).cpu().numpy()
    err = np.abs(s_all2 - (s_rest + s_tail)).max() / np.abs(s_all2).max()
    assert err < 1e-6, err
    del X, data, tail, rest
    torch.cuda.empty_cache()
