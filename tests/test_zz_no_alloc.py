"""srx_forward_score_static (the allocation-free scoring entry point) against
srx_forward_score, through the C ABI: same kernels, so the scores must be bit-equal.
Two calls in a row share the fixed table array of the GPU; the second, smaller one must
not see anything the first left there."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bank(sra, ctx, rng, M, S, D):
    return sra.HMMBank(ctx, rng.dirichlet(np.ones(S), size=M), rng.dirichlet(np.ones(S), size=(M, S)),
                       rng.standard_normal((M, S, D)), rng.uniform(0.5, 1.5, (M, S, D)))


def test_static_entry_point_matches_forward_score():
    import torch
    import speech_recognition_amd as sra
    from speech_recognition_amd import _lib
    from oracle import hmm_oracle as ho

    assert int(_lib.lib.srx_static_scratch_bytes()) == 256 * 1024
    rng = np.random.default_rng(42)
    ctx_ref = sra.Context.default()
    ctx = sra.Context()                       # a fresh context: its workspace must stay empty
    first = True
    for M, S, D, lengths in [(6, 8, 39, [200, 57, 1, 120, 33, 200, 2, 90] * 6),
                             (3, 3, 26, [40, 17, 5])]:
        X = rng.standard_normal((sum(lengths), D))
        bank = _bank(sra, ctx, np.random.default_rng(7 + S), M, S, D)
        bank_ref = _bank(sra, ctx_ref, np.random.default_rng(7 + S), M, S, D)
        data = sra.Ragged(ctx, X, lengths, dtype="float32")
        data_ref = sra.Ragged(ctx_ref, X, lengths, dtype="float32")
        want = bank_ref.score(data_ref).cpu().numpy()
        out = torch.full((data.U, M), float("nan"), dtype=torch.float64, device=ctx.device)
        got = bank.score_static(data, out).cpu().numpy()
        assert np.array_equal(got, want), np.max(np.abs(got - want))
        if first:                             # and both agree with the oracle to the float32 bar
            sp, A, mu, cv = bank.params_host()
            ref = np.stack([ho.score(ho.Params(sp[m], A[m], mu[m], cv[m]), X, lengths, loops=False)
                            for m in range(M)], axis=1)
            assert np.max(np.abs(got - ref) / np.abs(ref)) < 1e-4
            first = False
    assert ctx.workspace_bytes() == 0, "the allocation-free entry point grew the workspace"

    # a bank whose tables do not fit the fixed array is rejected with the sizes in the message
    big = _bank(sra, ctx, rng, 100, 8, 39)
    data = sra.Ragged(ctx, rng.standard_normal((10, 39)), [10], dtype="float32")
    out = torch.empty((1, 100), dtype=torch.float64, device=ctx.device)
    with pytest.raises(ValueError, match="too large"):
        big.score_static(data, out)
    # float64 has no allocation-free kernel
    d64 = sra.Ragged(ctx, rng.standard_normal((10, 39)), [10], dtype="float64")
    small = _bank(sra, ctx, rng, 2, 8, 39)
    with pytest.raises(sra._lib.SrxError):
        small.score_static(d64, torch.empty((1, 2), dtype=torch.float64, device=ctx.device))
    ctx.close()


def test_static_entry_point_on_two_streams():
    """Two streams of one GPU share the fixed table array: calls that alternate between them,
    each with another bank, are put in order by the library (the later stream waits for the
    earlier call's kernels), so every call's scores are those of srx_forward_score."""
    import torch
    import speech_recognition_amd as sra

    rng = np.random.default_rng(3)
    ctx_ref = sra.Context.default()
    D, U, T = 39, 4096, 100
    X = rng.standard_normal((U * T, D)).astype(np.float32)
    lengths = [T] * U
    shapes = [(6, 8), (12, 8), (3, 16), (10, 3)]
    ref_data = sra.Ragged(ctx_ref, X, lengths, dtype="float32")
    want = []
    for k, (M, S) in enumerate(shapes):
        want.append(_bank(sra, ctx_ref, np.random.default_rng(50 + k), M, S, D).score(ref_data).cpu().numpy())
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ctxs, datas, banks, outs = [], [], [], []
    for s in streams:
        with torch.cuda.stream(s):
            ctxs.append(sra.Context())
            datas.append(sra.Ragged(ctxs[-1], X, lengths, dtype="float32"))
    for k, (M, S) in enumerate(shapes):
        c = ctxs[k % 2]
        banks.append(_bank(sra, c, np.random.default_rng(50 + k), M, S, D))
        outs.append(torch.full((U, M), float("nan"), dtype=torch.float64, device=c.device))
    torch.cuda.synchronize()
    for rep in range(3):                       # queued back to back, nothing waits in between
        for k in range(len(shapes)):
            with torch.cuda.stream(streams[k % 2]):
                banks[k].score_static(datas[k % 2], outs[k])
    torch.cuda.synchronize()
    for k in range(len(shapes)):
        assert np.array_equal(outs[k].cpu().numpy(), want[k]), k
    for c in ctxs:
        assert c.workspace_bytes() == 0
        c.close()
