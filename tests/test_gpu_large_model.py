"""Large emission model (mixtures of diagonal Gaussians per state) and Viterbi over
a dense transition matrix, through the C ABI (srx_gmm_emission_loglik,
srx_viterbi_dense), against the numpy oracle.

Pinned to the reference where it has a counterpart: one component of weight 1 is
its diagonal density (golden loglik), and utils.viterbi on the golden log-densities
gives the golden paths.  Mixtures proper have no counterpart in the reference
(parity unpinned): they are checked against the oracle's restatement only.
Tolerances: float64 1e-10; float32 (split-bf16 matrix-core path) 1e-4 relative on
a log-density, state paths identical except at near-ties."""
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


def _rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


@pytest.mark.parametrize("case", ["s3d26", "s8d39", "s16d39"])
def test_single_component_is_reference_density(sra, ctx, hmm_golden, case):
    g, p = hmm_golden, case + "_"
    mu, var = g[p + "mu"], g[p + "var"]
    em = sra.GmmEmission(ctx, np.ones((mu.shape[0], 1)), mu[:, None, :], var[:, None, :])
    for dtype, tol in (("float64", 1e-11), ("float32", 2e-5)):
        data = sra.Ragged(ctx, g[p + "X"], g[p + "lengths"], dtype=dtype)
        ll = em.loglik(data).cpu().numpy()
        np.testing.assert_allclose(ll, g[p + "loglik"], rtol=tol, atol=tol * 10)


@pytest.mark.parametrize("S,MX,D,T", [(5, 3, 7, 100), (40, 16, 39, 333), (33, 9, 13, 257), (7, 20, 39, 64),
                                      (6, 4, 20, 50), (9, 5, 45, 70), (12, 16, 26, 90)])
def test_mixture_density_matches_oracle(sra, ctx, S, MX, D, T):
    rng = np.random.default_rng(S * 100 + MX)
    w = rng.dirichlet(np.ones(MX), size=S)
    w[0, 0] = 0.0                                  # a structurally absent component
    w[0] /= w[0].sum() if MX > 1 else 1.0
    if MX == 1:
        w[:] = 1.0
    mu = rng.standard_normal((S, MX, D)) * 1.5
    var = rng.uniform(0.3, 2.0, (S, MX, D))
    X = rng.standard_normal((T, D)) * 1.2
    ref = ho.gmm_log_density(X, w, mu, var)
    em = sra.GmmEmission(ctx, w, mu, var)
    for dtype, tol in (("float64", 1e-10), ("float32", 1e-4)):
        ll = em.loglik(sra.Ragged(ctx, X, [T], dtype=dtype)).cpu().numpy()
        assert ll.shape == (T, S)
        assert _rel(ll, ref) < tol
    # exact-f32 option takes the shape-generic arithmetic
    ctx.set_exact_f32(True)
    try:
        ll = em.loglik(sra.Ragged(ctx, X, [T], dtype="float32")).cpu().numpy()
    finally:
        ctx.set_exact_f32(False)
    assert _rel(ll, ref) < 2e-5


@pytest.mark.parametrize("case", ["s3d26", "s8d39", "s16d39", "s5d7lr"])
def test_dense_viterbi_matches_reference(sra, ctx, hmm_golden, case):
    g, p = hmm_golden, case + "_"
    ll = g[p + "loglik"]
    off = ctx.tensor(np.concatenate([[0], np.cumsum(g[p + "lengths"])]).astype(np.int64))
    lp, path = sra.viterbi_dense(ctx, ctx.tensor(ll), off, g[p + "pi"], g[p + "A"])
    np.testing.assert_array_equal(path.cpu().numpy(), g[p + "vit_path"])
    np.testing.assert_allclose(lp.cpu().numpy(), g[p + "vit_logprob"], rtol=1e-12)
    lp32, path32 = sra.viterbi_dense(ctx, ctx.tensor(ll.astype(np.float32)), off, g[p + "pi"], g[p + "A"])
    assert_ties_only(g[p + "pi"], g[p + "A"], ll, g[p + "lengths"], path32.cpu().numpy(),
                     g[p + "vit_path"], g[p + "vit_logprob"])
    assert _rel(lp32.cpu().numpy(), g[p + "vit_logprob"]) < 1e-5


@pytest.mark.parametrize("S", [64, 200, 1024])
def test_dense_viterbi_large_states(sra, ctx, S):
    rng = np.random.default_rng(S)
    lengths = [1, 2, 37, 50, 3, 64, 19] + [40] * 14          # more than one workgroup of utterances
    Tt = sum(lengths)
    ll = rng.standard_normal((Tt, S)) * 3.0 - 50.0
    A = rng.dirichlet(np.ones(S) * 0.2, size=S)
    A[A < 1e-4] = 0.0                                         # structural zeros
    A /= A.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(S))
    rlp, rpath = ho.viterbi_dense(pi, A, ll, lengths)
    off = ctx.tensor(np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64))
    lp, path = sra.viterbi_dense(ctx, ctx.tensor(ll), off, pi, A)
    np.testing.assert_array_equal(path.cpu().numpy(), rpath)
    np.testing.assert_allclose(lp.cpu().numpy(), rlp, rtol=1e-12)
    lp32, path32 = sra.viterbi_dense(ctx, ctx.tensor(ll.astype(np.float32)), off, pi, A)
    assert_ties_only(pi, A, ll, lengths, path32.cpu().numpy(), rpath, rlp)
    assert _rel(lp32.cpu().numpy(), rlp) < 1e-5


def test_dense_viterbi_float32_long_utterances(sra, ctx):
    """Utterances of thousands of frames through the dense float32 Viterbi: the lattice is
    made relative to its maximum every 64 steps, so paths part from the float64 ones only at
    near-ties (1e-6 of the score) and the log-probability keeps 1e-6 relative."""
    rng = np.random.default_rng(77)
    S = 40
    lengths = [5000, 1, 2049, 33, 17, 3000] + [16] * 12
    ll = (rng.standard_normal((sum(lengths), S)) * 3.0 - 50.0).astype(np.float32).astype(np.float64)
    A = rng.dirichlet(np.ones(S) * 0.3, size=S)
    pi = rng.dirichlet(np.ones(S))
    rlp, rpath = ho.viterbi_dense(pi, A, ll, lengths)
    off = ctx.tensor(np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64))
    lp32, path32 = sra.viterbi_dense(ctx, ctx.tensor(ll.astype(np.float32)), off, pi, A)
    n_diff, gap, short = assert_ties_only(pi, A, ll, lengths, path32.cpu().numpy(), rpath, rlp, eps=1e-6)
    print(f"dense viterbi float32, {sum(lengths)} frames: {n_diff} differ, largest tie gap {gap:.2e}")
    assert n_diff <= 40
    assert _rel(lp32.cpu().numpy(), rlp) < 1e-6


def test_large_model_errors(sra, ctx):
    with pytest.raises(ValueError):
        sra.GmmEmission(ctx, np.ones((2, 2)), np.zeros((2, 3, 4)), np.ones((2, 3, 4)))
    with pytest.raises(ValueError):
        sra.GmmEmission(ctx, np.ones((2, 3)), np.zeros((2, 3, 4)), -np.ones((2, 3, 4)))
    em = sra.GmmEmission(ctx, np.full((2, 3), 1 / 3), np.zeros((2, 3, 4)), np.ones((2, 3, 4)))
    with pytest.raises(ValueError):
        em.loglik(sra.Ragged(ctx, np.zeros((5, 6)), [5], dtype="float32"))
    off = ctx.tensor(np.array([0, 5], dtype=np.int64))
    with pytest.raises(ValueError):
        sra.viterbi_dense(ctx, ctx.tensor(np.zeros((5, 3))), off, np.ones(4) / 4, np.ones((3, 3)) / 3)
    with pytest.raises(ValueError):                            # more states than the kernel holds
        S = 1025
        sra.viterbi_dense(ctx, ctx.tensor(np.zeros((2, S), dtype=np.float32)),
                          ctx.tensor(np.array([0, 2], dtype=np.int64)), np.ones(S) / S, np.ones((S, S)) / S)


def test_large_model_config_full_shape(sra, ctx):
    """The large-model configuration at its full shape: 1024 states x 16 diagonal components
    x 39 dims.  Log-densities of 16 utterances x 40 frames against the oracle's mixture
    density (float64 1e-10, float32 1e-4 relative), then Viterbi over a dense 1024 x 1024
    transition matrix fed from those log-densities: float64 paths equal the oracle's
    (utils.viterbi restated) bit for bit, float32 paths differ at ties only."""
    S, MX, D, T, U = 1024, 16, 39, 40, 16
    rng = np.random.default_rng(1024)
    w = rng.dirichlet(np.ones(MX), size=S)
    mu = rng.standard_normal((S, MX, D))
    var = rng.uniform(0.5, 1.5, (S, MX, D))
    X = rng.standard_normal((U * T, D))
    ref = ho.gmm_log_density(X, w, mu, var)
    em = sra.GmmEmission(ctx, w, mu, var)
    lengths = [T] * U
    ll64 = em.loglik(sra.Ragged(ctx, X, lengths, dtype="float64"))
    ll32 = em.loglik(sra.Ragged(ctx, X, lengths, dtype="float32"))
    e64, e32 = _rel(ll64.cpu().numpy(), ref), _rel(ll32.cpu().numpy(), ref)
    print(f"1024 x 16 x 39 mixture log-densities: float64 {e64:.2e}, float32 {e32:.2e} relative")
    assert e64 < 1e-10 and e32 < 1e-4
    A = rng.dirichlet(np.ones(S) * 0.5, size=S)
    pi = rng.dirichlet(np.ones(S))
    rlp, rpath = ho.viterbi_dense(pi, A, ref, lengths)
    off = ctx.tensor(np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64))
    lp, path = sra.viterbi_dense(ctx, ll64, off, pi, A)
    # the float64 log-densities of the GPU differ from the oracle's in the last bits, so its
    # path may part from the oracle's at exact ties of that size only
    assert_ties_only(pi, A, ref, lengths, path.cpu().numpy(), rpath, rlp, eps=1e-12)
    np.testing.assert_allclose(lp.cpu().numpy(), rlp, rtol=1e-11)
    lp_o, path_o = sra.viterbi_dense(ctx, ctx.tensor(ref), off, pi, A)
    np.testing.assert_array_equal(path_o.cpu().numpy(), rpath)          # same inputs: bit-exact
    lp32, path32 = sra.viterbi_dense(ctx, ll32, off, pi, A)
    n_diff, gap, short = assert_ties_only(pi, A, ref, lengths, path32.cpu().numpy(), rpath, rlp)
    print(f"dense Viterbi 1024 states float32: {n_diff} of {U * T} frames differ, largest gap {gap:.2e}")
    assert _rel(lp32.cpu().numpy(), rlp) < 1e-5
