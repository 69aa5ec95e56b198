"""CPU oracle (test infrastructure) for diagonal-Gaussian HMM scoring,
Baum-Welch statistics, re-estimation and Viterbi.

fp64 restatement of ``my_hmmlearn`` with ``implementation="scaling"`` (the
correct one; the reference's ``forward_log`` adds ``transmat`` where it should
add ``log(transmat)``, my_hmmlearn/utils.py:162, and is NOT reproduced).
``*_loops`` functions keep the reference's per-frame / per-state Python loop
structure and are what the CPU baseline times; ``*_vec`` functions are the
same arithmetic with the inner state loops vectorised.  Checked against the
imported reference through ``tests/golden/hmm_*.npz`` (parity pinned).
"""
import numpy as np

LOG_2PI = np.log(2.0 * np.pi)


# ---------------------------------------------------------------- emissions
def log_density_diag(X, means, covars):
    """my_hmmlearn/hmm.py:154-163 -> (T, S)."""
    covars = np.maximum(covars, np.finfo(float).tiny)
    D = means.shape[1]
    d2 = ((X[:, None, :] - means[None]) ** 2 / covars[None]).sum(axis=-1)
    return -0.5 * (D * LOG_2PI + np.log(covars).sum(axis=-1) + d2)


def split_lengths(X, lengths):
    """my_hmmlearn/utils.py:278-287."""
    if lengths is None:
        return [X]
    cs = np.cumsum(lengths)
    if len(cs) == 0 or cs[-1] != len(X):
        raise ValueError(
            f"lengths array {lengths} doesn't sum to {len(X)} samples")
    return np.split(X, cs)[:-1]


# ------------------------------------------------- loop-faithful recursions
def forward_scaling_loops(pi, A, B):
    """my_hmmlearn/utils.py:132-150."""
    T, S = B.shape
    if pi.shape[0] != S or A.shape != (S, S):
        raise ValueError("Shape mismatch")
    alpha = np.zeros((T, S))
    c = np.zeros(T)
    alpha[0] = pi * B[0]
    c[0] = 1.0 / alpha[0].sum()
    alpha[0] *= c[0]
    for t in range(1, T):
        for j in range(S):
            alpha[t, j] = np.sum(alpha[t - 1] * A[:, j]) * B[t, j]
        c[t] = 1.0 / alpha[t].sum()
        alpha[t] *= c[t]
    return -np.sum(np.log(c)), alpha, c


def backward_scaling_loops(pi, A, B, c):
    """my_hmmlearn/utils.py:169-183."""
    T, S = B.shape
    if pi.shape[0] != S or A.shape != (S, S) or c.shape[0] != T:
        raise ValueError("Shape mismatch")
    beta = np.zeros((T, S))
    beta[-1] = c[-1]
    for t in range(T - 2, -1, -1):
        for i in range(S):
            acc = 0.0
            for j in range(S):
                acc += A[i, j] * B[t + 1, j] * beta[t + 1, j]
            beta[t, i] = acc * c[t]
    return beta


def xi_sum_loops(alpha, A, beta, B):
    """my_hmmlearn/utils.py:201-218."""
    T, S = B.shape
    if alpha.shape != (T, S) or A.shape != (S, S) or beta.shape != (T, S):
        raise ValueError("Shape mismatch")
    xi = np.zeros((S, S))
    for t in range(T - 1):
        for i in range(S):
            for j in range(S):
                xi[i, j] += alpha[t, i] * A[i, j] * B[t + 1, j] * beta[t + 1, j]
    return xi


def viterbi_loops(log_pi, log_A, log_B):
    """my_hmmlearn/utils.py:243-266: max-plus lattice, lowest index wins ties,
    backtrace by recomputation."""
    T, S = log_B.shape
    if log_pi.shape[0] != S or log_A.shape != (S, S):
        raise ValueError("Shape mismatch")
    lat = np.full((T, S), -np.inf)
    lat[0] = log_pi + log_B[0]
    for t in range(1, T):
        for j in range(S):
            lat[t, j] = np.max(lat[t - 1] + log_A[:, j]) + log_B[t, j]
    path = np.zeros(T, dtype=np.int64)
    path[-1] = np.argmax(lat[-1])
    for t in range(T - 2, -1, -1):
        path[t] = np.argmax(lat[t] + log_A[:, path[t + 1]])
    return np.max(lat[-1]), path


# ------------------------------------------------------ vectorised variants
def forward_scaling_vec(pi, A, B):
    T, S = B.shape
    alpha = np.empty((T, S))
    c = np.empty(T)
    a = pi * B[0]
    for t in range(T):
        if t:
            a = (a @ A) * B[t]
        c[t] = 1.0 / a.sum()
        a = a * c[t]
        alpha[t] = a
    return -np.sum(np.log(c)), alpha, c


def backward_scaling_vec(pi, A, B, c):
    T, S = B.shape
    beta = np.empty((T, S))
    beta[-1] = c[-1]
    for t in range(T - 2, -1, -1):
        beta[t] = (A @ (B[t + 1] * beta[t + 1])) * c[t]
    return beta


def xi_sum_vec(alpha, A, beta, B):
    return A * (alpha[:-1].T @ (B[1:] * beta[1:]))


def posteriors(alpha, beta):
    """my_hmmlearn/hmm.py:325-328 + utils.py:6-30 (zero rows divide by 1)."""
    g = alpha * beta
    s = g.sum(axis=1, keepdims=True)
    s[s == 0] = 1
    return g / s


# ------------------------------------------------------------ model helpers
class Params:
    """Plain parameter bundle: pi (S,), A (S,S), means (S,D), covars (S,D)."""

    def __init__(self, pi, A, means, covars):
        self.pi = np.array(pi, dtype=np.float64)
        self.A = np.array(A, dtype=np.float64)
        self.means = np.array(means, dtype=np.float64)
        self.covars = np.array(covars, dtype=np.float64)

    @property
    def S(self):
        return self.pi.shape[0]

    @property
    def D(self):
        return self.means.shape[1]

    def copy(self):
        return Params(self.pi, self.A, self.means, self.covars)


def empty_stats(S, D):
    """my_hmmlearn/hmm.py:224-235."""
    return {"nobs": 0, "start": np.zeros(S), "trans": np.zeros((S, S)),
            "post": np.zeros(S), "obs": np.zeros((S, D)),
            "obs**2": np.zeros((S, D))}


def score(p, X, lengths=None, loops=True):
    """score_scaling, my_hmmlearn/hmm.py:308-323: per-utterance forward
    log-probabilities (array) -- their sum is what ``score`` returns."""
    fwd = forward_scaling_loops if loops else forward_scaling_vec
    out = []
    for x in split_lengths(X, lengths):
        with np.errstate(under="ignore"):
            B = np.exp(log_density_diag(x, p.means, p.covars))
        out.append(fwd(p.pi, p.A, B)[0])
    return np.array(out)


def score_posteriors(p, X, lengths=None, loops=True):
    """score_scaling with compute_posteriors=True, my_hmmlearn/hmm.py:308-328:
    (per-utterance log-probabilities, posteriors of all samples (n, S))."""
    fwd = forward_scaling_loops if loops else forward_scaling_vec
    bwd = backward_scaling_loops if loops else backward_scaling_vec
    lps, post = [], [np.empty((0, p.S))]
    for x in split_lengths(X, lengths):
        with np.errstate(under="ignore"):
            B = np.exp(log_density_diag(x, p.means, p.covars))
        lp, alpha, c = fwd(p.pi, p.A, B)
        post.append(posteriors(alpha, bwd(p.pi, p.A, B, c)))
        lps.append(lp)
    return np.array(lps), np.concatenate(post)


def estep(p, X, lengths=None, loops=True):
    """do_estep, my_hmmlearn/hmm.py:419-433 with fit_scaling :330-337 and
    accumulate_sufficient_statistics(+_scaling) :237-253, :339-354.
    Returns (stats, per-utterance log-probs)."""
    fwd = forward_scaling_loops if loops else forward_scaling_vec
    bwd = backward_scaling_loops if loops else backward_scaling_vec
    xis = xi_sum_loops if loops else xi_sum_vec
    st = empty_stats(p.S, p.D)
    lps = []
    for x in split_lengths(X, lengths):
        with np.errstate(under="ignore"):
            B = np.exp(log_density_diag(x, p.means, p.covars))
        lp, alpha, c = fwd(p.pi, p.A, B)
        beta = bwd(p.pi, p.A, B, c)
        g = posteriors(alpha, beta)
        st["post"] += g.sum(axis=0)
        st["obs"] += g.T @ x
        st["obs**2"] += g.T @ x ** 2
        st["nobs"] += 1
        st["start"] += g[0]
        if x.shape[0] > 1:
            st["trans"] += xis(alpha, p.A, beta, B)
        lps.append(lp)
    return st, np.array(lps)


def _normalize_rows(a):
    s = a.sum(axis=-1, keepdims=True)
    if a.ndim > 1:
        s = np.where(s == 0, 1, s)
    return a / s


def mstep_transitions(p, st, startprob_prior=1.0, transmat_prior=1.0):
    """do_em_mstep, my_hmmlearn/hmm.py:435-451 (structural zeros are kept)."""
    q = p.copy()
    sp = np.maximum(startprob_prior - 1 + st["start"], 0)
    q.pi = _normalize_rows(np.where(p.pi == 0, 0, sp))
    tr = np.maximum(transmat_prior - 1 + st["trans"], 0)
    q.A = _normalize_rows(np.where(p.A == 0, 0, tr))
    return q


def mstep_full(p, st, startprob_prior=1.0, transmat_prior=1.0, means_prior=0.0,
               means_weight=0.0, covars_prior=1e-2, covars_weight=1.0):
    """do_mstep, my_hmmlearn/hmm.py:165-184 (the hmmlearn diag M-step)."""
    q = mstep_transitions(p, st, startprob_prior, transmat_prior)
    denom = st["post"][:, None]
    q.means = (means_weight * means_prior + st["obs"]) / (means_weight + denom)
    meandiff = q.means - means_prior
    c_n = (means_weight * meandiff ** 2 + st["obs**2"]
           - 2 * q.means * st["obs"] + q.means ** 2 * denom)
    c_d = max(covars_weight - 1, 0) + denom
    q.covars = (covars_prior + c_n) / np.maximum(c_d, 1e-5)
    return q


def viterbi(p, X, lengths=None):
    """utils.viterbi called as the survey prescribes: log pi, log A, log B."""
    out_lp, out_path = [], []
    with np.errstate(divide="ignore"):
        lpi, lA = np.log(p.pi), np.log(p.A)
    for x in split_lengths(X, lengths):
        lp, path = viterbi_loops(lpi, lA, log_density_diag(x, p.means, p.covars))
        out_lp.append(lp)
        out_path.append(path)
    return np.array(out_lp), np.concatenate(out_path)


def gmm_log_density(X, weights, means, covars):
    """Mixture emission ln sum_m w_sm N(x; mu_sm, var_sm) -> (T, S).  Each component is
    the reference's diagonal density (my_hmmlearn/hmm.py:154-163); the mixture
    itself is hmmlearn's GMMHMM emission, which the reference does not have --
    parity unpinned beyond the n_mix = 1 case, which must equal log_density_diag."""
    X = np.asarray(X, dtype=np.float64)
    S, MX, D = means.shape
    out = np.empty((X.shape[0], S))
    with np.errstate(divide="ignore"):
        lw = np.log(weights)
    for s in range(S):
        comp = log_density_diag(X, means[s], covars[s]) + lw[s][None, :]     # (T, MX)
        mx = comp.max(axis=1, keepdims=True)
        out[:, s] = (mx + np.log(np.exp(comp - mx).sum(axis=1, keepdims=True)))[:, 0]
    return out


def viterbi_dense(pi, A, log_B, lengths=None):
    """utils.viterbi on given log-densities, vectorised over states (same
    arithmetic and tie rule as viterbi_loops)."""
    with np.errstate(divide="ignore"):
        lpi, lA = np.log(pi), np.log(A)
    lps, paths = [], []
    for lb in split_lengths(log_B, lengths):
        T = lb.shape[0]
        lat = np.empty_like(lb)
        lat[0] = lpi + lb[0]
        for t in range(1, T):
            lat[t] = (lat[t - 1][:, None] + lA).max(axis=0) + lb[t]
        path = np.empty(T, dtype=np.int64)
        path[-1] = int(np.argmax(lat[-1]))
        lps.append(lat[-1, path[-1]])
        for t in range(T - 2, -1, -1):
            path[t] = int(np.argmax(lat[t] + lA[:, path[t + 1]]))
        paths.append(path)
    return np.array(lps), np.concatenate(paths)


def random_params(rng, S, D):
    """Synthetic model recipe used by the benchmarks and tests: mu ~ N(0,1),
    var ~ U(0.5,1.5), pi and rows of A ~ Dirichlet(1)."""
    return Params(rng.dirichlet(np.ones(S)), rng.dirichlet(np.ones(S), size=S),
                  rng.standard_normal((S, D)), rng.uniform(0.5, 1.5, (S, D)))
