"""Diagonal-Gaussian HMM with the reference's class surface, computed on the GPU.

``MyGaussianHMM`` keeps the constructor, attributes and methods of
``my_hmmlearn.hmm.MyGaussianHMM`` (my_hmmlearn/hmm.py:73-451), which is
itself duck-typed to ``hmmlearn.hmm.GaussianHMM`` as ``HMM_lib.py:33`` uses it
(``GaussianHMM`` below adds hmmlearn's own initialisation and switches on top of
it).  Parameters live on the host as
numpy arrays, like the reference's; scoring, the E-step, the M-step and
Viterbi run in libsrx kernels.

Differences from the reference, on purpose:
  * ``implementation`` is accepted for compatibility; both values run the same
    scaled recursion (the reference's "log" forward adds ``transmat`` where it
    needs ``log(transmat)``, my_hmmlearn/utils.py:162 -- that result is wrong
    and is not reproduced).
  * ``mstep`` chooses what ``fit`` re-estimates: "transitions_only" (default) is the
    reference's literal behaviour -- its ``fit`` calls only ``do_em_mstep``
    (hmm.py:213), so means and covariances stay as ``init`` drew them; "full" also
    runs ``do_mstep`` (hmm.py:165-184, hmmlearn's diagonal M-step), which the
    reference defines and never calls.  ``GaussianHMM`` (hmmlearn's class) is "full".
  * ``decode`` works (hmm.py:357-373 raises TypeError).
  * ``dtype="float64"`` (default) is the validation arithmetic; "float32" is
    the fast path.
"""
import sys
from collections import deque

import warnings

import numpy as np

from . import _lib
from .runtime import Context, HMMBank, Ragged


class ConvergenceMonitor:
    """my_hmmlearn/hmm.py:12-69: history of log-probabilities, one report line
    per iteration on stderr, converged when the iteration budget is used up or
    the last gain is below ``tol``."""

    _template = "{iter:>10d} {log_prob:>16.8f} {delta:>+16.8f}"

    def __init__(self, tol, n_iter, verbose):
        self.tol = tol
        self.n_iter = n_iter
        self.verbose = verbose
        self.history = deque()
        self.iter = 0

    def __repr__(self):
        fields = sorted(dict(vars(self), history=list(self.history)).items())
        body = "".join(f"    {k}={v},\n" for k, v in fields)
        return f"{self.__class__.__name__}(\n{body})"

    def _reset(self):
        self.iter = 0
        self.history.clear()

    def report(self, log_prob):
        if self.verbose:
            delta = log_prob - self.history[-1] if self.history else np.nan
            print(self._template.format(iter=self.iter + 1, log_prob=log_prob, delta=delta),
                  file=sys.stderr)
        slack = np.finfo(float).eps ** 0.5
        if self.history and (log_prob - self.history[-1]) < -slack:
            delta = log_prob - self.history[-1]
            print(f"Model is not converging.  Current: {log_prob} is not greater than "
                  f"{self.history[-1]}. Delta is {delta}")
        self.history.append(log_prob)
        self.iter += 1

    @property
    def converged(self):
        return (self.iter == self.n_iter or
                (len(self.history) >= 2 and self.history[-1] - self.history[-2] < self.tol))


class MyGaussianHMM:
    def __init__(self, n_components=1, covariance_type="diag", min_covar=1e-3,
                 startprob_prior=1.0, transmat_prior=1.0, means_prior=0, means_weight=0,
                 covars_prior=1e-2, covars_weight=1, random_state=None, n_iter=10, tol=1e-2,
                 verbose=True, implementation="log", *, mstep="transitions_only", dtype="float64",
                 device=None):
        if covariance_type != "diag":
            raise ValueError("only covariance_type='diag' is implemented")
        if implementation not in ("log", "scaling"):
            raise KeyError(implementation)
        if mstep not in ("full", "transitions_only"):
            raise ValueError("mstep must be 'full' or 'transitions_only'")
        self.n_components = n_components
        self.n_iter = n_iter
        self.tol = tol
        self.verbose = verbose
        self.implementation = implementation
        self.random_state = random_state
        self.startprob_prior = startprob_prior
        self.transmat_prior = transmat_prior
        self.monitor_ = ConvergenceMonitor(self.tol, self.n_iter, self.verbose)
        self.covariance_type = covariance_type
        self.min_covar = min_covar
        self.means_prior = means_prior
        self.means_weight = means_weight
        self.covars_prior = covars_prior
        self.covars_weight = covars_weight
        self.mstep = mstep
        self.dtype = dtype
        self.device = device

    # pickled models carry parameters only, never device state
    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_dev")}

    # ------------------------------------------------------------ attributes
    @property
    def covars_(self):
        """Full (S, D, D) covariance matrices built from the diagonals."""
        return np.array([np.diag(c) for c in self._covars_])

    @covars_.setter
    def covars_(self, covars):
        covars = np.array(covars, copy=True)
        if covars.ndim != 2:
            raise ValueError("'diag' covars must have shape (n_components, n_dim)")
        if np.any(covars <= 0):
            raise ValueError("'diag' covars must be positive")
        self._covars_ = covars

    def check_and_set_n_features(self, X):
        _, n_features = X.shape
        if hasattr(self, "n_features"):
            if self.n_features != n_features:
                raise ValueError(f"Unexpected number of dimensions, got {n_features} but "
                                 f"expected {self.n_features}")
        else:
            self.n_features = n_features

    # ---------------------------------------------------------------- set-up
    def init(self, X, lengths=None):
        """Initial parameters drawn the way the reference draws them
        (my_hmmlearn/hmm.py:116-134): Dirichlet(1/S) start/transition rows from
        numpy's global generator, then ``np.random.seed(random_state)`` and S
        distinct data rows as means, data variance + min_covar as covariances."""
        self.check_and_set_n_features(X)
        conc = np.full(self.n_components, 1.0 / self.n_components)
        rs = np.random.mtrand._rand
        self.startprob_ = rs.dirichlet(conc)
        self.transmat_ = rs.dirichlet(conc, size=self.n_components)
        np.random.seed(self.random_state)
        rows = np.random.choice(X.shape[0], self.n_components, replace=False)
        self.means_ = np.array(X[rows], dtype=np.float64)
        cv = np.cov(X.T) + self.min_covar * np.eye(X.shape[1])
        if not cv.shape:
            cv = cv.reshape(1, 1)
        self.covars_ = np.tile(np.diag(cv), (self.n_components, 1))

    def _ctx(self):
        return Context.default(self.device)

    def _bank(self):
        ctx = self._ctx()
        if self.dtype == "float32":
            # the float32 kernels evaluate the Gaussian exponent in expanded form (see
            # HMMBank.conditioning): fine for normalised features, not for raw ones
            cond = float(np.max(np.square(self.means_) / self._covars_))
            if cond > 1e4:
                warnings.warn(f"means are up to {cond ** 0.5:.0f} standard deviations from zero: float32 scoring "
                              "loses accuracy on such features, use dtype='float64' or normalise them",
                              RuntimeWarning, stacklevel=3)
        return HMMBank(ctx, self.startprob_, self.transmat_, self.means_, self._covars_)

    def _data(self, X, lengths):
        X = np.asarray(X)
        if X.ndim != 2:
            raise ValueError("X must be 2-D (n_samples, n_features)")
        self.check_and_set_n_features(X)
        return Ragged(self._ctx(), X, lengths, dtype=self.dtype)

    def _priors(self):
        return _lib.Priors(float(self.startprob_prior), float(self.transmat_prior),
                           float(self.means_prior), float(self.means_weight),
                           float(self.covars_prior), float(self.covars_weight))

    def _pull(self, bank):
        sp, A, mu, cv = bank.params_host()
        self.startprob_, self.transmat_, self.means_, self._covars_ = sp[0], A[0], mu[0], cv[0]

    # -------------------------------------------------------------- inference
    def compute_log_likelihood(self, X):
        """(T, S) emission log-densities, hmm.py:136-163."""
        data = self._data(X, None)
        return self._bank().emission_loglik(data).cpu().numpy().astype(np.float64)

    def score(self, X, lengths=None, *, compute_posteriors=False):
        """Sum over utterances of the forward log-probability, hmm.py:186-196
        (like the reference, only the log-probability is returned whatever
        compute_posteriors says)."""
        data = self._data(X, lengths)
        return float(self._bank().score(data).sum().item())

    def score_scaling(self, X, lengths=None, *, compute_posteriors):
        """(summed log-probability, posteriors (n_samples, S)), hmm.py:308-323;
        without compute_posteriors the second item is the reference's empty
        (0, S) array."""
        data = self._data(X, lengths)
        if not compute_posteriors:
            return (float(self._bank().score(data).sum().item()),
                    np.empty((0, self.n_components)))
        lp, gamma = self._bank().posteriors(data)
        return float(lp.sum().item()), gamma.cpu().numpy().astype(np.float64)

    def score_samples(self, X, lengths=None):
        """hmmlearn's name for (log-probability, posteriors)."""
        return self.score_scaling(X, lengths, compute_posteriors=True)

    def predict_proba(self, X, lengths=None):
        """State posteriors of every sample, (n_samples, S)."""
        return self.score_scaling(X, lengths, compute_posteriors=True)[1]

    def predict(self, X, lengths=None):
        """Most likely state sequence (the Viterbi path, hmmlearn's default)."""
        return self.decode(X, lengths)[1]

    def score_utterances(self, X, lengths=None):
        """Per-utterance forward log-probabilities (U,)."""
        data = self._data(X, lengths)
        return self._bank().score(data)[:, 0].cpu().numpy()

    def decode(self, X, lengths=None):
        """(summed Viterbi log-probability, state path), the contract of
        hmmlearn's decode that hmm.py:357-373 aims at; recursion of
        my_hmmlearn/utils.py:243-266 per utterance."""
        data = self._data(X, lengths)
        lp, path = self._bank().viterbi(data)
        return float(lp.sum().item()), path.cpu().numpy().astype(int)

    # --------------------------------------------------------------- sampling
    def generate_sample_from_state(self, state, random_state):
        """One draw from a state's Gaussian, hmm.py:411-417."""
        return random_state.multivariate_normal(self.means_[state], self.covars_[state])

    def sample(self, n_samples=1, random_state=None, currstate=None):
        """(X (n_samples, D), states) drawn from the model, hmm.py:375-398;
        random_state is a numpy RandomState (or a seed, or None for the model's).
        The reference's own sample() stops in its covars_ getter (hmm.py:104-108
        passes one argument too many), so this follows the code as written with
        the diagonal covariances expanded, and no reference output pins it."""
        if random_state is None:
            random_state = self.random_state
        if not hasattr(random_state, "rand"):
            random_state = np.random.RandomState(random_state)
        transmat_cdf = np.cumsum(self.transmat_, axis=1)
        if currstate is None:
            currstate = (np.cumsum(self.startprob_) > random_state.rand()).argmax()
        states = [currstate]
        X = [self.generate_sample_from_state(currstate, random_state)]
        for _ in range(n_samples - 1):
            currstate = (transmat_cdf[currstate] > random_state.rand()).argmax()
            states.append(currstate)
            X.append(self.generate_sample_from_state(currstate, random_state))
        return np.atleast_2d(X), np.array(states, dtype=int)

    # --------------------------------------------------------------- training
    def do_estep(self, X, lengths):
        """(stats dict, summed log-probability), hmm.py:419-433."""
        data = self._data(X, lengths)
        bank = self._bank()
        st = bank.unpack_stats(bank.estep(data))[0]
        lp = st.pop("logprob")
        return st, lp

    def _pack_stats(self, stats, logprob=0.0):
        S, D = self.n_components, self.n_features
        rec = np.concatenate([[float(stats["nobs"]), float(logprob)], np.ravel(stats["start"]),
                              np.ravel(stats["trans"]), np.ravel(stats["post"]),
                              np.ravel(stats["obs"]), np.ravel(stats["obs**2"])])
        assert rec.size == 2 + S + S * S + S + 2 * S * D
        return rec.astype(np.float64)

    def _mstep_from_dict(self, stats, mode):
        bank = self._bank()
        rec = bank.ctx.tensor(self._pack_stats(stats)[None])
        bank.mstep(rec, self._priors(), mode)
        self._pull(bank)

    def do_em_mstep(self, stats):
        """Start and transition probabilities only, hmm.py:435-451."""
        self._mstep_from_dict(stats, "transitions_only")

    def do_mstep(self, stats):
        """Full re-estimation, hmm.py:165-184."""
        self._mstep_from_dict(stats, "full")

    def fit(self, X, lengths=None):
        """EM driver, hmm.py:198-222: the log-probability reported for an
        iteration is the one computed before that iteration's M-step."""
        X = np.asarray(X)
        if lengths is None:
            lengths = np.asarray([X.shape[0]])
        self.init(X, lengths)
        return self.fit_from_current(X, lengths)

    def fit_from_current(self, X, lengths=None):
        """EM iterations from the parameters the object already holds (the way
        to get runs that do not depend on the global random state)."""
        import torch
        data = self._data(X, lengths)
        bank = self._bank()
        pri = self._priors()
        self.monitor_ = ConvergenceMonitor(self.tol, self.n_iter, self.verbose)
        # The host never waits for the iteration it has just queued: the log-probability of
        # iteration k travels to pinned host memory asynchronously and is looked at while
        # iteration k + 1 runs.  If it says "converged", iteration k + 1 is one too many; its
        # M-step is undone from the snapshot taken just before it, so the parameters and the
        # history are the ones of the reference's loop, which checks after every iteration.
        n_iter = int(self.n_iter)
        logp = torch.empty((max(n_iter, 1),), dtype=torch.float64).pin_memory()
        events = [torch.cuda.Event() for _ in range(2)]
        snap = None
        stats = None
        stopped = False
        # (the reference's own M-step uses the start and transition sums only: the E-step of a
        # small job then skips its pass over the features)
        bank.ctx.set_estep_stats("transitions" if self.mstep == "transitions_only" else "all")
        try:
            for k in range(n_iter):
                stats = bank.estep(data, out=stats)
                logp[k:k + 1].copy_(stats[0, 1:2], non_blocking=True)
                snap = bank.snapshot(snap)                      # parameters before M-step k
                bank.mstep(stats, pri, self.mstep)
                events[k % 2].record()
                if k >= 1:
                    events[(k - 1) % 2].synchronize()
                    self.monitor_.report(float(logp[k - 1]))
                    if self.monitor_.converged:
                        bank.restore(snap)                      # drop iteration k
                        stopped = True
                        break
        finally:
            bank.ctx.set_estep_stats("all")
        if not stopped and n_iter >= 1:
            events[(n_iter - 1) % 2].synchronize()
            self.monitor_.report(float(logp[n_iter - 1]))
        self._pull(bank)
        if (self.transmat_.sum(axis=1) == 0).any():
            print("Some rows of transmat_ have zero sum because no transition from the state "
                  "was ever observed.", file=sys.stderr)
        return self


class GaussianHMM(MyGaussianHMM):
    """``hmmlearn.hmm.GaussianHMM`` as ``HMM_lib.py:33-34`` constructs and uses it
    (``GaussianHMM(n_components=3, tol=1e-4, n_iter=2000, implementation="log",
    verbose=True).fit(X, lengths)``, then ``score``).  hmmlearn is not in the image
    and the reference pins no output at this boundary, so everything specific to
    this class -- the initialisation below, ``params`` / ``init_params``, the "map"
    decoder -- follows hmmlearn 0.3's published behaviour and is parity unpinned;
    the EM arithmetic is the pinned one of the base class.

    Initialisation (letters of ``init_params``): "s" / "t" uniform start and
    transition probabilities, "m" the centres of a k-means clustering of the data
    (scikit-learn, ``random_state``), "c" the data's per-feature variance plus
    ``min_covar`` for every state.  A letter left out keeps the attribute the caller
    has set.  ``params`` lists what ``fit`` re-estimates.  The clustering runs on the
    host: it happens once per fit and is not part of the per-iteration path."""

    def __init__(self, n_components=1, covariance_type="diag", min_covar=1e-3,
                 startprob_prior=1.0, transmat_prior=1.0, means_prior=0, means_weight=0,
                 covars_prior=1e-2, covars_weight=1, algorithm="viterbi", random_state=None,
                 n_iter=10, tol=1e-2, verbose=False, params="stmc", init_params="stmc",
                 implementation="log", *, dtype="float64", device=None):
        super().__init__(n_components, covariance_type, min_covar, startprob_prior,
                         transmat_prior, means_prior, means_weight, covars_prior, covars_weight,
                         random_state, n_iter, tol, verbose, implementation, mstep="full",
                         dtype=dtype, device=device)
        if algorithm not in ("viterbi", "map"):
            raise ValueError(f"Unknown decoder {algorithm!r}")
        self.algorithm = algorithm
        self.params = params
        self.init_params = init_params

    def init(self, X, lengths=None):
        from sklearn.cluster import KMeans
        X = np.asarray(X, dtype=np.float64)
        self.check_and_set_n_features(X)
        S = self.n_components
        if "s" in self.init_params or not hasattr(self, "startprob_"):
            self.startprob_ = np.full(S, 1.0 / S)
        if "t" in self.init_params or not hasattr(self, "transmat_"):
            self.transmat_ = np.full((S, S), 1.0 / S)
        if "m" in self.init_params or not hasattr(self, "means_"):
            km = KMeans(n_clusters=S, random_state=self.random_state, n_init=10)
            self.means_ = np.array(km.fit(X).cluster_centers_, dtype=np.float64)
        if "c" in self.init_params or not hasattr(self, "_covars_"):
            cv = np.cov(X.T) + self.min_covar * np.eye(X.shape[1])
            if not cv.shape:
                cv = cv.reshape(1, 1)
            self.covars_ = np.tile(np.diag(cv), (S, 1))

    def fit_from_current(self, X, lengths=None):
        wanted = set(self.params) & set("stmc")
        if wanted == set("stmc"):
            self.mstep = "full"
            return super().fit_from_current(X, lengths)
        if wanted == set("st"):
            self.mstep = "transitions_only"
            return super().fit_from_current(X, lengths)
        # any other subset: full re-estimation on the device, the frozen parameters put
        # back after every iteration
        data = self._data(X, lengths)
        pri = self._priors()
        self.monitor_ = ConvergenceMonitor(self.tol, self.n_iter, self.verbose)
        for _ in range(self.n_iter):
            keep = (self.startprob_.copy(), self.transmat_.copy(), self.means_.copy(),
                    self._covars_.copy())
            bank = self._bank()
            stats = bank.estep(data)
            logprob = float(stats[0, 1].item())
            bank.mstep(stats, pri, "full")
            self._pull(bank)
            if "s" not in wanted:
                self.startprob_ = keep[0]
            if "t" not in wanted:
                self.transmat_ = keep[1]
            if "m" not in wanted:
                self.means_ = keep[2]
            if "c" not in wanted:
                self._covars_ = keep[3]
            self.monitor_.report(logprob)
            if self.monitor_.converged:
                break
        return self

    def decode(self, X, lengths=None, algorithm=None):
        """(log-probability, states): "viterbi" is the base class's decoder; "map"
        takes the most probable state of every sample from the posteriors."""
        algorithm = algorithm or self.algorithm
        if algorithm == "viterbi":
            return super().decode(X, lengths)
        if algorithm != "map":
            raise ValueError(f"Unknown decoder {algorithm!r}")
        _, post = self.score_samples(X, lengths)
        return float(np.max(post, axis=1).sum()), np.argmax(post, axis=1)
