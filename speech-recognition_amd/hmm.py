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
from .runtime import Context, HMMBank, Ragged, column_moments


# EM iterations a fit runs launch by launch before it captures one as a graph (short fits
# never pay for the capture)
_GRAPH_AFTER = 6


def _is_tensor(X):
    import torch
    return isinstance(X, torch.Tensor)


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
                 device=None, use_graph=False):
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
        self.use_graph = use_graph      # replay long fits' iterations as a HIP graph (measured: no gain)

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
        if isinstance(X, Ragged):
            X = X.X
        self.check_and_set_n_features(X)
        conc = np.full(self.n_components, 1.0 / self.n_components)
        rs = np.random.mtrand._rand
        self.startprob_ = rs.dirichlet(conc)
        self.transmat_ = rs.dirichlet(conc, size=self.n_components)
        np.random.seed(self.random_state)
        rows = np.random.choice(X.shape[0], self.n_components, replace=False)
        if _is_tensor(X):
            # features that are already on the device stay there: the chosen rows and the
            # per-feature variance (the diagonal of np.cov) are all that comes back
            import torch
            self.means_ = X[torch.as_tensor(rows, device=X.device)].double().cpu().numpy()
            self.covars_ = np.tile(column_moments(self._ctx(), X)[1].cpu().numpy() + self.min_covar,
                                   (self.n_components, 1))
            return
        self.means_ = np.array(X[rows], dtype=np.float64)
        cv = np.cov(X.T) + self.min_covar * np.eye(X.shape[1])
        if not cv.shape:
            cv = cv.reshape(1, 1)
        self.covars_ = np.tile(np.diag(cv), (self.n_components, 1))

    def _ctx(self):
        return Context.default(self.device)

    def _param_key(self):
        return hash((self.startprob_.tobytes(), self.transmat_.tobytes(), self.means_.tobytes(),
                     self._covars_.tobytes(), self.means_.shape))

    def _bank(self):
        """The model's parameters on the device.  Kept between calls for as long as the
        host arrays hold the values it was made from (their bytes are compared, so assigning
        or editing an attribute is seen): score / decode on an unchanged model upload nothing."""
        ctx = self._ctx()
        key = (id(ctx), self.dtype, self._param_key())
        cached = self.__dict__.get("_dev_bank")
        if cached is not None and cached[0] == key:
            return cached[1]
        bank = self._make_bank(ctx)
        self._dev_bank = (key, bank)
        return bank

    def _make_bank(self, ctx):
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
        """X: (n_samples, n_features) host array, a device tensor (used where it is), or a
        ``Ragged`` batch that already carries its lengths."""
        if isinstance(X, Ragged):
            self.check_and_set_n_features(X.X)
            if X.code != (_lib.SRX_F64 if self.dtype == "float64" else _lib.SRX_F32):
                return Ragged(X.ctx, X.X, dtype=self.dtype, offsets=X.offsets, offsets_host=X.offsets_host)
            return X
        if not _is_tensor(X):
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
        # (the bank now holds exactly these values: it stays the model's device copy)
        self._dev_bank = ((id(bank.ctx), self.dtype, self._param_key()), bank)

    # -------------------------------------------------------------- inference
    def compute_log_likelihood(self, X):
        """(T, S) emission log-densities, hmm.py:136-163."""
        data = self._data(X, None)
        return self._bank().emission_loglik(data).cpu().numpy().astype(np.float64)

    def compute_likelihood(self, X):
        """(T, S) emission densities, hmm.py:144-152: the exponential of the log-densities
        (the kernels themselves never leave the log domain; far-off frames underflow to 0
        here exactly as they do in the reference)."""
        with np.errstate(under="ignore"):
            return np.exp(self.compute_log_likelihood(X))

    def score_log(self, X, lengths=None, *, compute_posteriors):
        """hmm.py:257-271.  Same recursion as score_scaling: the reference's log-space
        forward pass (utils.py:162) adds transmat where it needs its logarithm and is not
        reproduced."""
        return self.score_scaling(X, lengths, compute_posteriors=compute_posteriors)

    def fit_scaling(self, X):
        """(frameprob, log_prob, posteriors, None, None) of one utterance, hmm.py:330-337:
        the scaled forward / backward lattices are not materialised by the kernels."""
        lp, post = self.score_scaling(X, None, compute_posteriors=True)
        return self.compute_likelihood(X), lp, post, None, None

    fit_log = fit_scaling

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
    def initialize_sufficient_statistics(self):
        """The zeroed statistics dict, hmm.py:224-235."""
        S, D = self.n_components, self.n_features
        return {"nobs": 0, "start": np.zeros(S), "trans": np.zeros((S, S)), "post": np.zeros(S),
                "obs": np.zeros((S, D)), "obs**2": np.zeros((S, D))}

    def accumulate_sufficient_statistics(self, stats, X, lattice=None, posteriors=None,
                                         fwdlattice=None, bwdlattice=None):
        """Add one utterance's statistics to ``stats`` in place, hmm.py:237-253 and
        :339-354.  The lattice arguments of the reference's signature are accepted and not
        used: the E-step kernel computes the utterance's own."""
        X = np.asarray(X)
        one, _ = self.do_estep(X, [X.shape[0]])
        stats["nobs"] += one["nobs"]
        for k in ("start", "trans", "post", "obs", "obs**2"):
            stats[k] += one[k]

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
        self.__dict__.pop("_dev_bank", None)
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
        if not isinstance(X, Ragged):
            if not _is_tensor(X):
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
        # (the iterations rewrite the bank in place: until _pull has brought the host arrays
        # up to date it is not the device copy of what they hold -- a fit that is interrupted
        # must not leave it behind as one)
        self.__dict__.pop("_dev_bank", None)
        pri = self._priors()
        self.monitor_ = ConvergenceMonitor(self.tol, self.n_iter, self.verbose)
        # The host never waits for the iteration it has just queued: the log-probability of
        # iteration k travels to pinned host memory asynchronously and is looked at while
        # iteration k + 1 runs.  If it says "converged", iteration k + 1 is one too many; its
        # M-step is undone from the snapshot taken just before it, so the parameters and the
        # history are the ones of the reference's loop, which checks after every iteration.
        # An iteration is a fixed sequence of launches (E-step kernels, the copy of the
        # log-probability, the snapshot, the M-step kernel); a fit that is still running after
        # _GRAPH_AFTER iterations captures it once per parity (two pinned slots, two snapshots)
        # as a HIP graph and replays those: one host call per iteration instead of a dozen.
        # (use_graph=True.  Off by default: at the reference's own job size the iteration is
        # bound by the GPU's chain of dependent kernels -- 0.059 ms launch by launch and 0.067 ms
        # replayed on 150 utterances -- so the host calls were never what it waited for.)
        n_iter = int(self.n_iter)
        ctx = bank.ctx
        slots = torch.full((2,), float("nan"), dtype=torch.float64).pin_memory()
        events = [torch.cuda.Event() for _ in range(2)]
        # (the events go on the stream that carries the kernels and the copy: the context's
        # device may not be the process's current one)
        stream = torch.cuda.current_stream(ctx.device)
        stats = torch.empty((bank.M, bank.stats_size), dtype=torch.float64, device=ctx.device)
        snaps = [torch.empty_like(bank._flat) for _ in range(2)]
        # (the reference's own M-step uses the start and transition sums only: the E-step of a
        # small job then skips its pass over the features)
        which = "transitions" if self.mstep == "transitions_only" else "all"

        def iteration(p):
            bank.estep(data, out=stats, stats=which)
            slots[p:p + 1].copy_(stats[0, 1:2], non_blocking=True)
            snaps[p].copy_(bank._flat)                      # parameters before this M-step
            bank.mstep(stats, pri, self.mstep)

        def capture(p):
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(ctx.device)
            side.wait_stream(stream)
            with torch.cuda.stream(side):
                g.capture_begin(capture_error_mode="relaxed")
                try:
                    iteration(p)
                finally:
                    g.capture_end()
            return g

        graphs = None
        use_graphs = getattr(self, "use_graph", False) and n_iter > _GRAPH_AFTER + 2
        stopped = False
        with torch.cuda.device(ctx.device):
            for k in range(n_iter):
                if use_graphs and graphs is None and k == _GRAPH_AFTER:
                    try:
                        graphs = [capture(0), capture(1)]
                    except Exception as e:      # noqa: BLE001  (no graph: the launches stay eager)
                        warnings.warn(f"EM iteration not captured as a graph ({e}); running eagerly",
                                      RuntimeWarning, stacklevel=2)
                        use_graphs = False
                if graphs is not None:
                    graphs[k % 2].replay()
                else:
                    iteration(k % 2)
                events[k % 2].record(stream)
                if k >= 1:
                    events[(k - 1) % 2].synchronize()
                    self.monitor_.report(float(slots[(k - 1) % 2]))
                    if self.monitor_.converged:
                        bank.restore(snaps[k % 2])              # drop iteration k
                        stopped = True
                        break
            if not stopped and n_iter >= 1:
                events[(n_iter - 1) % 2].synchronize()
                self.monitor_.report(float(slots[(n_iter - 1) % 2]))
            torch.cuda.current_stream(ctx.device).synchronize()
        del graphs
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
        if isinstance(X, Ragged):
            X = X.X
        if _is_tensor(X):          # (the clustering runs on the host, once per fit)
            X = X.double().cpu().numpy()
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
