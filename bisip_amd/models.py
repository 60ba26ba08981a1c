"""Model classes: the reference's ``bisip.models`` surface over the HIP kernels.

Same class names, constructor arguments, parameter dictionaries, ``forward`` /
``_log_prior`` / ``_log_likelihood`` / ``_log_probability`` signatures and ``fit()``
workflow as the reference (src/bisip/models.py), so user code only changes its
import.  What differs is where the arithmetic runs: every log-probability and
forward evaluation is a launch of a gfx950 kernel through ``libbisip_hip.so``, and
all of them accept a whole ensemble ``theta (n, ndim)`` at once (the emcee
``vectorize=True`` contract) besides the reference's single ``theta (ndim,)``.

There is no CPU fallback: without the HIP library or without a GPU these methods
raise ``RuntimeError``.
"""

import warnings

import numpy as np

from . import _hip
from . import utils as _utils
from .sampler import DeviceEnsembleSampler, EnsembleSampler


def moves_sampler(sampler):
    """True for a real emcee sampler (fit(moves=...)): it has no device state to check."""
    return not isinstance(sampler, (DeviceEnsembleSampler, EnsembleSampler))


class Inversion(_utils.utils):
    """Base class for the SIP inversion models (reference: src/bisip/models.py:21-179).

    Args:
        filepath (str): path of the 5-column data file.
        nwalkers (int): number of walkers. Defaults to 32.
        nsteps (int): number of MCMC steps. Defaults to 5000.
        headers (int): number of header lines to skip. Defaults to 1.
        ph_units (str): 'mrad', 'rad' or 'deg'. Defaults to 'mrad'.
        device (int): GPU ordinal the model's operands live on (new; default 0).
    """

    _model_id = None

    def __init__(self, filepath, nwalkers=32, nsteps=5000, headers=1, ph_units='mrad',
                 device=0):
        self.filepath = filepath
        self.nwalkers = nwalkers
        self.nsteps = nsteps
        self.headers = headers
        self.ph_units = ph_units
        self.device = device

        self._p0 = None
        self._params = {}
        self.__fitted = False
        self._ctx_cache = {}
        self._sampler = None

        self._data = self.load_data(self.filepath, self.headers, self.ph_units)

    def __getattr__(self, name):
        # the reference mixes matplotlib figures in (src/bisip/plotlib.py); they are outside
        # the hot path this package replaces -- say so instead of a bare AttributeError
        if name.startswith('plot_') or name == 'print_latex_parameters':
            raise NotImplementedError(
                f'{name} is part of bisip\'s plotting/notebook layer, which bisip_amd does not '
                'reimplement; pass model.get_chain(...) / get_model_percentile(...) to bisip.plotlib')
        raise AttributeError(f'{type(self).__name__!s} object has no attribute {name!r}')

    # -- device contexts ----------------------------------------------------------------
    def _desc(self):
        """Model-specific kwargs of HipContext; overridden by subclasses."""
        return {}

    def _context(self, x=None, y=None, yerr=None, prior=True):
        """The HIP context holding (x, y, yerr) = (w, zn, zn_err).  Contexts are cached
        by operand bytes, so the usual call pattern (always the loaded data) builds
        exactly one.  ``prior=False``: a second context of the same operands whose box is
        the whole space -- the likelihood alone -- so that ``_log_likelihood`` never touches
        the bounds of the context ``fit()`` and ``_log_probability`` use."""
        d = self._data
        x = d['w'] if x is None else np.ascontiguousarray(x, dtype=np.float64)
        y = d['zn'] if y is None else np.ascontiguousarray(y, dtype=np.float64)
        yerr = d['zn_err'] if yerr is None else np.ascontiguousarray(yerr, dtype=np.float64)
        desc = self._desc()
        key = (x.tobytes(), y.tobytes(), yerr.tobytes(), bool(prior),
               tuple(sorted((k, np.asarray(v).tobytes()) for k, v in desc.items())))
        ctx = self._ctx_cache.get(key)
        if ctx is None:
            bounds = self.param_bounds
            if not prior:
                bounds = np.array([np.full(bounds.shape[1], -np.inf), np.full(bounds.shape[1], np.inf)])
                # no box to centre the QR-reduced form in: the per-frequency kernel is accurate
                # for any theta
                if desc.get('variant') in ('auto', 'reduced', 'reduced_comp'):
                    desc = dict(desc, variant='collapsed')
            ctx = _hip.HipContext(self._model_id, x, y, yerr, bounds, device=self.device, **desc)
            if len(self._ctx_cache) >= 8:
                self._ctx_cache.pop(next(iter(self._ctx_cache))).close()
            self._ctx_cache[key] = ctx
        return ctx

    def _is_own_forward(self, f):
        return f is None or f == self.forward

    @staticmethod
    def _as_rows(theta):
        theta = np.asarray(theta, dtype=np.float64)
        if theta.ndim == 1:
            return theta[None, :], True
        if theta.ndim != 2:
            raise ValueError(f'theta must be (ndim,) or (n, ndim), got shape {theta.shape}')
        return theta, False

    # -- the log-probability trio (reference: src/bisip/models.py:59-76) ------------------
    def _log_likelihood(self, theta, f, x, y, yerr):
        """Gaussian log-likelihood  -0.5*sum((y - f(theta,x))**2/yerr**2 + 2*log(yerr**2)).

        ``theta`` may be one vector or an (n, ndim) ensemble.  ``f`` is the model callable
        ``f(theta, x) -> (2, N)`` as in the reference (src/bisip/models.py:59-62): this
        model's own ``forward`` (or None) runs the fused forward + reduction kernel; any
        other callable is evaluated on the host, row by row as the reference does, and only
        the residual reduction runs on the device (``bisip_loglike_z``)."""
        rows, single = self._as_rows(theta)
        ctx = self._context(x, y, yerr, prior=False)
        if self._is_own_forward(f):
            out = ctx.logprob(rows)
        else:
            xs = self._data['w'] if x is None else x
            Z = np.stack([np.asarray(f(t, xs), dtype=np.float64) for t in rows])
            out = ctx.loglike_z(Z)
        return float(out[0]) if single else out

    def _log_prior(self, theta, bounds):
        """Uniform prior on the OPEN box: 0 inside, -inf on or outside a bound."""
        rows, single = self._as_rows(theta)
        bounds = np.asarray(bounds, dtype=np.float64)
        inside = np.logical_and(bounds[0] < rows, rows < bounds[1]).all(axis=1)
        out = np.where(inside, 0.0, -np.inf)
        return float(out[0]) if single else out

    def _log_probability(self, theta, model=None, bounds=None, x=None, y=None, yerr=None):
        """Bayes numerator: prior + likelihood, fused in one kernel launch.  Rows outside
        the prior return -inf without evaluating the forward model."""
        rows, single = self._as_rows(theta)
        bounds = self.param_bounds if bounds is None else np.asarray(bounds, dtype=np.float64)
        if self._is_own_forward(model):
            ctx = self._context(x, y, yerr)
            ctx.set_bounds(bounds)
            out = ctx.logprob(rows)
        else:   # a foreign model callable: prior on the host, f only for the rows inside it
            out = np.array(self._log_prior(rows, bounds), dtype=np.float64)
            inside = np.isfinite(out)
            if inside.any():
                out[inside] += self._log_likelihood(rows[inside], model, x, y, yerr)
        return float(out[0]) if single else out

    # public aliases named by the north star
    log_likelihood = _log_likelihood
    log_prior = _log_prior
    log_probability = _log_probability
    log_prob = _log_probability

    def _forward_rows(self, theta, w):
        rows, single = self._as_rows(theta)
        d = self._data
        w = d['w'] if w is None else np.ascontiguousarray(w, dtype=np.float64)
        if w.shape == d['w'].shape and np.array_equal(w, d['w']):
            ctx = self._context()
        else:  # forward on another frequency grid: operands y/yerr are irrelevant
            ones = np.ones((2, w.size))
            ctx = self._context(w, ones, ones)
        Z = ctx.forward(rows)
        return Z[0] if single else Z

    # -- sampling (reference: src/bisip/models.py:78-137) ----------------------------------
    def _check_if_fitted(self):
        if not self.fitted:
            raise AssertionError('Model is not fitted! Fit the model to a '
                                 'dataset before attempting to plot results.')

    # from this size on the host's sequential MT19937 stream (~30 ns per walker-step) is what a run waits for
    # (4096 walkers 16 k vs 44 k iterations/s, 32768 walkers 1.9 k vs 7.7 k; below ~1000 walkers the two
    # streams are within 10 %): fit() then RECOMMENDS rng='philox' and rng='auto' picks it
    _PHILOX_FROM_WALKERS = 2048

    def fit(self, p0=None, pool=None, moves=None, sampler='device', rng=None, thin_by=1,
            persistent=None, chain='host'):
        """Sample the posterior with the stretch-move ensemble sampler.

        Args:
            p0 (ndarray): starting positions (nwalkers, ndim); drawn uniformly from the
                prior box with the global NumPy RNG when None.
            pool: accepted for signature compatibility and ignored: the reference hands it to
                emcee to spread the per-walker Python calls over processes
                (src/bisip/models.py:91-94,115); here a whole half-ensemble is one kernel launch.
            moves: an emcee ``moves`` object; requires emcee (the native samplers
                implement the default StretchMove only) -- emcee then drives the vectorised
                GPU log-probability (``vectorize=True``).
            sampler (str): 'device' (default) keeps the ensemble and the chain on the GPU
                and runs one fused kernel per half-step; 'host' runs the stretch move in
                NumPy around the vectorised GPU log-probability.  Same chain either way.
            rng (str): 'numpy' (default) draws the stretch-move random stream on the host in emcee's
                consumption order from NumPy's global state: ``np.random.seed`` pins the run as it does in
                the reference's notebooks, and the chain is the host sampler's bit
                for bit.  'philox' generates the stream on the device: its own reproducible stream (keyed
                from the seeded global state), and the faster one for ensembles of thousands of walkers,
                where drawing MT19937 on the host (~30 ns per walker-step) is what a run waits for (32768
                walkers: 1.9 k vs 7.7 k iterations/s).  'auto': 'numpy' below 2048 walkers, 'philox' from
                there on.  Left at its default with 2048 walkers or more, fit() keeps NumPy's stream and
                says in a UserWarning that rng='philox' is 3-4x faster at that size.  Device sampler only.
            thin_by (int): store one sample every ``thin_by`` iterations.
            chain (str): device sampler: 'host' (default) copies the stored samples to host memory
                as the run proceeds; 'device' keeps them in HBM -- get_param_mean / get_param_std /
                get_param_percentile (called without a chain) then summarise them on the device
                and get_chain() copies on demand.  For big ensembles the copy costs more than the run.
            persistent (bool or None): device sampler: run all iterations of a chunk inside
                one kernel launch (one workgroup holds the ensemble) instead of one launch per
                half-step.  Same chain, bit for bit.  None (default): up to the ensemble size where
                one compute unit still beats launches spread over the chip (256-1024 walkers
                depending on the model, HipContext.persistent_walkers).
        """
        self._p0 = p0
        self.ndim = self.param_bounds.shape[1]
        if self._p0 is None:
            self._p0 = np.random.uniform(*self.param_bounds, (self.nwalkers, self.ndim))

        ctx = self._context()
        ctx.set_bounds(self.param_bounds)  # bounds are read at fit() time, not at construction
        if moves is not None:
            import emcee  # optional: non-default moves
            self._sampler = emcee.EnsembleSampler(self.nwalkers, self.ndim, ctx.logprob,
                                                  moves=moves, vectorize=True)
        elif sampler == 'device' and ctx.variant in ('faithful', 'wave'):
            # these formulations have no stretch-move kernel: the move runs on the host around
            # the vectorised log-probability of exactly that formulation (same chain contract)
            self._sampler = EnsembleSampler(self.nwalkers, self.ndim, ctx.logprob)
        elif sampler == 'device':
            if chain not in ('host', 'device'):
                raise ValueError("chain must be 'host' or 'device'")
            if rng is None:
                rng = 'numpy'
                if self.nwalkers >= self._PHILOX_FROM_WALKERS:
                    warnings.warn(f"{self.nwalkers} walkers: the stretch move's random numbers are drawn on the "
                                  "host in emcee's order (rng='numpy', the reference's seeding behaviour); "
                                  "rng='philox' draws them on the device, 3-4x faster at this size",
                                  UserWarning, stacklevel=2)
            elif rng == 'auto':
                rng = 'philox' if self.nwalkers >= self._PHILOX_FROM_WALKERS else 'numpy'
            self._sampler = DeviceEnsembleSampler(self.nwalkers, self.ndim, ctx, rng=rng,
                                                  persistent=persistent, chain_on_device=(chain == 'device'))
        elif sampler == 'host':
            self._sampler = EnsembleSampler(self.nwalkers, self.ndim, ctx.logprob)
        else:
            raise ValueError("sampler must be 'device' or 'host'")
        if moves is not None:
            self._sampler.run_mcmc(self._p0, self.nsteps, progress=True)
        else:
            self._sampler.run_mcmc(self._p0, self.nsteps, progress=True, thin_by=thin_by)
        self.__fitted = True
        self._check_reduced_kernel(ctx)

    # the tolerance the kernels are held to against the reference
    _LOGP_TOL = 1e-10

    def _check_reduced_kernel(self, ctx):
        """PolynomialDecomposition runs on a QR-reduced kernel chosen from an error ESTIMATE on probe
        rows (HipContext.reduced_error).  After a fit, measure -- against the library's long-double
        yardstick, on the host, microseconds for a few thousand rows -- (1) the final ensemble's
        log-probabilities and (2) the 256 stored samples whose log-probability is nearest to ZERO: the
        tolerance is relative to max(1, |logp|), so it is where a walker crossed logp = 0 on its way in
        that a kernel's absolute error counts, not around the mode.  The worse of the two stays in
        ``reduced_check_`` (the second alone in ``reduced_check_shell_``); beyond the parity tolerance it
        warns and names the way out."""
        self.reduced_check_ = self.reduced_check_shell_ = None
        if ctx.variant not in ('reduced', 'reduced_comp') or moves_sampler(self._sampler):
            return
        coords, lp = self._sampler._coords, self._sampler._lp
        if coords is None or lp is None:
            return
        step = max(1, len(coords) // 4096)
        self.reduced_check_ = ctx.reduced_check(coords[::step], lp[::step])
        where = 'the final ensemble'
        if hasattr(self._sampler, 'rows_nearest_zero_logp'):
            rows, rlp = self._sampler.rows_nearest_zero_logp(256)
            if len(rows):
                self.reduced_check_shell_ = ctx.reduced_check(rows, rlp)
                if self.reduced_check_shell_ > self.reduced_check_:
                    self.reduced_check_, where = self.reduced_check_shell_, 'the stored samples nearest to logp = 0'
        if not self.reduced_check_ <= self._LOGP_TOL:
            warnings.warn(f'the {ctx.variant!r} kernel is {self.reduced_check_:.1e} (relative) away from the exact '
                          f'log-probability on {where} (tolerance {self._LOGP_TOL:.0e}); '
                          "refit with variant='reduced_comp' (or 'collapsed')", RuntimeWarning)

    def get_chain(self, **kwargs):
        """MCMC chain; kwargs ``discard``, ``thin``, ``flat`` as in emcee."""
        self._check_if_fitted()
        return self._sampler.get_chain(**kwargs)

    # -- properties (reference: src/bisip/models.py:139-179) --------------------------------
    @property
    def p0(self):
        return self._p0

    @property
    def params(self):
        return self._params

    @params.setter
    def params(self, var):
        self._params = var

    @property
    def sampler(self):
        self._check_if_fitted()
        return self._sampler

    @property
    def data(self):
        return self._data

    @property
    def fitted(self):
        return self.__fitted

    @property
    def param_names(self):
        return list(self.params.keys())

    @property
    def param_bounds(self):
        return np.array(list(self.params.values()), dtype=np.float64).T


class PolynomialDecomposition(Inversion):
    """Debye / Warburg polynomial decomposition (reference: src/bisip/models.py:182-229).

    Args:
        poly_deg (int): polynomial degree. Defaults to 5.
        c_exp (float): fixed Cole-Cole exponent, 1.0 Debye, 0.5 Warburg. Defaults to 1.0.
        variant (str): kernel formulation, 'auto' | 'reduced' | 'collapsed' | 'faithful'.
    """

    _model_id = _hip.MODEL_POLYDECOMP

    def __init__(self, *args, poly_deg=5, c_exp=1.0, variant='auto', **kwargs):
        super().__init__(*args, **kwargs)
        self.c_exp = c_exp
        self.poly_deg = poly_deg
        self.variant = variant

        # relaxation-time grid: one decade beyond the period range, 2N points
        period = np.log10(1. / self._data['w'])
        min_tau = np.floor(min(period) - 1)
        max_tau = np.floor(max(period) + 1)
        n_tau = 2 * self._data['N']
        self.log_tau = np.linspace(min_tau, max_tau, n_tau)
        deg_range = list(range(self.poly_deg + 1))
        self.log_taus = np.array([self.log_tau ** i for i in deg_range])
        self.taus = 10 ** self.log_tau

        self.params.update({'r0': [0.9, 1.1]})
        self.params.update({f'a{x}': [-1, 1] for x in deg_range})

    def _desc(self):
        return dict(poly_deg=self.poly_deg, c_exp=float(self.c_exp), taus=self.taus,
                    log_taus=self.log_taus, variant=self.variant)

    def forward(self, theta, w=None):
        """Impedance for theta = (R0, a_0, ..., a_P) [ascending]; (2,N) or (n,2,N)."""
        return self._forward_rows(theta, w)


class PeltonColeCole(Inversion):
    """Generalised (multi-mode) Pelton Cole-Cole model
    (reference: src/bisip/models.py:232-271).

    Args:
        n_modes (int): number of Cole-Cole modes. Defaults to 1.
    """

    _model_id = _hip.MODEL_COLECOLE

    def __init__(self, *args, n_modes=1, **kwargs):
        super().__init__(*args, **kwargs)
        self.n_modes = n_modes
        modes = list(range(self.n_modes))
        self.params.update({'r0': [0.9, 1.1]})
        self.params.update({f'm{i+1}': [0.0, 1.0] for i in modes})
        self.params.update({f'log_tau{i+1}': [-15, 5] for i in modes})
        self.params.update({f'c{i+1}': [0.0, 1.0] for i in modes})

    def _desc(self):
        return dict(n_modes=self.n_modes)

    def forward(self, theta, w=None):
        """theta = (R0, m_1..m_D, log_tau_1..D [natural log], c_1..D)."""
        return self._forward_rows(theta, w)


ColeCole = PeltonColeCole  # the name BASELINE.json's north star uses


class Dias2000(Inversion):
    """Dias (2000) model (reference: src/bisip/models.py:274-305)."""

    _model_id = _hip.MODEL_DIAS2000

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.params.update({'r0': [0.9, 1.1],
                            'm': [0, 1],
                            'log_tau': [-20, 0],
                            'eta': [0, 150],
                            'delta': [0, 1]})

    def forward(self, theta, w=None):
        """theta = (R0, m, log_tau, eta, delta)."""
        return self._forward_rows(theta, w)


class Shin2015(Inversion):
    """Shin (2015) model (reference: src/bisip/models.py:308-349; flagged by its
    authors as yielding unexpected results, reproduced as written)."""

    _model_id = _hip.MODEL_SHIN2015

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.params.update({'R1': [0.0, 1.0],
                            'R2': [0.0, 1.0],
                            'log_Q1': [-15, -13],
                            'log_Q2': [-7, -5],
                            'n1': [0, 1],
                            'n2': [0, 1]})

    def forward(self, theta, w=None):
        """theta = (R1, R2, log_Q1, log_Q2, n1, n2)."""
        return self._forward_rows(theta, w)
