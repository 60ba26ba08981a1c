"""Batch of independent spectra (BASELINE config 5): E spectra x Wp walkers each.

The reference inverts one data file per ``Inversion`` object (src/bisip/models.py:41-57)
and a survey of many spectra is a Python loop over objects; here the E spectra share one
device context (``bisip_batch_create``) so one launch evaluates all E*Wp walkers and one
launch per half-step advances all E ensembles.  Ensembles are independent: across GPUs
they shard as whole replicas (spectra [start, stop) per rank, no collective).
"""

import numpy as np

from . import _hip
from .dist import shard_range
from .sampler import DeviceEnsembleSampler
from .utils import load_data_batch

_MODELS = {
    'PolynomialDecomposition': _hip.MODEL_POLYDECOMP,
    'PeltonColeCole': _hip.MODEL_COLECOLE,
    'ColeCole': _hip.MODEL_COLECOLE,
    'Dias2000': _hip.MODEL_DIAS2000,
    'Shin2015': _hip.MODEL_SHIN2015,
}


def default_params(model, n_modes=1, poly_deg=5):
    """Parameter names and prior box of a model class (same as the reference's
    __init__ dictionaries, src/bisip/models.py:211-213, 247-252, 287-291, 325-331)."""
    if model == 'PolynomialDecomposition':
        p = {'r0': [0.9, 1.1]}
        p.update({f'a{x}': [-1, 1] for x in range(poly_deg + 1)})
    elif model in ('PeltonColeCole', 'ColeCole'):
        p = {'r0': [0.9, 1.1]}
        p.update({f'm{i+1}': [0.0, 1.0] for i in range(n_modes)})
        p.update({f'log_tau{i+1}': [-15, 5] for i in range(n_modes)})
        p.update({f'c{i+1}': [0.0, 1.0] for i in range(n_modes)})
    elif model == 'Dias2000':
        p = {'r0': [0.9, 1.1], 'm': [0, 1], 'log_tau': [-20, 0], 'eta': [0, 150], 'delta': [0, 1]}
    elif model == 'Shin2015':
        p = {'R1': [0.0, 1.0], 'R2': [0.0, 1.0], 'log_Q1': [-15, -13], 'log_Q2': [-7, -5],
             'n1': [0, 1], 'n2': [0, 1]}
    else:
        raise ValueError(f'unknown model {model!r}')
    return p


class SpectraBatch:
    """E spectra inverted together with the same model class.

    Args:
        model (str): 'PolynomialDecomposition', 'PeltonColeCole', 'Dias2000' or 'Shin2015'.
        spectra: list of file paths or of raw (N,5) tables [freq, amp, pha, amp_err, pha_err];
            all must have the same number of frequencies.
        nwalkers (int): walkers per spectrum (even). nsteps (int): MCMC steps.
        n_modes / poly_deg / c_exp: model options as in the reference classes.
        rank, world: keep only this rank's block of spectra (whole-replica sharding).
    """

    def __init__(self, model, spectra, nwalkers=256, nsteps=1000, headers=1, ph_units='mrad',
                 n_modes=1, poly_deg=5, c_exp=1.0, device=0, rank=0, world=1):
        if model not in _MODELS:
            raise ValueError(f'unknown model {model!r}')
        self.model = 'PeltonColeCole' if model == 'ColeCole' else model
        lo, hi = shard_range(len(spectra), world, rank)
        self.spectrum_range = (lo, hi)
        self.n_spectra_total = len(spectra)
        if hi <= lo:
            raise ValueError('no spectra for this rank')
        batch = load_data_batch(spectra[lo:hi], headers, ph_units)
        self.n_spectra = hi - lo
        self.N = batch['N']
        self.w, self.zn, self.zn_err = batch['w'], batch['zn'], batch['zn_err']
        self.norm_factor = batch['norm_factor']
        self.nwalkers, self.nsteps = int(nwalkers), int(nsteps)
        self.n_modes, self.poly_deg, self.c_exp = n_modes, poly_deg, c_exp
        self.params = default_params(self.model, n_modes, poly_deg)
        kw = {}
        if self.model == 'PolynomialDecomposition':
            # one tau grid for the whole batch, from the union of the frequency ranges
            # (identical to the reference's per-file grid when the spectra share w)
            period = np.log10(1. / self.w)
            self.log_tau = np.linspace(np.floor(period.min() - 1), np.floor(period.max() + 1), 2 * self.N)
            self.log_taus = np.array([self.log_tau ** i for i in range(poly_deg + 1)])
            self.taus = 10 ** self.log_tau
            kw = dict(poly_deg=poly_deg, c_exp=float(c_exp), taus=self.taus, log_taus=self.log_taus)
        elif self.model == 'PeltonColeCole':
            kw = dict(n_modes=n_modes)
        self.ctx = _hip.HipContext(_MODELS[self.model], self.w, self.zn, self.zn_err,
                                   self.param_bounds, device=device, **kw)
        self.ctx.set_spectrum_offset(lo)     # chains do not depend on how the survey is split over ranks
        self._sampler = None

    @property
    def param_names(self):
        return list(self.params.keys())

    @property
    def param_bounds(self):
        return np.array(list(self.params.values()), dtype=np.float64).T

    @property
    def ndim(self):
        return self.param_bounds.shape[1]

    def log_prob(self, theta):
        """theta (E, n, ndim) -> logp (E, n): n walkers of every spectrum in one launch."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        if theta.ndim != 3 or theta.shape[0] != self.n_spectra or theta.shape[2] != self.ndim:
            raise ValueError(f'theta must have shape ({self.n_spectra}, n, {self.ndim})')
        self.ctx.set_bounds(self.param_bounds)
        return self.ctx.logprob(theta.reshape(-1, self.ndim)).reshape(theta.shape[:2])

    def forward(self, theta):
        """theta (E, n, ndim) -> Z (E, n, 2, N)."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        Z = self.ctx.forward(theta.reshape(-1, self.ndim))
        return Z.reshape(theta.shape[0], theta.shape[1], 2, self.N)

    def fit(self, p0=None, seed=None, thin_by=1, chain='host', persistent=None):
        """Run E independent stretch-move ensembles on the device (rng='philox').

        ``thin_by``: store one sample every ``thin_by`` iterations (``nsteps`` samples are
        stored).  ``chain='device'`` keeps the stored samples in HBM: ``get_param_mean`` /
        ``get_param_std`` then summarise them on the device and ``get_chain`` copies them to
        the host only when called.  ``persistent``: one workgroup per spectrum runs all
        iterations of a chunk in one launch (same chain; default: when the ensembles fit a
        workgroup and the batch fills the chip, see DeviceEnsembleSampler)."""
        if chain not in ('host', 'device'):
            raise ValueError("chain must be 'host' or 'device'")
        E, Wp, ndim = self.n_spectra, self.nwalkers, self.ndim
        if p0 is None:
            # the whole survey's starts from the global RNG, this rank's block kept: with the same
            # np.random.seed on every rank a spectrum starts (and runs) the same on any number of GPUs
            first, last = self.spectrum_range
            p0 = np.random.uniform(*self.param_bounds, (self.n_spectra_total, Wp, ndim))[first:last]
        self.ctx.set_bounds(self.param_bounds)
        self._sampler = DeviceEnsembleSampler(Wp, ndim, self.ctx, rng='philox', seed=seed,
                                              n_ensembles=E, chain_on_device=(chain == 'device'),
                                              persistent=persistent)
        self._sampler.run_mcmc(np.asarray(p0).reshape(E * Wp, ndim), self.nsteps, thin_by=thin_by)
        # PolynomialDecomposition: the reduced kernels were chosen from an estimate; measure them on
        # the final ensembles against long double (see Inversion._check_reduced_kernel)
        self.reduced_check_ = None
        if self.ctx.variant in ('reduced', 'reduced_comp'):
            self.reduced_check_ = self.ctx.reduced_check(self._sampler._coords, self._sampler._lp)
            if not self.reduced_check_ <= 1e-10:
                import warnings
                warnings.warn(f'the {self.ctx.variant!r} kernel is {self.reduced_check_:.1e} (relative) away from the '
                              'exact log-probability on the final ensembles (tolerance 1e-10)', RuntimeWarning)
        return self

    def _moments(self, discard, thin):
        if self._sampler is None:
            raise AssertionError('Model is not fitted!')
        if self._sampler.chain_on_device:
            return self._sampler.param_moments(discard=discard, thin=thin)
        flat = self.get_chain(discard=discard, thin=thin, flat=True)   # (E, n, ndim)
        return flat.mean(axis=1), flat.std(axis=1)

    def get_param_mean(self, discard=0, thin=1):
        """Posterior mean of every parameter of every spectrum, ``(E, ndim)`` -- per spectrum
        what the reference's ``get_param_mean`` returns (src/bisip/utils.py:55-69)."""
        return self._moments(discard, thin)[0]

    def get_param_percentile(self, p=(2.5, 50, 97.5), discard=0, thin=1):
        """Percentiles of every parameter of every spectrum, ``(len(p), E, ndim)`` -- per spectrum
        what the reference's ``get_param_percentile`` returns (src/bisip/utils.py:37-53)."""
        if self._sampler is None:
            raise AssertionError('Model is not fitted!')
        if self._sampler.chain_on_device:
            return self._sampler.param_percentiles(p, discard=discard, thin=thin)
        flat = self.get_chain(discard=discard, thin=thin, flat=True)   # (E, n, ndim)
        return np.percentile(flat, p, axis=1)

    def get_model_percentile(self, p=(2.5, 50, 97.5), discard=0, thin=1):
        """Percentiles of the MODEL response over every spectrum's chain, ``(len(p), E, 2, N)`` -- per
        spectrum what the reference's ``get_model_percentile`` returns (src/bisip/utils.py:17-35): the
        band a fit is plotted with.  On the device, many spectra per pass: one forward launch that
        writes the responses column by column, one selection of the order statistics; with
        ``chain='device'`` the chain never leaves HBM."""
        import torch
        if self._sampler is None:
            raise AssertionError('Model is not fitted!')
        s = self._sampler
        be = s.backend
        E, Wp, ndim, N = self.n_spectra, self.nwalkers, self.ndim, self.N
        p = np.atleast_1d(np.asarray(p, dtype=np.float64))
        discard, thin = int(discard), int(thin)
        if s.chain_on_device:
            t = s.device_chain()
        else:
            t = torch.from_numpy(s.get_chain()).to(be.device)
        first = discard + thin - 1
        used = t[first::thin] if thin >= 1 else t[:0]
        n = int(used.shape[0])
        if thin < 1 or discard < 0 or n < 1:
            raise ValueError(f'no samples left with discard={discard}, thin={thin} of {int(t.shape[0])} stored')
        rows_per = n * Wp
        cols = 2 * N
        # spectra per pass: the responses of a pass stay under ~8 GB
        G = int(min(E, max(1, (8 << 30) // (rows_per * cols * 8))))
        out = np.empty((p.size, E, cols))
        grid = used.reshape(n, E, Wp, ndim)
        for g0 in range(0, E, G):
            g1 = min(E, g0 + G)
            k = g1 - g0
            rows = grid[:, g0:g1].permute(1, 0, 2, 3).reshape(k, rows_per, ndim)   # one copy: spectrum-major
            Zc = be.empty((k, cols, rows_per), torch.float64)                      # one column per (spectrum, part, frequency)
            self.ctx.forward_columns_dev(g0, k, rows.data_ptr(), k * rows_per, Zc.data_ptr(), be.stream())
            res = be.empty((p.size, k * cols), torch.float64)
            _hip.columns_percentiles_dev(Zc.data_ptr(), k * cols, rows_per, p, res.data_ptr(), be.stream())
            be.synchronize()
            out[:, g0:g1] = res.cpu().numpy().reshape(p.size, k, cols)
            del rows, Zc, res
        return out.reshape(p.size, E, 2, N)

    def get_param_std(self, discard=0, thin=1):
        """Posterior standard deviation, ``(E, ndim)`` (src/bisip/utils.py:71-85)."""
        return self._moments(discard, thin)[1]

    def get_chain(self, discard=0, thin=1, flat=False):
        """(nsteps', E, Wp, ndim); flat=True -> (E, nsteps'*Wp, ndim)."""
        if self._sampler is None:
            raise AssertionError('Model is not fitted!')
        ch = self._sampler.get_chain(discard=discard, thin=thin)
        ch = ch.reshape(ch.shape[0], self.n_spectra, self.nwalkers, self.ndim)
        if flat:
            ch = ch.transpose(1, 0, 2, 3).reshape(self.n_spectra, -1, self.ndim)
        return ch

    def get_log_prob(self, discard=0, thin=1):
        if self._sampler is None:
            raise AssertionError('Model is not fitted!')
        lp = self._sampler.get_log_prob(discard=discard, thin=thin)
        return lp.reshape(lp.shape[0], self.n_spectra, self.nwalkers)

    @property
    def acceptance_fraction(self):
        if self._sampler is None:
            raise AssertionError('Model is not fitted!')
        return self._sampler.acceptance_fraction.reshape(self.n_spectra, self.nwalkers)

    def gather(self, per_spectrum, group=None):
        """The end of a multi-GPU survey: every rank passes a per-spectrum result of ITS block --
        ``(n_spectra, ...)``, e.g. ``get_param_mean()`` -- and gets the whole survey's
        ``(n_spectra_total, ...)`` in spectrum order.  The only collective of the batch path (the
        ensembles never exchange anything while they run); a single process gets its input back.
        A leading percentile axis is the caller's to move: ``gather(np.moveaxis(pct, 1, 0))``."""
        from .dist import all_gather_rows
        a = np.ascontiguousarray(per_spectrum, dtype=np.float64)
        if a.ndim < 1 or a.shape[0] != self.n_spectra:
            raise ValueError(f'expected {self.n_spectra} rows (one per spectrum of this rank), got {a.shape}')
        flat = all_gather_rows(a.reshape(self.n_spectra, -1), self.n_spectra_total, group)
        return flat.reshape((flat.shape[0],) + a.shape[1:])

    def close(self):
        """Release the device context."""
        self.ctx.close()
