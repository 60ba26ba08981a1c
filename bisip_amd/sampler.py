"""Affine-invariant ensemble sampler driving the vectorised GPU log-probability.

The reference delegates sampling to ``emcee.EnsembleSampler`` (call sites
src/bisip/models.py:111-118, 137); emcee is third-party, unpinned
(requirements.txt:2) and not part of the reference tree.  This module restates the
published algorithm emcee runs by default -- Goodman & Weare's stretch move with a
red/blue split -- around ONE difference: the log-probability
is evaluated for a whole half-ensemble per call (``vectorize=True`` in emcee's
terms), which is what the HIP kernels consume.

RNG contract (so runs are reproducible and shardable): a private
``numpy.random.RandomState`` seeded from the global NumPy state at construction
(which is why ``np.random.seed(42)`` before ``fit()`` pins a run, as in the
reference's notebooks).  Per iteration it is consumed in this order: one
``choice`` over the move list, one ``shuffle`` of the split labels, then for each
of the two halves ``rand(Ns)`` (stretch factors), ``randint(Nc, size=Ns)``
(partners) and one ``rand()`` per walker of the half (accept test).

Sampler parity with emcee itself is UNPINNED (no emcee here, and the reference's
tests assert nothing at this boundary); tests pin this implementation against its
own NumPy/oracle replay instead.
"""

import numpy as np

from .dist import all_gather_rows, shard_range


def walkers_independent(coords):
    """True when the initial ensemble spans the parameter space (no degenerate
    directions): condition number of the centred, column-scaled positions <= 1e8."""
    coords = np.asarray(coords, dtype=np.float64)
    if not np.all(np.isfinite(coords)):
        return False
    c = coords - coords.mean(axis=0)[None, :]
    colmax = np.abs(c).max(axis=0)
    if np.any(colmax == 0):
        return False
    c = c / colmax
    c = c / np.sqrt((c ** 2).sum(axis=0))
    return np.linalg.cond(c) <= 1e8


class EnsembleSampler:
    """Stretch-move ensemble sampler with the subset of emcee's interface that the
    reference uses: ``run_mcmc``, ``get_chain``, ``get_log_prob``,
    ``acceptance_fraction``.

    Args:
        nwalkers, ndim: ensemble shape.
        log_prob_fn: vectorised callable ``theta (n, ndim) -> logp (n,)``.
        a: stretch scale (emcee default 2.0).
        args: extra positional arguments appended to every ``log_prob_fn`` call.
        distributed, group: shard the evaluations over a ``torch.distributed`` group
            (see bisip_amd/dist.py); the chain is bit-identical to a single-rank run.
    """

    def __init__(self, nwalkers, ndim, log_prob_fn, a=2.0, args=None, pool=None, moves=None,
                 live_dangerously=False, group=None, distributed=False):
        if moves is not None:
            raise NotImplementedError('only the default StretchMove is implemented natively; '
                                      'install emcee to use other moves')
        self.nwalkers = int(nwalkers)
        self.ndim = int(ndim)
        self.log_prob_fn = log_prob_fn
        self.args = tuple(args) if args is not None else ()
        self.a = float(a)
        self.live_dangerously = live_dangerously
        # distributed=True: shard every half-step's log-prob evaluations over the ranks of
        # `group` (all ranks must construct the sampler with the same NumPy RNG state)
        self._group = group
        self._world, self._rank = 1, 0
        if distributed:
            import torch.distributed as dist
            if dist.is_initialized():
                self._world = dist.get_world_size(group)
                self._rank = dist.get_rank(group)
        self._random = np.random.mtrand.RandomState()
        self._random.set_state(np.random.get_state())
        self.reset()

    def reset(self):
        self.iteration = 0
        self._chain = np.empty((0, self.nwalkers, self.ndim))
        self._log_prob = np.empty((0, self.nwalkers))
        self._accepted = np.zeros(self.nwalkers)
        self._coords = None
        self._lp = None

    # -- log-probability ----------------------------------------------------------------
    def compute_log_prob(self, coords):
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        if np.any(np.isinf(coords)):
            raise ValueError('At least one parameter value was infinite')
        if np.any(np.isnan(coords)):
            raise ValueError('At least one parameter value was NaN')
        lp = np.asarray(self.log_prob_fn(coords, *self.args), dtype=np.float64)
        if lp.shape != (coords.shape[0],):
            raise ValueError(f'log_prob_fn returned shape {lp.shape}, expected ({coords.shape[0]},)')
        if np.any(np.isnan(lp)):
            raise ValueError('Probability function returned NaN')
        return lp

    # -- one iteration ------------------------------------------------------------------
    def _stretch_iteration(self):
        rng = self._random
        W, ndim = self.nwalkers, self.ndim
        rng.choice(1)  # the draw over the (single-entry) move list
        all_inds = np.arange(W)
        inds = all_inds % 2
        rng.shuffle(inds)
        accepted = np.zeros(W, dtype=bool)
        for split in (0, 1):
            S1 = inds == split
            s = self._coords[S1]
            c = self._coords[~S1]  # read after the other half's update
            Ns, Nc = len(s), len(c)
            zz = ((self.a - 1.0) * rng.rand(Ns) + 1) ** 2.0 / self.a
            factors = (ndim - 1.0) * np.log(zz)
            rint = rng.randint(Nc, size=(Ns,))
            q = c[rint] - (c[rint] - s) * zz[:, None]
            u = rng.rand(Ns)  # one uniform per walker, in walker order
            # this rank's block of the active half (the whole half when not distributed)
            lo, hi = shard_range(Ns, self._world, self._rank)
            new_lp = self.compute_log_prob(q[lo:hi])
            old_lp = self._lp[S1][lo:hi]
            with np.errstate(divide='ignore'):
                acc = factors[lo:hi] + new_lp - old_lp > np.log(u[lo:hi])
            block = np.concatenate([np.where(acc[:, None], q[lo:hi], s[lo:hi]),
                                    np.where(acc, new_lp, old_lp)[:, None],
                                    acc[:, None].astype(np.float64)], axis=1)
            if self._world > 1:  # one all-gather of the just-updated rows per half-step
                block = all_gather_rows(block, Ns, self._group)
            idx = all_inds[S1]
            self._coords[idx] = block[:, :ndim]
            self._lp[idx] = block[:, ndim]
            accepted[idx] = block[:, ndim + 1] > 0
        self._accepted += accepted

    def run_mcmc(self, initial_state, nsteps, progress=False, **kwargs):
        """Advance the ensemble ``nsteps`` iterations from ``initial_state`` (W, ndim);
        pass ``None`` to continue from the last position."""
        if initial_state is None:
            if self._coords is None:
                raise ValueError('Cannot have `initial_state=None` if run_mcmc has never been called.')
        else:
            p0 = np.array(initial_state, dtype=np.float64, copy=True)
            if p0.shape != (self.nwalkers, self.ndim):
                raise ValueError(f'incompatible input dimensions {p0.shape}')
            if not self.live_dangerously and self.nwalkers < 2 * self.ndim:
                raise RuntimeError('It is unadvisable to use a red-blue move with fewer walkers '
                                   'than twice the number of dimensions.')
            if not self.live_dangerously and not walkers_independent(p0):
                raise ValueError('Initial state has a large condition number. Make sure that '
                                 'your walkers are linearly independent for the best performance')
            self._coords = p0
            self._lp = self.compute_log_prob(p0)
        nsteps = int(nsteps)
        chain = np.empty((nsteps, self.nwalkers, self.ndim))
        logp = np.empty((nsteps, self.nwalkers))
        it = range(nsteps)
        if progress:
            try:
                from tqdm import tqdm
                it = tqdm(it, total=nsteps)
            except ImportError:
                pass
        for i in it:
            self._stretch_iteration()
            chain[i] = self._coords
            logp[i] = self._lp
        self._chain = np.concatenate([self._chain, chain], axis=0)
        self._log_prob = np.concatenate([self._log_prob, logp], axis=0)
        self.iteration += nsteps
        return self._coords.copy(), self._lp.copy()

    # -- chain access (emcee backend semantics: chain[discard+thin-1 : iteration : thin]) ----
    def _get_value(self, arr, discard=0, thin=1, flat=False):
        if self.iteration <= 0:
            raise AttributeError('you must run the sampler before accessing the results')
        v = arr[discard + thin - 1:self.iteration:thin]
        if flat:
            v = v.reshape((-1,) + v.shape[2:])
        return v

    def get_chain(self, discard=0, thin=1, flat=False):
        return self._get_value(self._chain, discard, thin, flat)

    def get_log_prob(self, discard=0, thin=1, flat=False):
        return self._get_value(self._log_prob, discard, thin, flat)

    @property
    def acceptance_fraction(self):
        return self._accepted / float(self.iteration)

    @property
    def random_state(self):
        return self._random.get_state()
