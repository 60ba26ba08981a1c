"""Affine-invariant ensemble samplers around the GPU log-probability.

The reference delegates sampling to ``emcee.EnsembleSampler`` (call sites
src/bisip/models.py:111-118, 137); emcee is third-party, unpinned
(requirements.txt:2) and not part of the reference tree.  This module restates the
published algorithm emcee runs by default -- Goodman & Weare's stretch move with a
red/blue split -- in two drivers that produce THE SAME CHAIN:

* ``EnsembleSampler``        host loop; the log-probability of each half-ensemble is one
                             vectorised call (emcee's ``vectorize=True`` contract).
* ``DeviceEnsembleSampler``  ensemble, chain and the whole half-step (proposal,
                             log-probability, accept, update) stay on the GPU: one
                             kernel launch per half-step through the C ABI
                             (``bisip_stretch_half_dev``), or eval + RCCL all-gather +
                             apply when walkers are sharded over several GPUs.

RNG contract (reproducible and shardable): a private ``numpy.random.RandomState``
seeded from the global NumPy state at construction (so ``np.random.seed(42)`` before
``fit()`` pins a run, as in the reference's notebooks).  Per iteration it is consumed
in this order: one uniform double (emcee's weighted ``choice`` over its move list draws one even
when the list has a single entry), one ``shuffle`` of the split labels,
then for each of the two halves ``rand(Ns)`` (stretch factors), ``randint(Nc, Ns)``
(partners) and ``rand(Ns)`` (accept uniforms, walker order).  ``draw_step`` states that
order; ``bisip_numpy_stretch_stream`` (csrc/host_rng.cpp) replays it in C for whole chunks,
bit for bit, RandomState included (checked against NumPy in tests/test_host_logic.py).

``rng='philox'`` (DeviceEnsembleSampler only) removes the host from the loop: the same
per-slot quantities come from a counter-based Philox4x32-10 stream generated on the
device (contract in bisip_amd/csrc/sampler_kernels.h: key = seed, counter = (slot,
step, half, purpose)); only the per-step split -- an affine bijection (A, B) of the
walker indices, ``affine_splits``, itself a pure function of (seed, step) -- is computed
on the host.

Sampler parity with emcee itself is UNPINNED (no emcee here, and the reference's
tests assert nothing at this boundary); tests pin the two drivers against each other
and against an oracle-driven replay.
"""

import os

import numpy as np

from .dist import all_gather_rows, shard_range


class _one_blas_thread(object):
    """The Gram product of a block on ONE thread.  A threaded BLAS wakes its whole pool for a product of a
    few hundred microseconds and the pool then spins for milliseconds: inside a container's CPU quota that
    stalls the process (measured: 10 ms in the first device allocation after the product -- the kind of
    stall utils.respect_cpu_quota describes).  Without threadpoolctl (or when it fails in any way): einsum's
    own loops, no BLAS.  The limit is process-wide while it holds, so entries are serialised by a lock (the
    restore of one thread must not undo the limit of another), and the controller is rebuilt when libraries
    were loaded since it was made (a BLAS / OpenMP runtime that arrived later -- torch's -- is then limited too)."""

    _lock = None
    _controller = None      # NumPy's BLAS is mapped long before the first ensemble: looked up once (0.4 ms)
    _n_libs = -1

    def __enter__(self):
        import threading
        cls = _one_blas_thread
        if cls._lock is None:
            cls._lock = threading.Lock()
        self._ctx = None
        cls._lock.acquire()
        try:
            import sys
            import threadpoolctl
            n_libs = len(sys.modules)          # a cheap proxy for "something new may have been mapped"
            if cls._controller is None or n_libs != cls._n_libs:
                cls._controller, cls._n_libs = threadpoolctl.ThreadpoolController(), n_libs
            self._ctx = cls._controller.limit(limits=1, user_api='blas')
            return lambda b: b.T @ b
        except Exception:                      # no threadpoolctl, one older than its controller class, a library it cannot read
            return lambda b: np.einsum('ij,ik->jk', b, b)

    def __exit__(self, *exc):
        try:
            if self._ctx is not None:
                self._ctx.restore_original_limits()
        finally:
            _one_blas_thread._lock.release()
        return False


def gram_decides(g, nwalkers):
    """True when the centred Gram matrix ``g`` of ``nwalkers`` positions settles emcee's test in favour of the
    ensemble: the correlation matrix's eigenvalue ratio is far inside the limit (see walkers_independent).
    False says nothing -- non-finite, degenerate or merely not clear-cut: the singular values then decide."""
    with np.errstate(all='ignore'):
        d = np.sqrt(np.diag(g))
        if np.all(np.isfinite(g)) and np.all(d > 1e-150):
            lam = np.linalg.eigvalsh(g / np.outer(d, d))
            return bool(lam[-1] > 0 and lam[0] / lam[-1] >= max(1e-8, 1e4 * nwalkers * np.finfo(np.float64).eps))
    return False


def gram_from_shifted_sums(sums, nwalkers, ndim):
    """The centred Gram matrix from what bisip_ensemble_gram_dev returns: S (ndim,) and the upper triangle of P,
    both of the positions shifted by walker 0:  G = P - S S^T / W."""
    S = np.asarray(sums[:ndim], dtype=np.float64)
    P = np.zeros((ndim, ndim))
    P[np.triu_indices(ndim)] = sums[ndim:]
    P = P + np.triu(P, 1).T
    with np.errstate(all='ignore'):           # (non-finite sums: the caller's decision sends them to the exact test)
        return P - np.outer(S, S) / float(nwalkers)


def walkers_independent(coords):
    """True when the initial ensemble spans the parameter space (no degenerate
    directions): condition number of the centred, column-scaled positions <= 1e8 (emcee's test).

    The singular values of a (W, ndim) matrix cost more than hundreds of half-steps once W is
    in the tens of thousands (3 ms at 32,768 walkers, 150 ms at a million: run_mcmc's
    ``timing['check_s']``).  The columns scaled to unit length have the correlation matrix
    R = D c^T c D as their Gram matrix and cond = sqrt(lambda_max / lambda_min) of it -- one
    matrix product.  A Gram matrix squares the condition number, so it DECIDES only where
    that costs nothing: lambda_min / lambda_max above 1e4 W eps (a condition number below
    ~1e3 at a million walkers, ~5e3 at 32,768; ensembles drawn in a ball or a box have
    1-100) is far inside the limit whatever the rounding of the product.  Everything else --
    near-degenerate, degenerate, non-finite or under/overflowing products -- takes the
    singular values as before, so the answer is emcee's in every case.

    INVARIANT callers rely on (DeviceEnsembleSampler._start_from uploads without a second finiteness pass):
    True implies every coordinate is finite -- a NaN or an inf makes the mean, hence the Gram matrix,
    non-finite, which sends the decision to the slow branch and its isfinite test
    (tests/test_gpu_sampler.py::test_non_finite_start_of_a_big_ensemble_is_refused)."""
    coords = np.asarray(coords, dtype=np.float64)
    if coords.ndim == 2 and coords.shape[0] >= 64 * max(coords.shape[1], 1):
        with np.errstate(all='ignore'):
            # block by block (a block stays in cache between its centring and its product); a NaN or an inf
            # anywhere makes the mean, and with it g, non-finite: no separate pass for them here
            mean = coords.mean(axis=0)
            g = np.zeros((coords.shape[1], coords.shape[1]))
            with _one_blas_thread() as gram:
                for lo in range(0, coords.shape[0], 16384):
                    blk = coords[lo:lo + 16384] - mean
                    g += gram(blk)
            if gram_decides(g, coords.shape[0]):
                return True
    if not np.all(np.isfinite(coords)):
        return False
    c = coords - coords.mean(axis=0)[None, :]
    colmax = np.abs(c).max(axis=0)
    if np.any(colmax == 0):
        return False
    c = c / colmax
    c = c / np.sqrt((c ** 2).sum(axis=0))
    return np.linalg.cond(c) <= 1e8


def draw_step(rng, nwalkers, ndim, a=2.0):
    """Consume the RNG for ONE iteration and return the two half-steps.

    Each half is a dict of arrays over its Ns slots: ``active`` (walker moved by the
    slot), ``partner`` (walker of the complementary half it stretches from), ``zz``
    (stretch factor), ``factor`` = (ndim-1) ln zz and ``logu`` = ln u.
    """
    # emcee picks the move with ``random.choice(moves, p=weights)``: with ``p`` given, the
    # legacy RandomState draws ONE uniform double even for a single-entry list
    rng.random_sample()
    all_inds = np.arange(nwalkers)
    inds = all_inds % 2
    rng.shuffle(inds)
    halves = []
    for split in (0, 1):
        S1 = inds == split
        active = all_inds[S1]
        comp = all_inds[~S1]
        Ns, Nc = len(active), len(comp)
        zz = ((a - 1.0) * rng.rand(Ns) + 1) ** 2.0 / a
        rint = rng.randint(Nc, size=(Ns,))
        u = rng.rand(Ns)
        with np.errstate(divide='ignore'):
            logu = np.log(u)
        halves.append(dict(active=active, partner=comp[rint], zz=zz,
                           factor=(ndim - 1.0) * np.log(zz), logu=logu))
    return halves


def _mod_inverse(a, W):
    """Vectorised modular inverse of a (coprime to W) by the extended Euclid iteration."""
    a = np.asarray(a, dtype=np.int64)
    r0, r1 = np.full_like(a, W), a % W
    t0, t1 = np.zeros_like(a), np.ones_like(a)
    while np.any(r1 != 0):
        live = r1 != 0
        q = np.where(live, r0 // np.where(live, r1, 1), 0)
        r0, r1 = np.where(live, r1, r0), np.where(live, r0 - q * r1, r1)
        t0, t1 = np.where(live, t1, t0), np.where(live, t0 - q * t1, t1)
    return t0 % W


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11) vectorised over uint64 arrays holding 32-bit
    words; the host twin of bisip_amd/csrc/philox.h (cross-checked in the tests)."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    mask = np.uint64(0xffffffff)
    c0, c1, c2, c3 = [np.asarray(x, dtype=np.uint64) & mask for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask, p1 & mask, \
                         ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask, p0 & mask
        k0 = (k0 + np.uint64(W0)) & mask
        k1 = (k1 + np.uint64(W1)) & mask
    return c0, c1, c2, c3


def affine_splits(seed, nwalkers, step0, nsteps):
    """Per-step random balanced split for rng='philox': pi(i) = (A*i + B) mod W with
    gcd(A, W) = 1; walker i belongs to half pi(i) & 1.  Returns int32 (nsteps, 3) rows
    (A, A^-1 mod W, B).  A pure function of (seed, step): for step k the philox block with
    counter (k, attempt, 0, 2) gives A = 1 + x0 mod (W-1) (first attempt coprime to W wins)
    and attempt 0's x1 gives B = x1 mod W -- so runs do not depend on how they are chunked."""
    W = int(nwalkers)
    k0, k1 = int(seed) & 0xffffffff, int(seed) >> 32
    steps = np.arange(step0, step0 + nsteps, dtype=np.uint64)
    x0, x1, _, _ = philox4x32_10(steps, 0, 0, 2, k0, k1)
    B = (x1 % np.uint64(W)).astype(np.int64)
    A = (1 + x0 % np.uint64(max(W - 1, 1))).astype(np.int64)
    bad = np.gcd(A, W) != 1
    attempt = 1
    while bad.any():
        y0, _, _, _ = philox4x32_10(steps[bad], attempt, 0, 2, k0, k1)
        A[bad] = (1 + y0 % np.uint64(max(W - 1, 1))).astype(np.int64)
        bad = np.gcd(A, W) != 1
        attempt += 1
    out = np.empty((nsteps, 3), dtype=np.int32)
    out[:, 0], out[:, 1], out[:, 2] = A, _mod_inverse(A, W), B
    return out


class _DeviceSlabs:
    """Stored samples that stay in device memory (one torch tensor per chunk).  Behaves like
    a chain part for the bookkeeping (``shape``) and becomes a host array the first time the
    host asks for it."""

    def __init__(self, tensors):
        self.tensors = list(tensors)

    @property
    def shape(self):
        return (sum(int(t.shape[0]) for t in self.tensors),) + tuple(self.tensors[0].shape[1:])

    def tensor(self):
        if len(self.tensors) > 1:
            import torch
            self.tensors = [torch.cat(self.tensors, dim=0)]
        return self.tensors[0]

    def materialize(self):
        """Host copy, made once; the samples also stay on the device for the summaries."""
        if getattr(self, '_host', None) is None:
            self._host = self.tensor().cpu().numpy()
        return self._host


def _host_parts(parts):
    """Chain parts as host arrays.  Device-resident parts are first merged into one (on the
    device) and stay what they are: get_chain() must not take the chain away from
    param_moments() / param_percentiles()."""
    if len(parts) > 1 and all(isinstance(p, _DeviceSlabs) for p in parts):
        parts[:] = [_DeviceSlabs(t for p in parts for t in p.tensors)]
    return [p.materialize() if isinstance(p, _DeviceSlabs) else p for p in parts]


class _SamplerBase:
    """Chain bookkeeping shared by the two drivers (emcee backend semantics)."""

    def __init__(self, nwalkers, ndim, a, live_dangerously, group, distributed):
        self.nwalkers = int(nwalkers)
        self.ndim = int(ndim)
        self.a = float(a)
        self.live_dangerously = live_dangerously
        # distributed=True: shard every half-step's evaluations over the ranks of `group`
        # (all ranks must construct the sampler with the same NumPy RNG state)
        self._group = group
        self._world, self._rank = 1, 0
        self._in_group = False      # True: collectives go through torch.distributed's `group`
        if distributed:
            import torch.distributed as dist
            if dist.is_initialized():
                self._world = dist.get_world_size(group)
                self._rank = dist.get_rank(group)
                self._in_group = True
        self._random = np.random.mtrand.RandomState()
        self._random.set_state(np.random.get_state())
        self._coords = None
        self._lp = None
        self.reset()

    def reset(self):
        """emcee's reset(): the stored samples and the acceptance counts go; the ensemble's current
        state stays, so ``run_mcmc(None, n)`` continues from it."""
        self.iteration = 0          # stored samples
        self._moves_done = 0        # stretch-move iterations performed (= stored * thin_by)
        self._chain_parts = []      # one (n, W, ndim) array per run_mcmc call
        self._log_prob_parts = []
        self._accepted = np.zeros(self.nwalkers)

    @staticmethod
    def _joined(parts, empty_shape):
        host = _host_parts(parts)
        if len(host) > 1:                 # host-resident parts of several runs: keep the joined array
            parts[:] = host = [np.concatenate(host, axis=0)]
        return host[0] if host else np.empty(empty_shape)

    @property
    def _chain(self):
        return self._joined(self._chain_parts, (0, self.nwalkers, self.ndim))

    @property
    def _log_prob(self):
        return self._joined(self._log_prob_parts, (0, self.nwalkers))

    _DEPENDENT = ('Initial state has a large condition number. Make sure that '
                  'your walkers are linearly independent for the best performance')

    def _check_initial(self, initial_state, copy=True, independence=True):
        """emcee's checks of a new initial state.  ``copy=False``: the caller only reads it (the device sampler
        uploads it); ``independence=False``: the caller runs that test itself (on the device, where a big
        ensemble is going anyway)."""
        p0 = np.array(initial_state, dtype=np.float64, copy=True) if copy else np.ascontiguousarray(initial_state, dtype=np.float64)
        if p0.shape != (self.nwalkers, self.ndim):
            raise ValueError(f'incompatible input dimensions {p0.shape}')
        if not self.live_dangerously and self.nwalkers < 2 * self.ndim:
            raise RuntimeError('It is unadvisable to use a red-blue move with fewer walkers '
                               'than twice the number of dimensions.')
        if independence and not self.live_dangerously and not walkers_independent(p0):
            raise ValueError(self._DEPENDENT)
        return p0

    @staticmethod
    def _check_coords(coords):
        if np.isfinite(coords).all():      # one pass in the usual case
            return
        if np.any(np.isinf(coords)):
            raise ValueError('At least one parameter value was infinite')
        raise ValueError('At least one parameter value was NaN')

    def _append(self, chain, logp):
        self._chain_parts.append(chain)
        self._log_prob_parts.append(logp)
        self.iteration += chain.shape[0]

    # chain access: chain[discard + thin - 1 : iteration : thin]
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

    def rows_nearest_zero_logp(self, k=256):
        """The k stored samples whose log-probability is nearest to zero, as (theta (k, ndim), logp (k,)) on
        the host: where the relative tolerance max(1, |logp|) has denominator 1 -- the rows of a run on which a
        kernel's absolute error shows (Inversion._check_reduced_kernel).  Chains kept in HBM are searched
        there; only the k rows come back."""
        parts_c, parts_l = self._chain_parts, self._log_prob_parts
        if not parts_c:
            return np.empty((0, self.ndim)), np.empty(0)
        if all(isinstance(p, _DeviceSlabs) for p in parts_l) and all(isinstance(p, _DeviceSlabs) for p in parts_c):
            import torch
            lp = torch.cat([p.tensor().reshape(-1) for p in parts_l])
            ch = torch.cat([p.tensor().reshape(-1, self.ndim) for p in parts_c])
            score = torch.where(torch.isfinite(lp), lp.abs(), torch.full_like(lp, float('inf')))
            idx = torch.topk(score, min(int(k), score.numel()), largest=False).indices
            return ch[idx].cpu().numpy(), lp[idx].cpu().numpy()
        lp = self.get_log_prob(flat=True)
        ch = self.get_chain(flat=True)
        score = np.where(np.isfinite(lp), np.abs(lp), np.inf)
        k = min(int(k), score.size)
        idx = np.argpartition(score, k - 1)[:k] if k else np.empty(0, dtype=int)
        return np.ascontiguousarray(ch[idx]), lp[idx]

    @property
    def acceptance_fraction(self):
        return self._accepted / float(self._moves_done)

    @property
    def random_state(self):
        return self._random.get_state()


class EnsembleSampler(_SamplerBase):
    """Host-loop stretch-move sampler with the subset of emcee's interface that the
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
        self.log_prob_fn = log_prob_fn
        self.args = tuple(args) if args is not None else ()
        super().__init__(nwalkers, ndim, a, live_dangerously, group, distributed)

    def compute_log_prob(self, coords):
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        self._check_coords(coords)
        lp = np.asarray(self.log_prob_fn(coords, *self.args), dtype=np.float64)
        if lp.shape != (coords.shape[0],):
            raise ValueError(f'log_prob_fn returned shape {lp.shape}, expected ({coords.shape[0]},)')
        if np.any(np.isnan(lp)):
            raise ValueError('Probability function returned NaN')
        return lp

    def _stretch_iteration(self):
        ndim = self.ndim
        accepted = np.zeros(self.nwalkers, dtype=bool)
        for h in draw_step(self._random, self.nwalkers, ndim, self.a):
            idx = h['active']
            Ns = len(idx)
            s = self._coords[idx]
            c = self._coords[h['partner']]  # the other half, read after its update
            q = c - (c - s) * h['zz'][:, None]
            # this rank's block of the active half (the whole half when not distributed)
            lo, hi = shard_range(Ns, self._world, self._rank)
            new_lp = self.compute_log_prob(q[lo:hi])
            old_lp = self._lp[idx][lo:hi]
            acc = h['factor'][lo:hi] + new_lp - old_lp > h['logu'][lo:hi]
            block = np.concatenate([np.where(acc[:, None], q[lo:hi], s[lo:hi]),
                                    np.where(acc, new_lp, old_lp)[:, None],
                                    acc[:, None].astype(np.float64)], axis=1)
            if self._world > 1:  # one all-gather of the just-updated rows per half-step
                block = all_gather_rows(block, Ns, self._group)
            self._coords[idx] = block[:, :ndim]
            self._lp[idx] = block[:, ndim]
            accepted[idx] = block[:, ndim + 1] > 0
        self._accepted += accepted

    def run_mcmc(self, initial_state, nsteps, progress=False, thin_by=1, **kwargs):
        """Store ``nsteps`` samples, one every ``thin_by`` iterations (emcee's ``thin_by``),
        starting from ``initial_state`` (W, ndim); pass ``None`` to continue."""
        if initial_state is None:
            if self._coords is None:
                raise ValueError('Cannot have `initial_state=None` if run_mcmc has never been called.')
        else:
            self._coords = self._check_initial(initial_state)
            self._lp = self.compute_log_prob(self._coords)
        nsteps, thin_by = int(nsteps), int(thin_by)
        if thin_by < 1:
            raise ValueError('thin_by must be >= 1')
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
            for _ in range(thin_by):
                self._stretch_iteration()
            chain[i] = self._coords
            logp[i] = self._lp
        self._append(chain, logp)
        self._moves_done += nsteps * thin_by
        return self._coords.copy(), self._lp.copy()


_PINNED = {}


def _pinned_scratch(key, nbytes):
    """Grow-only pinned host blocks (uint8 tensors) that outlive a sampler: pinning costs more than
    a short run.  Keys carry the owning thread and device (HipStretchBackend._key), so samplers
    that follow each other on one thread share blocks -- an event guards each reuse -- and samplers
    on different threads or GPUs never see each other's."""
    import torch
    blk = _PINNED.get(key)
    if blk is None or blk.numel() < nbytes:
        blk = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, pin_memory=True)
        _PINNED[key] = blk
    return blk


class HipStretchBackend:
    """Device side of ``DeviceEnsembleSampler``: buffers are torch CUDA tensors (device
    memory + stream plumbing), the arithmetic is the HIP library's."""

    def __init__(self, ctx):
        import torch
        from .utils import respect_cpu_quota
        respect_cpu_quota()
        self.torch = torch
        self.ctx = ctx
        self.device = torch.device('cuda', ctx.device)
        import threading
        self._key = f'{threading.get_ident()}:{ctx.device}:'     # the prefetch worker inherits it

    def tensor(self, array, dtype=None, slot='a'):
        """Host array -> device tensor WITHOUT waiting on the host: the data is staged in a
        process-wide pinned scratch block and copied asynchronously on the compute stream (an
        event guards the block's reuse).  A blocking copy here would also wait for the HIP
        runtime to finish retiring the launch records of whatever ran before -- 20-35 ms after
        a 2000-launch run (benchmarks/micro/upload_cost*.py) -- while an asynchronous one lets
        that happen behind the new run's kernels.  ``slot`` names the scratch block: uploads
        that follow each other within one run use different slots so that none waits for another."""
        torch = self.torch
        t = torch.as_tensor(np.ascontiguousarray(array), dtype=dtype)
        nbytes = t.numel() * t.element_size()
        if nbytes == 0:
            return torch.empty(t.shape, dtype=t.dtype, device=self.device)
        ev = _PINNED.get(self._key + 'upload_event_' + slot)
        if ev is not None:
            ev.synchronize()             # the previous upload has left this scratch block
        pinned = _pinned_scratch(self._key + 'upload_' + slot, nbytes)[:nbytes].view(t.dtype).view(t.shape)
        pinned.copy_(t)
        dev = pinned.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        _PINNED[self._key + 'upload_event_' + slot] = ev
        return dev

    def flag_nan(self, logp_t, status_t):
        """status |= 2 when any of logp is NaN -- on the device, no host round trip."""
        torch = self.torch
        status_t.bitwise_or_(torch.isnan(logp_t).any().to(torch.int32) * 2)

    def snapshot(self, dev_t, slot='snapshot', frozen=True):
        """Start a device->host copy of a small tensor AS IT IS NOW (ordered after the work queued so
        far, on a side stream, so work queued later does not delay it).  Returns (pinned host tensor,
        event of the copy's completion).  ``slot`` names the pinned block (two snapshots that are alive
        together use different slots); ``frozen=False``: nothing queued later writes dev_t, no copy of it."""
        torch = self.torch
        if not hasattr(self, '_copy_stream'):
            self._copy_stream = torch.cuda.Stream(self.device)
        src = dev_t.clone() if frozen else dev_t  # on the compute stream, before anything later touches dev_t
        ev2 = torch.cuda.Event()
        ev2.record(torch.cuda.current_stream(self.device))
        # pinned memory that outlives the sampler (pinning costs more than a short run); a previous
        # snapshot's copy has long been waited for by the run that asked for it
        nbytes = dev_t.numel() * dev_t.element_size()
        host = _pinned_scratch(self._key + slot, nbytes)[:nbytes].view(dev_t.dtype).view(dev_t.shape)
        self._copy_stream.wait_event(ev2)
        with torch.cuda.stream(self._copy_stream):
            host.copy_(src, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        src.record_stream(self._copy_stream)
        return host, done

    def gram(self, coords_t):
        """Shifted sums and second moments of the ensemble (``bisip_ensemble_gram_dev``) behind the work queued
        so far; their copy to the host starts at once.  Returns (pinned host tensor, event)."""
        from . import _hip
        torch = self.torch
        W, ndim = (int(x) for x in coords_t.shape)
        out = torch.empty((ndim + ndim * (ndim + 1) // 2,), dtype=torch.float64, device=self.device)
        work = torch.empty((_hip.ensemble_gram_workspace(W, ndim),), dtype=torch.float64, device=self.device)
        _hip.ensemble_gram_dev(coords_t.data_ptr(), W, ndim, out.data_ptr(), work.data_ptr(), self.stream())
        return self.snapshot(out, slot='gram', frozen=False)

    def side_stream(self, name, after=None):
        """Context: torch's current stream is a side stream of this backend (one per ``name``) that starts behind
        ``after`` (an event of the compute stream; default: everything queued on it so far) -- work queued inside
        runs beside what the compute stream has after that point."""
        torch = self.torch
        streams = self.__dict__.setdefault('_side_streams', {})
        if name not in streams:
            streams[name] = torch.cuda.Stream(self.device)
        if after is None:
            after = torch.cuda.Event()
            after.record(torch.cuda.current_stream(self.device))
        streams[name].wait_event(after)
        return torch.cuda.stream(streams[name])

    def shell_rows(self, chain_t, logp_t, n_samples, n_ensembles, walkers_per_ensemble, k, n_stride=0, ties=True,
                   slot='guard'):
        """Per ensemble the k stored samples of smallest |logp| of a chain slab (``bisip_chain_shell_rows_dev``,
        + n_stride walkers of the first sample), selected on the device behind the work queued so far; their
        copy to the host starts at once.  Returns (pinned host tensor (E, k + n_stride, ndim + 1), event)."""
        from . import _hip
        torch = self.torch
        ndim = int(chain_t.shape[-1])
        out = torch.empty((n_ensembles, k + n_stride, ndim + 1), dtype=torch.float64, device=self.device)
        work = torch.empty((_hip.chain_shell_rows_workspace(n_ensembles),), dtype=torch.uint8, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        chain_t.record_stream(cur)               # (read here on what may be a side stream: the allocator must not
        logp_t.record_stream(cur)                #  hand the slab to the next chunk before this selection has run)
        _hip.chain_shell_rows_dev(chain_t.data_ptr(), logp_t.data_ptr(), n_samples, n_ensembles, walkers_per_ensemble,
                                  ndim, k, n_stride, out.data_ptr(), work.data_ptr(), self.stream(), ties=ties)
        return self.snapshot(out, slot=slot, frozen=False)

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype):
        return self.torch.zeros(shape, dtype=dtype, device=self.device)

    def stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def logprob(self, coords_t, out_t):
        self.ctx.logprob_dev(coords_t.data_ptr(), coords_t.shape[0], out_t.data_ptr(), self.stream())

    def _args(self, st, k, h, n_slots, base=False):
        """Pointers of iteration k / half h of the resident chunk.  ``base`` = the chunk's
        first rows regardless of thinning (what bisip_stretch_run_dev wants)."""
        from ._hip import StretchArgs
        a = StretchArgs()
        a.walkers_per_spectrum = st.get('wp', 0)
        a.coords = st['coords'].data_ptr()
        a.logp = st['logp'].data_ptr()
        off = (k * 2 + h) * st['nh']
        if 'active' in st:                   # (a chunk whose stream is drawn in place has no arrays: run)
            a.active = st['active'].data_ptr() + 4 * off
            a.partner = st['partner'].data_ptr() + 4 * off
            a.zz = st['zz'].data_ptr() + 8 * off
            a.factor = st['factor'].data_ptr() + 8 * off
            a.logu = st['logu'].data_ptr() + 8 * off
        a.n_slots = n_slots
        W, ndim = st['coords'].shape
        thin = st.get('thin', 1)
        if base or (k + 1) % thin == 0:      # only every thin-th iteration is stored
            row = 0 if base else k // thin
            a.chain_row = st['chain'].data_ptr() + 8 * row * W * ndim
            a.logp_row = st['logp_chain'].data_ptr() + 8 * row * W
        a.naccept = st['naccept'].data_ptr()
        a.status = st['status'].data_ptr()
        return a

    def half(self, st, k, h, n_slots):
        self.ctx.stretch_half_dev(self._args(st, k, h, n_slots), self.stream())

    def eval(self, st, k, h, n_slots, lo, hi, block_t):
        a = self._args(st, k, h, n_slots)
        a.slot_lo, a.slot_hi = lo, hi
        a.block = block_t.data_ptr()
        self.ctx.stretch_eval_dev(a, self.stream())

    def apply(self, st, k, h, n_slots, gathered_t, pad, world):
        a = self._args(st, k, h, n_slots)
        a.block = gathered_t.data_ptr()
        a.pad, a.world = pad, world
        self.ctx.stretch_apply_dev(a, self.stream())

    def run(self, st, n_steps):
        """All n_steps iterations of a chunk in one C call (2 launches per step, no host
        round trip)."""
        W = st['coords'].shape[0]
        if 'inline' in st:                   # the Philox stream drawn by the half-step launches themselves
            a, seed, step0 = st['inline']
            self.ctx.stretch_run_philox_dev(self._args(st, 0, 0, (W + 1) // 2, base=True), W, n_steps, st.get('thin', 1),
                                            a, seed, step0, st['perm'].data_ptr(), self.stream())
            return
        self.ctx.stretch_run_dev(self._args(st, 0, 0, (W + 1) // 2, base=True), W, n_steps,
                                 st.get('thin', 1), self.stream())

    def run_sharded(self, st, n_steps, comm):
        """All n_steps iterations of a chunk with the walkers sharded over the ranks of the RCCL
        communicator ``comm`` (an ncclComm_t as int): eval -> ncclAllGather -> apply per half-step,
        enqueued by one C call on the compute stream."""
        W = st['coords'].shape[0]
        self.ctx.stretch_run_sharded_dev(comm, self._args(st, 0, 0, (W + 1) // 2, base=True), W, n_steps,
                                         st.get('thin', 1), self.stream())

    def run_sharded_sim(self, st, n_steps, world):
        """The same C loop with every rank of a ``world``-rank group evaluated on this device in turn and
        no collective: the slot ranges, pads and slab offsets of a multi-rank run, on one GPU (test aid)."""
        W = st['coords'].shape[0]
        self.ctx.stretch_run_sharded_sim_dev(world, self._args(st, 0, 0, (W + 1) // 2, base=True), W, n_steps,
                                             st.get('thin', 1), self.stream())

    def rccl_comm(self, group, world, rank, prefer='torch', in_group=True):
        """An ncclComm_t (int) spanning the ranks of ``group``, or None when the group does not
        run over RCCL (gloo: the caller keeps the Python eval / all_gather / apply loop).
        ``prefer='torch'`` hands the C loop the communicator torch.distributed already built for
        this group; 'own' (and the fallback) creates one from a unique id broadcast through the
        group.  Returns (comm, owned)."""
        from . import _hip
        if not in_group:      # a single process outside torch.distributed: a one-rank communicator
            return _hip.rccl_comm_create(1, 0, _hip.rccl_unique_id(), self.ctx.device), True
        import torch.distributed as dist
        if dist.get_backend(group) != 'nccl':
            return None, False
        if prefer == 'torch':
            # Borrow-or-own is decided BY THE GROUP: a rank that borrowed while another fell through
            # to the broadcast below would issue different collectives and hang.  The first
            # all-reduce also makes torch build its (lazily created) communicator on every rank.
            torch = self.torch
            flag = torch.ones(1, dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            ptr = 0
            try:
                pg = group if group is not None else dist.distributed_c10d._get_default_group()
                ptr = int(pg._get_backend(self.device)._comm_ptr())
            except Exception:      # older / newer torch without the accessor
                ptr = 0
            flag.fill_(1 if ptr else 0)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()):
                return ptr, False
        ids = [_hip.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                   group=group)
        return _hip.rccl_comm_create(world, rank, ids[0], self.ctx.device), True

    def run_persistent(self, st, wp, n_steps):
        """One launch for the whole chunk (workgroup per ensemble, state in LDS, same pre-drawn
        stream as ``run``).  Returns False when the ensemble does not fit a workgroup."""
        from ._hip import PersistArgs
        p = PersistArgs()
        p.coords = st['coords'].data_ptr()
        p.logp = st['logp'].data_ptr()
        p.n_walkers = st['coords'].shape[0]
        p.walkers_per_ensemble = wp
        p.n_steps, p.thin_by = n_steps, st.get('thin', 1)
        for name in ('active', 'partner', 'zz', 'factor', 'logu'):
            setattr(p, name, st[name].data_ptr())
        p.chain = st['chain'].data_ptr()
        p.logp_chain = st['logp_chain'].data_ptr()
        p.naccept = st['naccept'].data_ptr()
        p.status = st['status'].data_ptr()
        return self.ctx.stretch_persistent_dev(p, self.stream())

    def stream_staging(self, n, nh, slot=0):
        """Pinned host arrays (n, 2, nh) for one chunk of the NumPy-order stream.  Two slots
        alternate from chunk to chunk (one is being filled by the host while the other's upload
        is in flight); a slot is reused once its previous upload has left it."""
        torch = self.torch
        need = int(n) * 2 * int(nh)
        kinds = (('active', torch.int32), ('partner', torch.int32), ('zz', torch.float64),
                 ('factor', torch.float64), ('logu', torch.float64))
        ev = _PINNED.get(f'{self._key}stream_event_{slot}')
        if ev is not None:
            ev.synchronize()
        out = {}
        for name, dt in kinds:
            blk = _pinned_scratch(f'{self._key}stream_{slot}_{name}', need * 8)
            out[name] = blk[:need * (4 if dt == torch.int32 else 8)].view(dt).view(int(n), 2, int(nh))
        return out

    def upload_staged(self, stage, slot=0):
        """Asynchronous host->device copies of the staged stream on the compute stream."""
        out = {name: t.to(self.device, non_blocking=True) for name, t in stage.items()}
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device))
        _PINNED[f'{self._key}stream_event_{slot}'] = ev
        return out

    def host_buffer(self, shape):
        """Pinned host memory for big chains: slabs come back at PCIe rate (~55 GB/s vs ~13
        pageable); small chains stay pageable (pinning costs more than it saves)."""
        nbytes = 8 * int(np.prod(shape))
        return self.torch.empty(shape, dtype=self.torch.float64, pin_memory=nbytes >= (32 << 20))

    def copy_out(self, dst_host, src_dev):
        """Asynchronous device->host copy on a side stream, ordered after the kernels queued
        so far; the next chunk's kernels overlap with it."""
        torch = self.torch
        if not hasattr(self, '_copy_stream'):
            self._copy_stream = torch.cuda.Stream(self.device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._copy_stream.wait_event(ev)
        with torch.cuda.stream(self._copy_stream):
            dst_host.copy_(src_dev, non_blocking=True)
        src_dev.record_stream(self._copy_stream)

    def draws_in_place(self, W):
        """Can a chunk of this context's single ensemble of ``W`` walkers run without stream arrays
        (bisip_stretch_run_philox_dev)?  BISIP_NO_INLINE_DRAW (read per run) keeps the arrays: A/B runs and the test of
        their equality."""
        import os
        return os.environ.get('BISIP_NO_INLINE_DRAW') is None and self.ctx.stretch_philox_inline(W)

    def draw(self, st, W, a, seed, step0, n_steps, after=None):
        """Fill a chunk's stream arrays (Philox).  ``after`` = events the draw must wait for (its
        inputs uploaded, its output buffers no longer read): the draw then runs on a side stream,
        next to the kernels of the previous chunk -- a persistent sampler kernel keeps one wave per
        SIMD busy and leaves the rest of the chip to it -- and the compute stream waits for it.
        ``after=None``: on the compute stream, as any other kernel."""
        stream = self.stream()
        if after is not None:
            torch = self.torch
            if not hasattr(self, '_draw_stream'):
                self._draw_stream = torch.cuda.Stream(self.device)
            for ev in after:
                if ev is not None:
                    self._draw_stream.wait_event(ev)
            stream = self._draw_stream.cuda_stream
        self.ctx.stretch_draw_dev(W, a, seed, step0, n_steps, st['perm'].data_ptr(),
                                  st['active'].data_ptr(), st['partner'].data_ptr(),
                                  st['zz'].data_ptr(), st['factor'].data_ptr(),
                                  st['logu'].data_ptr(), stream)
        if after is not None:
            done = self.torch.cuda.Event()
            done.record(self._draw_stream)
            self.torch.cuda.current_stream(self.device).wait_event(done)

    def mark(self):
        """An event at the current end of the compute stream."""
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device))
        return ev

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)


class DeviceEnsembleSampler(_SamplerBase):
    """Stretch-move sampler whose ensemble, chain and half-step arithmetic live on the GPU.

    Same interface, RNG contract and chain as ``EnsembleSampler``.  ``ctx`` is a
    ``bisip_amd._hip.HipContext``; ``backend`` is injectable so the multi-rank driver
    logic can be exercised on CPU (tests/test_dist.py).
    ``chunk`` bounds the number of steps whose RNG stream / chain slab are resident at once.

    **No chunk is kept that a failing tier produced.**  PolynomialDecomposition samples with a QR-reduced
    kernel that BISIP_VARIANT_AUTO chose from an error estimate on probe rows.  The launches here hand over
    device pointers and never synchronise, so the library cannot measure them itself; the sampler does: rows
    of the initial ensemble (the samples nearest to the shell logp = 0 and a stride across it) and, chunk by
    chunk, the stored samples nearest to the shell (``bisip_chain_shell_rows_dev``: selected where the chain
    lies, 256 rows per check) are measured against the host's binary128 / long-double yardstick while the NEXT
    chunk runs (``bisip_ctx_reduced_guard_rows``).  Past 2e-11 the context moves to the next formulation
    (compensated, then per-frequency), the chunk's initial state -- saved on the device -- comes back, its
    log-probabilities are evaluated again and the chunk re-runs on the same random stream (counter-based, or
    re-drawn from the saved generator state); what was queued behind it is discarded.  ``guard_`` reports
    checks, rows, worst accepted error, escalations and re-runs.  A caller-forced variant is not moved
    (``Inversion.fit`` measures it after the run and warns).
    """

    def __init__(self, nwalkers, ndim, ctx=None, a=2.0, live_dangerously=False, group=None,
                 distributed=False, backend=None, chunk=None, rng='numpy', seed=None,
                 n_ensembles=1, force_sharded_path=False, persistent=None, chain_on_device=False,
                 sharded_loop='python'):
        if rng not in ('numpy', 'philox'):
            raise ValueError("rng must be 'numpy' or 'philox'")
        # n_ensembles > 1: independent ensembles of `nwalkers` walkers each (batch of spectra),
        # stacked as rows [e*nwalkers, (e+1)*nwalkers); the chain is (nsteps, E*nwalkers, ndim)
        self.n_ensembles = int(n_ensembles)
        self.walkers_per_ensemble = int(nwalkers)
        if self.n_ensembles > 1:
            if rng != 'philox':
                raise ValueError("n_ensembles > 1 needs rng='philox'")
            if nwalkers % 2:
                raise ValueError('n_ensembles > 1 needs an even number of walkers per ensemble')
            if distributed:
                raise ValueError('independent ensembles shard as whole replicas; run one sampler per rank')
        self.backend = backend if backend is not None else HipStretchBackend(ctx)
        self.chunk = chunk
        self.rng = rng
        # run eval -> all_gather -> apply even with one rank (benchmarks the sharded path)
        self.force_sharded_path = bool(force_sharded_path)
        # who runs the sharded half-step loop: 'python' (default) = the per-half-step loop over
        # torch.distributed (what gloo groups and injected test backends always use); 'rccl' = one
        # C call per chunk enqueues eval -> ncclAllGather -> apply for every half-step
        # (bisip_stretch_run_sharded_dev) on the communicator torch.distributed already has;
        # 'rccl-own' = the same on a communicator of this sampler's own.  Same chain every way.
        # The C loop is 5x faster with one rank but has never met a second
        # rank on hardware (this pool gives one GPU, and RCCL refuses two ranks on one device), so it
        # is opt-in until `bench.py --gpus N`'s extras have shown it equal to the Python loop there.
        # 'simulate:N' (test aid, one process): the C loop with all N ranks' blocks evaluated on this device
        self.simulate_world = 0
        if isinstance(sharded_loop, str) and sharded_loop.startswith('simulate:'):
            self.simulate_world = int(sharded_loop.split(':', 1)[1])
            if self.simulate_world < 1 or distributed:
                raise ValueError("sharded_loop='simulate:N' needs N >= 1 and a single process")
            sharded_loop = 'python'
            self.force_sharded_path = True
        if sharded_loop not in ('rccl', 'rccl-own', 'python'):
            raise ValueError("sharded_loop must be 'rccl', 'rccl-own', 'python' or 'simulate:N'")
        self.sharded_loop = sharded_loop
        self._comm, self._comm_owned, self._comm_tried = None, False, False
        # one rank and the ensemble fits a workgroup: ONE launch per chunk (workgroups of whole
        # ensembles, state in LDS); bit-identical to the launch-per-half-step path, which is the
        # automatic fallback for bigger ensembles.  A workgroup lives on ONE compute unit, so
        # for a single ensemble this wins up to a few hundred walkers: 2-3x fewer microseconds per
        # iteration at <= 128, until one CU no longer keeps up with launches that spread the
        # half-step over the chip
        # (the crossover depends on the kernel: HipContext.persistent_walkers, 256 ... 1024 walkers
        # after round 2's rework of the kernel).  A big batch of
        # ensembles fills the chip with whole-ensemble workgroups either way, and then the
        # persistent kernel saves the launch, the gathers and the per-launch ramp of every
        # half-step (512 spectra x 256 walkers: 6.5 vs 12.4 us per half-step).
        # None = that rule; True / False force it.
        if persistent is None:
            ctx = getattr(self.backend, 'ctx', None)
            limit = getattr(ctx, 'persistent_walkers', 128)
            if int(nwalkers) * self.n_ensembles >= 65536 and self.n_ensembles > 1:      # a batch that fills the chip
                persistent = limit > 0 and int(nwalkers) <= 1024 and getattr(ctx, 'persistent_in_big_batches', True)
            else:
                # one workgroup up to `limit`; beyond 1,024 walkers (where one workgroup ends) a single ensemble has
                # the multi-workgroup kernel up to HipContext.group_walkers
                persistent = int(nwalkers) <= limit or (self.n_ensembles == 1 and 1024 < int(nwalkers) <= getattr(ctx, 'group_walkers', 0))
        self.persistent = bool(persistent)
        # keep the stored samples in HBM: nothing is copied to the host until get_chain() /
        # get_log_prob() ask for it, and param_moments() summarises the chain where it lies
        self.chain_on_device = bool(chain_on_device)
        self.last_path = None
        self.last_stream = None      # run_mcmc: 'host' (NumPy order), 'arrays' (Philox, drawn per chunk) or 'in place'
        super().__init__(int(nwalkers) * self.n_ensembles, ndim, a, live_dangerously, group, distributed)
        # philox key: explicit seed, else drawn from the (seeded) private RandomState
        # (never in 'numpy' mode: that stream must stay aligned with EnsembleSampler's)
        self.seed = None
        if rng == 'philox':
            self.seed = int(seed) if seed is not None else int(self._random.randint(0, 2 ** 31 - 1))
        self._dev = None
        self._iterations_run = 0   # philox counter: iterations done so far (stored or not)

    def reset(self):
        """emcee's reset(): forget the stored samples and the acceptance counts; the ensemble itself
        stays where it is, so ``run_mcmc(None, n)`` continues from it."""
        super().reset()
        dev = getattr(self, '_dev', None)
        if dev is not None:
            dev['naccept'].zero_()            # the device counter belongs to the counts just forgotten
        self._accepted_before = np.zeros(self.nwalkers)

    def _sharded_comm(self):
        """The RCCL communicator of the C half-step loop (made on first use), or None: groups
        that do not run over RCCL and injected test backends keep the Python loop."""
        if self._comm is None and not self._comm_tried:
            self._comm_tried = True
            if hasattr(self.backend, 'rccl_comm') and self.sharded_loop != 'python':
                self._comm, self._comm_owned = self.backend.rccl_comm(
                    self._group, self._world, self._rank,
                    prefer='own' if self.sharded_loop == 'rccl-own' else 'torch', in_group=self._in_group)
        return self._comm

    def close(self):
        """Release the communicator this sampler created (no-op otherwise)."""
        if getattr(self, '_comm', None) is not None and self._comm_owned:
            from . import _hip
            self.backend.synchronize()
            _hip.rccl_comm_destroy(self._comm)
        self._comm, self._comm_owned = None, False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _upload_state(self, coords, lp=None):
        import torch
        be = self.backend
        W = self.nwalkers
        # the device counter restarts with every new initial state; acceptances of earlier runs
        # stay in the numerator, as their iterations stay in the denominator (_moves_done)
        self._accepted_before = np.asarray(self._accepted, dtype=np.float64).copy()
        dev = dict(coords=be.tensor(coords, torch.float64),
                   naccept=be.zeros((W,), torch.int32), status=be.zeros((1,), torch.int32))
        if lp is None:
            # initial log-probabilities: computed, and checked for NaN (status bit 1), on the
            # device with no host synchronisation; run_mcmc reads the status word when it ends
            dev['logp'] = be.empty((W,), torch.float64)
            be.logprob(dev['coords'], dev['logp'])
            be.flag_nan(dev['logp'], dev['status'])
            # ... and its copy to the host starts now, beside whatever is queued next: run_mcmc looks at
            # it once the first chunk is enqueued, so a NaN in the initial state raises before the run,
            # as in emcee, at the price of one wait that the chunk's kernels overlap
            if hasattr(be, 'snapshot'):
                dev['status0'] = be.snapshot(dev['status'])
            dev['fresh'] = True           # (run_mcmc's guard measures rows of this ensemble beside the first chunk)
        else:
            dev['logp'] = be.tensor(lp, torch.float64, slot='b')
        self._dev = dev

    _GRAM_ON_DEVICE_FROM = 16384      # walkers from which the independence test of a new ensemble runs on the device

    def _independent_on_device(self, pending):
        """Decide emcee's independence test from the moments the device formed of the uploaded ensemble; what they
        do not settle (near-degenerate, degenerate, non-finite) goes through the host's singular values."""
        (host, ev), p0 = pending
        ev.synchronize()
        g = gram_from_shifted_sums(np.array(host.numpy(), copy=True), self.nwalkers, self.ndim)
        return gram_decides(g, self.nwalkers) or walkers_independent(p0)

    def _guard_plan(self):
        """How this run measures the kernel it samples with, or None when there is nothing to measure: only a
        PolynomialDecomposition context that chose a QR-reduced kernel from its estimate ('auto') is guarded
        (HipContext.guards_itself).  ``k`` rows per ensemble and check: 256 for one ensemble, fewer per spectrum
        of a batch (4096 rows in all, at least 4 each).  Several ranks: ties are left out of the selection, so
        that every rank measures the same rows and takes the same decision without a collective."""
        be = self.backend
        ctx = getattr(be, 'ctx', None)
        if ctx is None or not hasattr(be, 'shell_rows') or not getattr(ctx, 'guards_itself', False):
            return None
        return dict(k=int(max(4, min(256, 4096 // self.n_ensembles))), ties=self._world == 1)

    def _guard_passes(self, rec):
        """Measure the rows selected from chunk rec['k'] (and, with chunk 0, from the initial ensemble) against
        the host's yardstick (bisip_ctx_reduced_guard_rows).  False: the context has just left the tier that
        produced them -- the chunk is to be run again."""
        import time
        t0 = time.perf_counter()
        ctx = self.backend.ctx
        E, ndim = self.n_ensembles, self.ndim
        parts = []
        for host, ev in rec['rows']:
            ev.synchronize()
            parts.append(host.numpy().reshape(E, -1, ndim + 1))
        rows = (np.concatenate(parts, axis=1) if len(parts) > 1 else parts[0]).reshape(-1, ndim + 1)
        theta, lp = np.ascontiguousarray(rows[:, :ndim]), np.ascontiguousarray(rows[:, ndim])
        worst, escalated = ctx.reduced_guard_rows(theta, lp)
        g = self.guard_
        g['checks'] += 1
        g['rows'] += int(np.isfinite(lp).sum())
        if escalated:
            g['escalations'] += 1
            g['rejected'] = max(g.get('rejected', 0.0), worst) if worst == worst else float('nan')
            g['kernel'] = ctx.kernel_name
        elif not worst <= g['worst']:
            g['worst'] = worst
        self.timing['guard_s'] = self.timing.get('guard_s', 0.0) + time.perf_counter() - t0
        return not escalated

    def _chunk_steps(self, nsteps, stream_in_place=False):
        if self.chunk:
            return max(1, int(self.chunk))
        # bytes resident per iteration: chain row + log-prob + the five stream arrays (none when the half-step
        # launches draw the stream themselves).  2 GiB
        # per chunk is <1 % of the 288 GB of HBM and keeps the host work per chunk negligible
        # (the Philox stream is double-buffered: the next chunk's is drawn while this one runs)
        per_step = self.nwalkers * (8 * self.ndim + 8 + (0 if stream_in_place else 2 if self.rng == 'philox' else 1) * (3 * 8 + 2 * 4))
        n = max(1, min(nsteps, (2 << 30) // per_step))
        if self.rng == 'numpy':
            # the NumPy-order stream is drawn on the host (MT19937 is sequential: ~20-60 ns per
            # walker-step): chunks of ~250k walker-steps keep that to a few milliseconds, so the
            # stream of chunk k+1 is drawn while the kernels of chunk k run instead of before them
            n = min(n, max(16, 250_000 // self.nwalkers))
        return n

    # -- run_mcmc and its stages ---------------------------------------------------------
    def _start_from(self, initial_state):
        """Validate a new initial state (emcee's checks) and make it the device-resident
        ensemble; ``None`` continues from the current one."""
        import time
        t0 = time.perf_counter()
        W, ndim = self.nwalkers, self.ndim
        if initial_state is None:
            if self._dev is None:
                raise ValueError('Cannot have `initial_state=None` if run_mcmc has never been called.')
            return {}
        on_device = False
        if self.n_ensembles == 1:
            # emcee's independence test of a big ensemble runs where the ensemble is going anyway: the device forms
            # the second moments (bisip_ensemble_gram_dev), the ndim x ndim decision is looked at once the first
            # chunk is queued (run_mcmc) -- on the host it costs more than hundreds of half-steps (26 ms at a
            # million walkers).  ndim <= 8; anything the moments do not settle clearly still takes the singular values.
            on_device = (not self.live_dangerously and W >= self._GRAM_ON_DEVICE_FROM and ndim <= 8
                         and hasattr(self.backend, 'gram'))
            p0 = self._check_initial(initial_state, copy=False, independence=not on_device)
        else:
            p0 = np.asarray(initial_state, dtype=np.float64).reshape(W, ndim)   # only read
            Wp = self.walkers_per_ensemble
            if not self.live_dangerously:
                if Wp < 2 * ndim:
                    raise RuntimeError('It is unadvisable to use a red-blue move with fewer '
                                       'walkers than twice the number of dimensions.')
                for e in np.unique(np.linspace(0, self.n_ensembles - 1, 32).astype(int)):
                    if not walkers_independent(p0[e * Wp:(e + 1) * Wp]):
                        raise ValueError(f'Initial state of ensemble {e} has a large condition number.')
        if self.live_dangerously or self.n_ensembles > 1:
            self._check_coords(p0)        # (a single ensemble that passed walkers_independent is finite throughout)
        t1 = time.perf_counter()
        self._upload_state(p0)
        if on_device:
            self._dev['gram0'] = (self.backend.gram(self._dev['coords']), p0)
        return dict(check_s=t1 - t0, upload_s=time.perf_counter() - t1)

    def _numpy_stream_host(self, n, nh, slot):
        """Chunk of the NumPy-order stream, host part: generated in C from the RandomState's
        MT19937 state (bit-identical to calling draw_step n times, ~30x cheaper) straight into
        pinned staging memory; the logs are NumPy's so they match the host sampler's.  Runs on a
        worker thread for every chunk but the first (the C generator and NumPy's log release the
        GIL), so chunk k+1 is drawn while the main thread enqueues chunk k's kernels."""
        from ._hip import numpy_stretch_stream
        state0 = self._random.get_state()         # a chunk the guard sends back is drawn again from here
        stage = self.backend.stream_staging(n, nh, slot)
        _, _, zz, u = numpy_stretch_stream(
            self._random, self.nwalkers, self.a, n,
            out=tuple(stage[name].numpy() for name in ('active', 'partner', 'zz', 'logu')))
        with np.errstate(divide='ignore'):
            factor = stage['factor'].numpy()
            np.log(zz, out=factor)
            factor *= self.ndim - 1.0
            np.log(u, out=u)             # 'logu' staging held u
        return stage, state0

    def _advance(self, st, n, nh, it0):
        """Enqueue the n iterations of a chunk: persistent kernel, fused launches, or (several
        ranks) eval -> all-gather -> apply per half-step."""
        import torch
        be = self.backend
        W, ndim = self.nwalkers, self.ndim
        single = self._world == 1 and not self.force_sharded_path
        if single and self.persistent and 'inline' not in st and be.run_persistent(st, self.walkers_per_ensemble, n):
            # one workgroup per ensemble, or -- a single ensemble beyond that -- several with a barrier of their own
            self.last_path = 'persistent' if self.walkers_per_ensemble * (ndim + 1) * 8 <= 65536 and (self.walkers_per_ensemble + 1) // 2 <= 512 \
                else 'persistent-multi-workgroup'
        elif single:
            be.run(st, n)
            self.last_path = 'launch-per-half-step'
        elif self.simulate_world:
            be.run_sharded_sim(st, n, self.simulate_world)
            self.last_path = f'sharded-simulated-{self.simulate_world}'
        elif self._sharded_comm() is not None:
            be.run_sharded(st, n, self._comm)
            self.last_path = 'sharded-rccl'
        else:
            import torch.distributed as dist
            self.last_path = 'sharded'
            for k in range(n):
                for h in (0, 1):
                    m = nh if h == 0 else W // 2
                    lo, hi = shard_range(m, self._world, self._rank)
                    pad = -(-m // self._world)
                    block = be.zeros((pad, ndim + 2), torch.float64)
                    be.eval(st, k, h, m, lo, hi, block)
                    gathered = be.empty((self._world * pad, ndim + 2), torch.float64)
                    if block.is_cuda and dist.get_backend(self._group) == 'gloo':
                        # rehearsal of the multi-rank path on a box whose ranks share one GPU
                        # (RCCL refuses that): the exchange goes through host memory
                        host = torch.empty(gathered.shape, dtype=torch.float64)
                        dist.all_gather_into_tensor(host, block.cpu(), group=self._group)
                        gathered.copy_(host)
                    else:
                        dist.all_gather_into_tensor(gathered, block, group=self._group)
                    be.apply(st, k, h, m, gathered, pad, self._world)

    def run_mcmc(self, initial_state, nsteps, progress=False, thin_by=1, **kwargs):
        """Store ``nsteps`` samples, one every ``thin_by`` iterations; the ensemble, the
        random stream and the chain slab of each chunk stay on the device, chain slabs
        return to pinned host memory asynchronously while the next chunk runs (or stay in
        HBM with ``chain_on_device``)."""
        import time
        import torch
        t_start = time.perf_counter()
        be = self.backend
        W, ndim = self.nwalkers, self.ndim
        setup_detail = self._start_from(initial_state)
        nsteps, thin_by = int(nsteps), int(thin_by)
        if thin_by < 1:
            raise ValueError('thin_by must be >= 1')
        nh = (W + 1) // 2                       # slots per half (the first half gets the odd one)
        chain_host = logp_host = None
        dev_chain = dev_logp = None              # chain_on_device: the whole run's samples, one block
        stream_bufs = [None, None]               # philox stream arrays: two sets, one being drawn while the other is read
        free_ev = [None, None]                   # ... and the event after which each set may be overwritten
        perm_ev = None
        perm_all = None
        if self.rng == 'philox' and nsteps > 0:
            # the per-step splits of the WHOLE run (12 B per iteration) go up in one copy: a
            # host->device copy per chunk would wait for the previous chunk's kernels
            perm_all = be.tensor(affine_splits(self.seed, self.walkers_per_ensemble,
                                               self._iterations_run, nsteps * thin_by), slot='c')
            perm_ev = be.mark() if hasattr(be, 'mark') else None
        # where a run spends its time
        self.timing = dict(setup_s=time.perf_counter() - t_start, stream_s=0.0, enqueue_s=0.0, alloc_s=0.0,
                           drain_s=0.0, finish_s=0.0, **setup_detail)
        it_start = self._iterations_run
        # A single ensemble big enough for the packed-state half-step has its Philox stream drawn by the half-step
        # launches themselves: no stream arrays (34 MB per iteration at a million walkers), no draw kernel.
        in_place = (self.rng == 'philox' and self.n_ensembles == 1 and self._world == 1 and not self.force_sharded_path
                    and not self.simulate_world and hasattr(be, 'draws_in_place') and be.draws_in_place(W))
        self.last_stream = 'in place' if in_place else ('arrays' if self.rng == 'philox' else 'host')
        # stored samples per chunk, known up front so that the next chunk's stream can be drawn ahead
        sizes, left = [], nsteps
        while left > 0:
            sizes.append(min(max(1, self._chunk_steps(nsteps * thin_by, in_place) // thin_by), left))
            left -= sizes[-1]
        if not in_place and self.rng == 'philox' and sizes:
            # a chunk holds whole thinning intervals: with a very long interval its stream arrays (two sets, 32 bytes per
            # slot and half-step) grow with it -- say so before the allocator does
            need = 2 * sizes[0] * thin_by * 2 * nh * 32
            if need > (32 << 30):
                import warnings
                warnings.warn(f'thin_by={thin_by} with {W} walkers needs {need / 2**30:.0f} GiB of random-stream arrays per chunk '
                              '(a chunk holds whole thinning intervals); consider a smaller thin_by', ResourceWarning, stacklevel=2)
        first_sample = [0]                       # stored samples before chunk k
        for ns in sizes:
            first_sample.append(first_sample[-1] + ns)
        ahead = None                             # (worker thread, its result holder) for the next chunk
        rng_state0 = self._random.get_state() if self.rng == 'numpy' else None
        nan_initial = False
        # The guard of the QR-reduced kernels (class docstring): the tier a context on 'auto' runs was chosen from
        # an estimate; rows of this run -- the initial ensemble, then chunk by chunk the stored samples nearest to
        # the shell logp = 0 -- are measured against the host's yardstick while the next chunk runs, and a chunk
        # whose tier fails is run again, from its saved initial state, by the next formulation.
        guard = self._guard_plan()
        self.guard_ = dict(checks=0, rows=0, worst=0.0, escalations=0, reruns=0, kernel=None)
        saved = {}                               # chunk -> (coords, naccept, status) as the chunk found them
        final = {}                               # the last state's copies to the host, started behind the last chunk
        E, Wp = self.n_ensembles, self.walkers_per_ensemble

        def enqueue(k):
            """Everything chunk k needs, queued behind chunk k-1: stream, chain slab, kernels, copies out."""
            nonlocal ahead, dev_chain, dev_logp, chain_host, logp_host, perm_ev, nan_initial
            t_a = time.perf_counter()
            ns = sizes[k]
            n = ns * thin_by                     # iterations in this chunk
            done, it0 = first_sample[k], it_start + first_sample[k] * thin_by
            rec = dict(k=k, ns=ns)
            st = dict(self._dev)
            st['nh'] = nh
            st['thin'] = thin_by
            if self.n_ensembles > 1:
                st['wp'] = self.walkers_per_ensemble
            if self.rng == 'numpy':
                if ahead is None:
                    stage, rec['rng_state'] = self._numpy_stream_host(n, nh, k % 2)
                else:
                    ahead[0].join()
                    if 'error' in ahead[1]:
                        raise ahead[1]['error']
                    stage, rec['rng_state'] = ahead[1]['stage']
                st.update(be.upload_staged(stage, k % 2))
                ahead = None
                if k + 1 < len(sizes):           # draw the next chunk while this one is enqueued
                    import threading
                    box = {}

                    def draw(n_next=sizes[k + 1] * thin_by, slot=(k + 1) % 2, box=box):
                        try:
                            box['stage'] = self._numpy_stream_host(n_next, nh, slot)
                        except BaseException as exc:     # re-raised on the main thread
                            box['error'] = exc
                    worker = threading.Thread(target=draw, daemon=True)
                    worker.start()
                    ahead = (worker, box)
            else:
                # only the per-step split is drawn on the host; the stream is generated on
                # the device from (seed, step, half, slot) counters
                off = it0 - self._iterations_run
                st['perm'] = perm_all[off:off + n]
            if self.chain_on_device:
                if dev_chain is None:
                    dev_chain = be.empty((nsteps, W, ndim), torch.float64)
                    dev_logp = be.empty((nsteps, W), torch.float64)
                st['chain'], st['logp_chain'] = dev_chain[done:done + ns], dev_logp[done:done + ns]
            else:
                st['chain'] = be.empty((ns, W, ndim), torch.float64)
                st['logp_chain'] = be.empty((ns, W), torch.float64)
            t_b = time.perf_counter()
            if in_place:
                st['inline'] = (self.a, self.seed, it0)
            elif self.rng == 'philox':
                b = k % 2
                if stream_bufs[0] is None:       # chunks never grow: later ones reuse these two sets
                    for i, rows in enumerate([sizes[0] * thin_by] + ([sizes[1] * thin_by] if len(sizes) > 1 else [])):
                        stream_bufs[i] = {name: be.empty((rows, 2, nh), dt) for name, dt in (
                            ('active', torch.int32), ('partner', torch.int32), ('zz', torch.float64),
                            ('factor', torch.float64), ('logu', torch.float64))}
                    if perm_ev is not None:      # the side stream starts after the split upload AND these
                        perm_ev = be.mark()      # allocations (the allocator may hand out memory still in use upstream)
                for name, buf in stream_bufs[b].items():
                    st[name] = buf[:n]
                if len(sizes) > 1 and perm_ev is not None:
                    # beside the previous chunk's kernels; this set was last read two chunks ago
                    be.draw(st, self.walkers_per_ensemble, self.a, self.seed, it0, n, after=(perm_ev, free_ev[b]))
                else:
                    be.draw(st, self.walkers_per_ensemble, self.a, self.seed, it0, n)
            fresh = bool(self._dev.pop('fresh', False)) and k == 0
            saved_ev = None
            if guard is not None:
                saved[k] = tuple(self._dev[name].clone() for name in ('coords', 'naccept', 'status', 'logp'))
                saved_ev = be.mark()             # the copies exist from here on: what reads them need not wait for the chunk
            # A big chunk's rows are selected in two parts: the first 7/8 of its samples on a side stream WHILE the
            # last eighth still runs, so that what follows the chunk's last kernel -- and, for the last chunk, ends
            # the run -- is the selection over an eighth of the samples (~25 us instead of ~100 at 6.5M samples).
            tail = ns // 8 if (guard is not None and not self.persistent and ns >= 16 and ns * W >= (1 << 20)) else 0
            if tail:
                head = ns - tail

                def part(lo, hi):
                    sub = dict(st)
                    if in_place:
                        sub['perm'] = st['perm'][lo * thin_by:hi * thin_by]
                        sub['inline'] = (self.a, self.seed, it0 + lo * thin_by)
                    for name in ('active', 'partner', 'zz', 'factor', 'logu') if not in_place else ():
                        sub[name] = st[name][lo * thin_by:hi * thin_by]
                    sub['chain'], sub['logp_chain'] = st['chain'][lo:hi], st['logp_chain'][lo:hi]
                    return sub
                first, last = part(0, head), part(head, ns)
                self._advance(first, head * thin_by, nh, it0)
                with be.side_stream('guard'):
                    rec['rows'] = [be.shell_rows(first['chain'], first['logp_chain'], head, E, Wp, guard['k'],
                                                 ties=guard['ties'], slot=f'guard{k % 2}')]
                self._advance(last, tail * thin_by, nh, it0 + head * thin_by)
            else:
                self._advance(st, n, nh, it0)
            if self.rng == 'philox' and not in_place and len(sizes) > 1 and perm_ev is not None:
                free_ev[k % 2] = be.mark()
            if guard is not None:
                if tail:
                    rec['rows'].append(be.shell_rows(last['chain'], last['logp_chain'], tail, E, Wp, max(4, guard['k'] // 4),
                                                     ties=guard['ties'], slot=f'guard{k % 2}b'))
                else:
                    rec['rows'] = [be.shell_rows(st['chain'], st['logp_chain'], ns, E, Wp, guard['k'], ties=guard['ties'],
                                                 slot=f'guard{k % 2}')]
                if fresh:
                    # the guard's first rows: the initial ensemble's samples nearest to the shell and a stride across
                    # it, selected from the copy saved above on a side stream: QUEUED behind the chunk's launches (the
                    # device had been idle until the host had queued them) but ORDERED only behind the copy, so that it
                    # runs, and is measured, beside the chunk
                    near = min(Wp, max(1, 3 * guard['k'] // 4))
                    with be.side_stream('guard_init', after=saved_ev):
                        rec['init'] = be.shell_rows(saved[0][0], saved[0][3], 1, E, Wp, near,
                                                    n_stride=min(Wp, guard['k'] - near), ties=guard['ties'], slot='guard_init')
            if k == len(sizes) - 1 and hasattr(be, 'snapshot'):
                # what the run ends with starts its way to the host now, behind the last chunk and beside the
                # measurement of its rows (four blocking copies cost 0.2-0.3 ms at the end of a 5 ms run)
                final.clear()
                final.update({name: be.snapshot(self._dev[name], slot='final_' + name, frozen=False)
                              for name in ('status', 'naccept', 'coords', 'logp')})
            gram0 = self._dev.pop('gram0', None) if k == 0 else None
            if gram0 is not None and not self._independent_on_device(gram0):
                raise ValueError(self._DEPENDENT)
            early = self._dev.pop('status0', None) if k == 0 else None
            if early is not None:
                early[1].synchronize()               # the initial log-probabilities only: not this chunk
                if int(early[0][0]) & 2:
                    nan_initial = True
                    raise ValueError('Probability function returned NaN')
            t_alloc = 0.0
            if not self.chain_on_device:
                if chain_host is None:
                    # pinning a big host chain takes tens of ms: do it while the first chunk runs
                    t_h = time.perf_counter()
                    chain_host = be.host_buffer((nsteps, W, ndim))
                    logp_host = be.host_buffer((nsteps, W))
                    t_alloc = time.perf_counter() - t_h
                    self.timing['alloc_s'] = t_alloc
                be.copy_out(chain_host[done:done + ns], st['chain'])
                be.copy_out(logp_host[done:done + ns], st['logp_chain'])
            self.timing['stream_s'] += t_b - t_a
            self.timing['enqueue_s'] += time.perf_counter() - t_b - t_alloc
            return rec

        def send_back(rec):
            """Chunk rec['k'] ran on a tier that its own rows have just failed: the context has moved on; put the
            ensemble back where the chunk found it, evaluate it with the new kernel, and draw the chunk again."""
            nonlocal ahead, guard
            be.synchronize()                     # whatever was queued behind the chunk is discarded work
            if ahead is not None:                # the next chunk's stream was drawn past the point we return to
                ahead[0].join()
                ahead = None
            if 'rng_state' in rec:
                self._random.set_state(rec['rng_state'])
            coords0, naccept0, status0, _ = saved[rec['k']]
            self._dev['coords'].copy_(coords0)
            self._dev['naccept'].copy_(naccept0)
            self._dev['status'].copy_(status0)
            be.logprob(self._dev['coords'], self._dev['logp'])
            be.flag_nan(self._dev['logp'], self._dev['status'])
            saved.clear()
            free_ev[0] = free_ev[1] = None
            self.guard_['reruns'] += 1
            guard = self._guard_plan()           # the per-frequency form needs no guard
            return rec['k']

        try:
            k, pending = 0, None                 # pending: the chunk whose selected rows have not been measured yet
            while True:
                rec = enqueue(k) if k < len(sizes) else None
                if rec is not None and 'init' in rec and not self._guard_passes(dict(k=0, rows=[rec.pop('init')])):
                    k, pending = send_back(rec), None            # the initial ensemble itself fails the tier
                    continue
                if pending is not None:
                    if not self._guard_passes(pending):
                        k, pending = send_back(pending), None
                        continue
                    saved.pop(pending['k'], None)
                pending = rec if (rec is not None and 'rows' in rec) else None
                if rec is None:
                    break
                k += 1
            done, it0 = nsteps, it_start + nsteps * thin_by
        except BaseException:
            self._dev = None             # the device ensemble is part-way through a chunk: nothing to continue from
            try:                         # nothing of this run may still be writing when its buffers go back to the allocator
                be.synchronize()         # (the stream draw runs on a side stream)
            except Exception:
                pass
            raise
        finally:
            if ahead is not None:        # an error above: let the worker finish with the RandomState first
                ahead[0].join()
            if nan_initial and rng_state0 is not None:
                self._random.set_state(rng_state0)     # emcee raises before drawing anything
        if chain_host is None and dev_chain is None:  # nsteps == 0
            chain_host, logp_host = be.host_buffer((0, W, ndim)), be.host_buffer((0, W))
        t_d = time.perf_counter()
        be.synchronize()
        t_e = time.perf_counter()
        self.timing['drain_s'] = t_e - t_d
        self._iterations_run = it0
        ends = {}
        for name in ('status', 'naccept', 'coords', 'logp'):
            if name in final:
                final[name][1].synchronize()
                ends[name] = np.array(final[name][0].numpy(), copy=True)     # (the pinned block is scratch)
            else:
                ends[name] = self._dev[name].cpu().numpy()
        if int(ends['status'][0]) & 4:                 # bit 2: the multi-workgroup kernel's workgroups never met
            self._dev = None
            raise RuntimeError('the multi-workgroup persistent sampler gave up at a barrier: its workgroups were not all '
                               'resident (a GPU shared with other work?); nothing of this run is kept -- run with persistent=False')
        if int(ends['status'][0]) & 3:                 # bit 0: a proposal, bit 1: the initial state
            self._dev = None                           # nothing of this run is kept
            raise ValueError('Probability function returned NaN')
        if dev_chain is not None:
            self._append(_DeviceSlabs([dev_chain]), _DeviceSlabs([dev_logp]))
        else:
            self._append(chain_host.numpy(), logp_host.numpy())
        self._moves_done += nsteps * thin_by
        self._accepted = self._accepted_before + ends['naccept']
        self._coords = ends['coords']
        self._lp = ends['logp']
        self.timing['finish_s'] = time.perf_counter() - t_e
        if self._coords.nbytes >= (8 << 20):      # a big ensemble: the arrays returned are the sampler's own host copy
            return self._coords, self._lp         # of its last state (another 56 MB copy at a million walkers: 5 ms)
        return self._coords.copy(), self._lp.copy()

    # -- summaries of a device-resident chain -------------------------------------------
    def device_chain(self):
        """All stored samples as ONE torch tensor (iteration, W, ndim) on the device
        (``chain_on_device=True`` runs only)."""
        parts = self._chain_parts
        if not parts or not all(isinstance(p, _DeviceSlabs) for p in parts):
            raise AttributeError('the chain is not resident on the device '
                                 '(run with chain_on_device=True)')
        if len(parts) > 1:
            parts[:] = [_DeviceSlabs(t for p in parts for t in p.tensors)]
        return parts[0].tensor()

    def param_moments(self, discard=0, thin=1):
        """Mean and standard deviation of every parameter over
        ``chain[discard + thin - 1::thin]`` flattened over the walkers of each ensemble --
        ``np.mean`` / ``np.std`` of ``get_chain(discard, thin, flat=True)`` (reference:
        src/bisip/utils.py:55-85) -- computed on the device, only the two
        ``(n_ensembles, ndim)`` results come back.  Returns ``(mean, std)``."""
        import torch
        from . import _hip
        t = self.device_chain()
        n_total, W, ndim = (int(x) for x in t.shape)
        discard, thin = int(discard), int(thin)
        first = discard + thin - 1
        n = len(range(first, n_total, thin))
        if thin < 1 or discard < 0 or n < 1:
            raise ValueError(f'no samples left with discard={discard}, thin={thin} of {n_total} stored')
        be = self.backend
        E, Wp = self.n_ensembles, self.walkers_per_ensemble
        mean = be.empty((E, ndim), torch.float64)
        std = be.empty((E, ndim), torch.float64)
        work = be.empty((max(1, _hip.chain_moments_workspace(n, E, ndim)),), torch.float64)
        _hip.chain_moments_dev(t.data_ptr() + 8 * first * W * ndim, n, thin * W * ndim, E, Wp, ndim,
                               mean.data_ptr(), std.data_ptr(), work.data_ptr(), be.stream())
        be.synchronize()
        return mean.cpu().numpy(), std.cpu().numpy()

    def model_percentiles(self, p=(2.5, 50, 97.5), discard=0, thin=1):
        """``np.percentile(forward(get_chain(discard, thin, flat=True)), p, axis=0)`` -- the
        reference's get_model_percentile (src/bisip/utils.py:17-35) -- without the chain leaving
        the device: batched forward over the stored samples, written column by column, then the
        selection of the order statistics from each column.  One ensemble only (NotImplementedError
        otherwise); returns ``(len(p), 2, N)``."""
        import torch
        from . import _hip
        if self.n_ensembles != 1:
            raise NotImplementedError('model percentiles of a batch of spectra: one spectrum at a time')
        t = self.device_chain()
        n_total = int(t.shape[0])
        discard, thin = int(discard), int(thin)
        if thin < 1 or discard < 0 or len(range(discard + thin - 1, n_total, thin)) < 1:
            raise ValueError(f'no samples left with discard={discard}, thin={thin} of {n_total} stored')
        rows = t[discard + thin - 1::thin].reshape(-1, self.ndim).contiguous()
        be, ctx = self.backend, self.backend.ctx
        p = np.atleast_1d(np.asarray(p, dtype=np.float64))
        n, cols = int(rows.shape[0]), 2 * ctx.N
        Zc = be.empty((cols, n), torch.float64)                  # the responses, one column per (part, frequency)
        ctx.forward_columns_dev(0, 1, rows.data_ptr(), n, Zc.data_ptr(), be.stream())
        out = be.empty((p.size, cols), torch.float64)
        _hip.columns_percentiles_dev(Zc.data_ptr(), cols, n, p, out.data_ptr(), be.stream())
        be.synchronize()
        return out.cpu().numpy().reshape(p.size, 2, ctx.N)

    def param_percentiles(self, p=(2.5, 50, 97.5), discard=0, thin=1):
        """``np.percentile(get_chain(discard, thin, flat=True), p, axis=0)`` per ensemble
        (reference: src/bisip/utils.py:37-53), sorted and interpolated on the device; returns
        ``(len(p), n_ensembles, ndim)``."""
        import torch
        from . import _hip
        t = self.device_chain()
        n_total, W, ndim = (int(x) for x in t.shape)
        discard, thin = int(discard), int(thin)
        first = discard + thin - 1
        n = len(range(first, n_total, thin))
        if thin < 1 or discard < 0 or n < 1:
            raise ValueError(f'no samples left with discard={discard}, thin={thin} of {n_total} stored')
        p = np.atleast_1d(np.asarray(p, dtype=np.float64))
        be = self.backend
        E, Wp = self.n_ensembles, self.walkers_per_ensemble
        nbytes = _hip.chain_percentiles_workspace(n, E, Wp, ndim, p.size)
        if nbytes <= 0:
            raise ValueError('chain too large for one device sort (more than 2^31 values); thin it or use get_chain()')
        work = be.empty((nbytes,), torch.uint8)
        out = be.empty((p.size, E, ndim), torch.float64)
        _hip.chain_percentiles_dev(t.data_ptr() + 8 * first * W * ndim, n, thin * W * ndim, E, Wp, ndim, p,
                                   out.data_ptr(), work.data_ptr(), nbytes, be.stream())
        be.synchronize()
        return out.cpu().numpy()
