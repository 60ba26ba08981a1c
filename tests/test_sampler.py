"""The native ensemble sampler (host logic), driven by the CPU oracle as log_prob_fn.

Sampler parity with emcee is unpinned; these tests pin this
implementation: RNG contract / reproducibility, chain bookkeeping, and that it
samples the right distribution.
"""

import numpy as np
import pytest

import oracle
from bisip_amd.sampler import EnsembleSampler, walkers_independent
from conftest import golden_cases


def gaussian_logp(theta, mu, icov):
    d = theta - mu
    return -0.5 * np.einsum('ni,ij,nj->n', d, icov, d)


def test_samples_a_correlated_gaussian():
    np.random.seed(123)
    mu = np.array([1.0, -2.0, 0.5])
    cov = np.array([[1.0, 0.6, 0.0], [0.6, 2.0, -0.3], [0.0, -0.3, 0.5]])
    s = EnsembleSampler(48, 3, gaussian_logp, args=(mu, np.linalg.inv(cov)))
    p0 = mu + 0.1 * np.random.randn(48, 3)
    s.run_mcmc(p0, 1500)
    flat = s.get_chain(discard=400, flat=True)
    assert flat.shape == (1100 * 48, 3)
    assert np.allclose(flat.mean(0), mu, atol=0.1)
    assert np.allclose(np.cov(flat.T), cov, atol=0.2)
    assert 0.2 < s.acceptance_fraction.mean() < 0.9


def test_rng_contract_and_bookkeeping():
    def run(seed):
        np.random.seed(seed)
        s = EnsembleSampler(16, 2, gaussian_logp, args=(np.zeros(2), np.eye(2)))
        p0 = np.random.uniform(-1, 1, (16, 2))
        s.run_mcmc(p0, 40)
        return s
    a, b, c = run(7), run(7), run(8)
    assert np.array_equal(a.get_chain(), b.get_chain())       # same global seed -> same chain
    assert not np.array_equal(a.get_chain(), c.get_chain())
    ch = a.get_chain()
    assert ch.shape == (40, 16, 2) and a.get_log_prob().shape == (40, 16)
    # emcee slicing: chain[discard + thin - 1 : iteration : thin]
    assert np.array_equal(a.get_chain(discard=10, thin=3), ch[12:40:3])
    assert a.get_chain(discard=10, thin=3, flat=True).shape == (len(ch[12:40:3]) * 16, 2)
    # the stored log-prob is the log-prob of the stored position
    assert np.allclose(a.get_log_prob()[-1], gaussian_logp(ch[-1], np.zeros(2), np.eye(2)))
    # continuing from the last state appends
    a.run_mcmc(None, 5)
    assert a.get_chain().shape[0] == 45 and a.iteration == 45


def test_stream_follows_emcees_consumption_order():
    """emcee's per-iteration RNG use, restated here literally:
    ``random.choice(moves, p=weights)`` -- one uniform double even for a single move --, a
    shuffle of the alternating labels, then rand / randint / rand per half.  draw_step must
    leave the RandomState exactly where that sequence leaves it, every iteration."""
    from bisip_amd.sampler import draw_step
    W, ndim, a = 21, 4, 2.0
    r1, r2 = np.random.RandomState(99), np.random.RandomState(99)
    moves, weights = [object()], [1.0]
    for _ in range(25):
        halves = draw_step(r1, W, ndim, a)
        r2.choice(moves, p=weights)
        inds = np.arange(W) % 2
        r2.shuffle(inds)
        for split, h in zip((0, 1), halves):
            S1 = inds == split
            Ns, Nc = int(S1.sum()), int((~S1).sum())
            zz = ((a - 1.0) * r2.rand(Ns) + 1) ** 2.0 / a
            rint = r2.randint(Nc, size=(Ns,))
            lnu = np.log(r2.rand(Ns))
            assert np.array_equal(h['active'], np.arange(W)[S1])
            assert np.array_equal(h['partner'], np.arange(W)[~S1][rint])
            assert np.array_equal(h['zz'], zz) and np.array_equal(h['logu'], lnu)
        assert r1.rand() == r2.rand()


def test_input_validation():
    s = EnsembleSampler(8, 3, gaussian_logp, args=(np.zeros(3), np.eye(3)))
    with pytest.raises(ValueError):
        s.run_mcmc(np.zeros((7, 3)), 2)
    with pytest.raises(ValueError):                       # degenerate ensemble
        s.run_mcmc(np.ones((8, 3)), 2)
    with pytest.raises(RuntimeError):                     # fewer walkers than 2*ndim
        EnsembleSampler(4, 3, gaussian_logp, args=(np.zeros(3), np.eye(3))).run_mcmc(
            np.random.randn(4, 3), 1)
    with pytest.raises(ValueError):                       # NaN from the log-prob
        EnsembleSampler(8, 3, lambda t: np.full(len(t), np.nan)).run_mcmc(np.random.randn(8, 3), 1)
    with pytest.raises(AttributeError):
        s.get_chain()
    assert walkers_independent(np.random.randn(10, 3))
    assert not walkers_independent(np.zeros((10, 3)))


def _walkers_independent_by_singular_values(coords):
    """emcee's test as it reads (the published algorithm): the yardstick of the shortcut."""
    coords = np.asarray(coords, dtype=np.float64)
    if not np.all(np.isfinite(coords)):
        return False
    c = coords - coords.mean(axis=0)[None, :]
    colmax = np.abs(c).max(axis=0)
    if np.any(colmax == 0):
        return False
    c = c / colmax
    c = c / np.sqrt((c ** 2).sum(axis=0))
    return bool(np.linalg.cond(c) <= 1e8)


def test_walkers_independent_decides_as_the_singular_values_do():
    """Big ensembles take the condition number from the correlation matrix where that is safe (far inside the
    limit) and from the singular values elsewhere: the answer is the singular values' in every case -- balls
    and boxes, columns of very different scale and offset, condition numbers from 1 to 1e12 (both sides of the
    1e8 limit, and the band the shortcut must hand over), duplicated and constant columns, NaN / inf, columns
    whose squares under- or overflow."""
    rng = np.random.RandomState(5)
    cases = []
    for W, nd in ((512, 3), (2048, 7), (40000, 7), (70000, 13)):
        ball = np.array(rng.uniform(-12, 12, nd)) + 1e-4 * rng.randn(W, nd)
        cases.append(('ball', ball))
        cases.append(('box', rng.uniform(-1, 1, (W, nd)) * 10.0 ** rng.uniform(-6, 6, nd) + 1e3))
        for k in (1e1, 1e3, 1e5, 1e7, 3e7, 3e8, 1e10, 1e12):
            q, _ = np.linalg.qr(rng.randn(nd, nd))
            x = rng.randn(W, nd) * np.logspace(0, -np.log10(k), nd)
            cases.append((f'cond {k:g}', x @ q.T + rng.uniform(-5, 5, nd)))
        dup = rng.randn(W, nd)
        dup[:, -1] = dup[:, 0]
        cases.append(('duplicated column', dup))
        const = rng.randn(W, nd)
        const[:, 1] = 3.0
        cases.append(('constant column', const))
        bad = rng.randn(W, nd)
        bad[7, 0] = np.nan
        cases.append(('nan', bad))
        bad = rng.randn(W, nd)
        bad[9, 1] = np.inf
        cases.append(('inf', bad))
        tiny = rng.randn(W, nd)
        tiny[:, 0] *= 1e-170
        cases.append(('squares underflow', tiny))
        huge = rng.randn(W, nd)
        huge[:, 0] *= 1e160
        cases.append(('squares overflow', huge))
    from bisip_amd.sampler import gram_decides, gram_from_shifted_sums
    seen, settled = set(), 0
    for name, x in cases:
        want = _walkers_independent_by_singular_values(x)
        assert walkers_independent(x) == want, (name, x.shape)
        seen.add(want)
        # the device route of big ensembles (bisip_ensemble_gram_dev): shifted sums and second moments, here formed
        # by NumPy as the kernel forms them; what they settle is settled as the singular values would, the rest
        # goes to the host's test
        W, nd = x.shape
        with np.errstate(all='ignore'):
            d = x - x[0]
            sums = np.concatenate([d.sum(axis=0), (d.T @ d)[np.triu_indices(nd)]])
            decided = gram_decides(gram_from_shifted_sums(sums, W, nd), W)
        assert not decided or want, name                 # "independent" from the moments is never wrong
        settled += decided
    assert seen == {True, False} and settled >= 12      # and the moments do settle the ordinary ensembles


def test_minus_inf_proposals_are_always_rejected():
    """Walkers start inside the box; -inf (out-of-prior) proposals never get accepted."""
    g = np.load([p for p in golden_cases() if 'PeltonColeCole_SIP-K389175' in p][0])
    prob = oracle.OracleProblem.from_golden(g, 'PeltonColeCole')
    np.random.seed(5)
    lo, hi = g['bounds']
    p0 = np.random.uniform(lo, hi, (16, lo.size))
    s = EnsembleSampler(16, lo.size, lambda t: oracle.logprob(prob, t))
    s.run_mcmc(p0, 60)
    ch = s.get_chain(flat=True)
    assert np.all(ch > lo) and np.all(ch < hi)
    assert np.all(np.isfinite(s.get_log_prob()))
