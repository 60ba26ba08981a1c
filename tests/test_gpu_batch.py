"""Batch of independent spectra (BASELINE config 5: E spectra x Wp walkers, double
Cole-Cole) -- log-prob / forward against the oracle per spectrum, and the E-ensemble
device sampler against a NumPy replay of the same contract."""

import numpy as np
import pytest

import oracle
from conftest import assert_logp_close, assert_Z_close

pytestmark = pytest.mark.gpu


def _tables(E, n_freq=32):
    from bisip_amd.synthetic import synthetic_columns
    return [synthetic_columns(n_freq, i) for i in range(E)]


def _oracle_logp(batch, theta):
    out = np.empty(theta.shape[:2])
    kw = {}
    if batch.model == 'PolynomialDecomposition':
        kw = dict(taus=batch.taus, log_taus=batch.log_taus, c_exp=batch.c_exp)
    if batch.model == 'PeltonColeCole':
        kw = dict(n_modes=batch.n_modes)
    for e in range(batch.n_spectra):
        prob = oracle.OracleProblem(batch.model, batch.w[e], batch.zn[e], batch.zn_err[e],
                                    batch.param_bounds, **kw)
        out[e] = oracle.logprob(prob, theta[e])
    return out


@pytest.mark.parametrize('model,kw,E,Wp', [
    ('PeltonColeCole', dict(n_modes=2), 12, 256),      # cfg5 shape per spectrum, uniform waves
    ('PeltonColeCole', dict(n_modes=2), 5, 50),        # waves straddle spectra
    ('PolynomialDecomposition', dict(poly_deg=5), 9, 128),
    ('PolynomialDecomposition', dict(poly_deg=4, c_exp=0.5), 3, 30),
    ('Dias2000', {}, 4, 64),
    ('Shin2015', {}, 4, 20),
])
def test_batch_logprob_and_forward(model, kw, E, Wp):
    import bisip_amd
    batch = bisip_amd.SpectraBatch(model, _tables(E), nwalkers=Wp, **kw)
    rng = np.random.RandomState(E * 1000 + Wp)
    lo, hi = batch.param_bounds
    theta = rng.uniform(lo, hi, (E, Wp, lo.size))
    theta[0, 1, 0] = hi[0]                        # on-bound row -> -inf
    got = batch.log_prob(theta)
    assert got.shape == (E, Wp)
    assert_logp_close(got, _oracle_logp(batch, theta))
    if model == 'PolynomialDecomposition':
        batch.ctx.set_variant('collapsed')
        assert_logp_close(batch.log_prob(theta), _oracle_logp(batch, theta))
        with pytest.raises(RuntimeError):
            batch.ctx.set_variant('faithful')
        batch.ctx.set_variant('auto')
    Z = batch.forward(theta[:, :5])
    assert Z.shape == (E, 5, 2, 32)
    # spectrum e of the batch == a single-spectrum context on the same data
    from bisip_amd import _hip
    okw = dict(kw)
    if model == 'PolynomialDecomposition':
        okw = dict(poly_deg=batch.poly_deg, c_exp=batch.c_exp, taus=batch.taus, log_taus=batch.log_taus)
    e = E - 1
    model_id = {'PolynomialDecomposition': 0, 'PeltonColeCole': 1, 'Dias2000': 2, 'Shin2015': 3}[model]
    # (on 'auto' every spectrum of a batch runs the kernel a context of its own would run)
    single = _hip.HipContext(model_id, batch.w[e], batch.zn[e], batch.zn_err[e], batch.param_bounds, **okw)
    assert np.array_equal(single.logprob(theta[e]), got[e])
    assert_Z_close(Z[e], single.forward(theta[e, :5]), 1e-15)
    with pytest.raises(ValueError):
        batch.ctx.logprob(theta.reshape(-1, lo.size)[:E * Wp - 1])   # not a multiple of E


def test_batch_sampler_replay_and_independence():
    """E ensembles advanced by one launch per half-step == NumPy replay of the philox
    contract with the GPU log-prob; and each ensemble's chain does not depend on its
    neighbours (same seed, same spectrum position -> same chain)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bisip_amd
    from bisip_amd.sampler import DeviceEnsembleSampler
    from numpy_stretch_backend import NumpyStretchBackend
    for E, Wp in [(6, 128), (5, 30)]:
        batch = bisip_amd.SpectraBatch('PeltonColeCole', _tables(E), nwalkers=Wp, nsteps=40, n_modes=2)
        rng = np.random.RandomState(E)
        centre = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6])
        p0 = centre + 1e-3 * rng.randn(E, Wp, 7)
        np.random.seed(1)
        batch.fit(p0=p0, seed=77)
        chain = batch.get_chain()
        assert chain.shape == (40, E, Wp, 7)
        np.random.seed(1)
        rep = DeviceEnsembleSampler(Wp, 7, backend=NumpyStretchBackend(batch.ctx.logprob, E),
                                    rng='philox', seed=77, n_ensembles=E)
        rep.run_mcmc(p0.reshape(E * Wp, 7), 40)
        assert np.array_equal(chain.reshape(40, E * Wp, 7), rep.get_chain())
        assert 0.05 < batch.acceptance_fraction.mean() < 0.9
        # walkers never leave their own ensemble's posterior: compare with a smaller batch
        sub = bisip_amd.SpectraBatch('PeltonColeCole', _tables(E)[:2], nwalkers=Wp, nsteps=40, n_modes=2)
        np.random.seed(1)
        sub.fit(p0=p0[:2], seed=77)
        assert np.array_equal(sub.get_chain(), chain[:, :2])
        assert batch.get_chain(discard=10, flat=True).shape == (E, 30 * Wp, 7)


def test_cfg5_scaled_shape_runs():
    """A slice of BASELINE config 5 (512 spectra per GPU at full scale): 64 spectra x 256
    walkers of double Cole-Cole, a few steps; finite log-probs, in-prior positions."""
    import bisip_amd
    E, Wp = 64, 256
    batch = bisip_amd.SpectraBatch('PeltonColeCole', _tables(E), nwalkers=Wp, nsteps=20, n_modes=2)
    rng = np.random.RandomState(0)
    centre = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6])
    p0 = centre + 1e-3 * rng.randn(E, Wp, 7)
    batch.fit(p0=p0, seed=5)
    lp = batch.get_log_prob()
    assert lp.shape == (20, E, Wp) and np.isfinite(lp).all()
    lo, hi = batch.param_bounds
    ch = batch.get_chain()
    assert np.all(ch > lo) and np.all(ch < hi)
    assert lp[-1].mean() > lp[0].mean() - 50


def test_batch_persistent_equals_launch_path():
    """E ensembles: one workgroup each (persistent) == one launch per half-step over all."""
    import bisip_amd
    from bisip_amd.sampler import DeviceEnsembleSampler
    for model, kw, ndim, variant in [('PeltonColeCole', dict(n_modes=2), 7, 'auto'), ('PolynomialDecomposition', {}, 7, 'auto'),
                                     ('PolynomialDecomposition', {}, 7, 'reduced_comp'), ('PolynomialDecomposition', {}, 7, 'collapsed'),
                                     ('Dias2000', {}, 5, 'auto')]:
        for E, Wp in [(7, 64), (3, 30), (5, 256)]:
            batch = bisip_amd.SpectraBatch(model, _tables(E), nwalkers=Wp, nsteps=12, **kw)
            batch.ctx.set_variant(variant)
            rng = np.random.RandomState(E + Wp)
            if model == 'PeltonColeCole':
                centre = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6])
            elif model == 'Dias2000':
                centre = np.array([1.0, 0.5, -8.0, 10.0, 0.5])
            else:
                centre = np.array([1.0, 0.005, -0.003, -0.001, 0.0005, 0.0002, 0.00001])
            p0 = (centre + 1e-4 * rng.randn(E, Wp, ndim)).reshape(E * Wp, ndim)
            chains = []
            for persistent in (True, False):
                np.random.seed(4)
                s = DeviceEnsembleSampler(Wp, ndim, batch.ctx, rng='philox', seed=31, n_ensembles=E,
                                          persistent=persistent)
                s.run_mcmc(p0, 12)
                chains.append((s.get_chain(), s.get_log_prob(), s.acceptance_fraction, s.last_path))
            assert chains[0][3] == 'persistent' and chains[1][3] == 'launch-per-half-step'
            for x, y in zip(chains[0][:3], chains[1][:3]):
                assert np.array_equal(x, y)


def test_every_spectrum_of_a_batch_runs_the_tier_its_own_context_would(monkeypatch):
    """A PolynomialDecomposition batch on 'auto' decides plain / compensated PER SPECTRUM inside one launch
    (BatchArgs::tier): the spectra that pass the plain estimate -- most of a survey at the headline's shape --
    no longer pay for the few that do not.  Each spectrum's log-probabilities are the bits of a single-spectrum context on 'auto'
    -- from the bulk batch kernels (whole workgroups per spectrum or not), from the launch-per-half-step
    sampler and from the persistent one; a forced variant still runs one tier for all; and when
    bisip_logprob's guard finds the batch wanting it first closes the mix (every spectrum compensated)."""
    import warnings
    import bisip_amd
    from bisip_amd import _hip
    from bisip_amd.sampler import DeviceEnsembleSampler
    E = 24
    tables = _tables(E, 20)                    # degree 5, c = 0.5, 20 frequencies: 4 of these 24 pass the plain estimate
    for Wp in (128, 40):                       # 128: the streaming kernel (one spectrum per workgroup); 40: rows of two spectra in a wave
        batch = bisip_amd.SpectraBatch('PolynomialDecomposition', tables, nwalkers=Wp, nsteps=8, poly_deg=5, c_exp=0.5)
        n_plain, n_comp = batch.ctx.reduced_tiers
        assert n_plain + n_comp == E and n_plain >= 2 and n_comp >= 2, (n_plain, n_comp)
        assert batch.ctx.variant == 'reduced_comp' and batch.ctx.reduced_error <= 1e-12
        lo, hi = batch.param_bounds
        rng = np.random.RandomState(Wp)
        theta = rng.uniform(lo, hi, (E, Wp, lo.size))
        theta[:, : Wp // 2, 1:] *= 1e-2
        got = batch.log_prob(theta)
        tiers = []
        for e in range(E):
            single = _hip.HipContext(0, batch.w[e], batch.zn[e], batch.zn_err[e], batch.param_bounds, poly_deg=5,
                                     c_exp=batch.c_exp, taus=batch.taus, log_taus=batch.log_taus)
            tiers.append(single.variant)
            assert np.array_equal(single.logprob(theta[e]), got[e]), (e, single.variant)
            single.close()
        assert tiers.count('reduced') == n_plain and tiers.count('reduced_comp') == n_comp
        # the samplers: persistent == launches, and the stored log-probabilities are the bulk kernel's bits
        centre = np.array([1.0, 0.005, -0.003, -0.001, 0.0005, 0.0002, 0.00001])
        p0 = (centre + 1e-4 * rng.randn(E, Wp, 7)).reshape(E * Wp, 7)
        chains = []
        for persistent in (True, False):
            s = DeviceEnsembleSampler(Wp, 7, batch.ctx, rng='philox', seed=2, n_ensembles=E, persistent=persistent)
            s.run_mcmc(p0, 8)
            chains.append((s.get_chain(), s.get_log_prob(), s.last_path))
        assert chains[0][2] == 'persistent' and chains[1][2] == 'launch-per-half-step'
        assert np.array_equal(chains[0][0], chains[1][0]) and np.array_equal(chains[0][1], chains[1][1])
        last = chains[0][0][-1].reshape(E, Wp, 7)
        assert np.array_equal(batch.ctx.logprob(last.reshape(-1, 7)), chains[0][1][-1].ravel())
        # a forced variant: one tier for all
        batch.ctx.set_variant('reduced_comp')
        assert batch.ctx.reduced_tiers == (0, E)
        comp_all = batch.ctx.logprob(theta.reshape(-1, 7)).reshape(E, Wp)
        plain_spectra = [e for e in range(E) if tiers[e] == 'reduced']
        assert any(not np.array_equal(comp_all[e], got[e]) for e in plain_spectra)      # different arithmetic ...
        assert np.max(np.abs(comp_all - got) / np.maximum(1.0, np.abs(got))) <= 1e-11  # ... the same number
        batch.ctx.set_variant('auto')
        assert batch.ctx.reduced_tiers == (n_plain, n_comp)
        assert np.array_equal(batch.ctx.logprob(theta.reshape(-1, 7)).reshape(E, Wp), got)


def test_compensated_operands_follow_each_spectrum_s_own_frequency_list():
    """The compensated tier's operands come from a QR in binary128 whose kernel sums are computed once per
    DISTINCT frequency list of the spectra that need them (bisip_hip.hip: make_quad_operands).  A batch whose
    spectra alternate between three lists -- all of degree 8, so every one needs the compensated kernel --
    gives each spectrum the bits of a context of its own, and stays within 2e-11 of the yardstick."""
    import bisip_amd
    from bisip_amd import _hip
    E = 9
    tables = _tables(E, 24)
    for e in range(E):
        tables[e] = tables[e].copy()
        tables[e][:, 0] *= (1.0, 1.013, 0.97)[e % 3]
    batch = bisip_amd.SpectraBatch('PolynomialDecomposition', tables, nwalkers=64, nsteps=4, poly_deg=8, c_exp=0.5)
    assert batch.ctx.variant == 'reduced_comp' and batch.ctx.reduced_tiers[1] >= E - 1
    lo, hi = batch.param_bounds
    rng = np.random.RandomState(3)
    theta = rng.uniform(lo, hi, (E, 64, lo.size))
    theta[:, :32, 1:] *= 1e-3
    got = batch.log_prob(theta)
    for e in range(E):
        single = _hip.HipContext(0, batch.w[e], batch.zn[e], batch.zn_err[e], batch.param_bounds, poly_deg=8,
                                 c_exp=batch.c_exp, taus=batch.taus, log_taus=batch.log_taus)
        assert np.array_equal(single.logprob(theta[e]), got[e]), (e, single.variant)
        exact = _hip.polydecomp_reduced_reference(batch.w[e], batch.zn[e], batch.zn_err[e], batch.taus, batch.log_taus,
                                                  batch.c_exp, theta[e])
        assert np.max(np.abs(got[e] - exact) / np.maximum(1.0, np.abs(exact))) <= 2e-11
        single.close()
    batch.close()


def test_guard_closes_a_batch_mix_before_it_closes_a_tier(monkeypatch):
    """A batch whose spectra run different tiers and whose plain spectra are then measured > 2e-11 off on the
    caller's rows: bisip_logprob's guard first sends EVERY spectrum through the compensated kernel (one
    escalation), re-evaluates, and is satisfied.  The batch is built with the estimate's shell probes at a
    fifth of their weight (BISIP_SHELL_WEIGHT=0.01), so that 15 of 16 degree-6 designs keep the plain tier
    although they read 4-8e-11 on the shell; the rows: each spectrum's own shell logp = 0."""
    import bisip_amd
    from bisip_amd import _hip
    from test_gpu_parity import _shell_rows
    E, Wp = 16, 256
    tables = _tables(E, 64)
    monkeypatch.setenv('BISIP_SHELL_WEIGHT', '0.01')
    batch = bisip_amd.SpectraBatch('PolynomialDecomposition', tables, nwalkers=Wp, nsteps=4, poly_deg=6)
    monkeypatch.delenv('BISIP_SHELL_WEIGHT')
    n_plain, n_comp = batch.ctx.reduced_tiers
    assert n_plain >= 4 and n_comp >= 1 and batch.ctx.variant == 'reduced_comp', (n_plain, n_comp)
    bounds = np.array(batch.param_bounds)
    theta = np.empty((E, Wp, 8))
    for e in range(E):
        ops = _hip.polydecomp_operands(batch.w[e], batch.zn[e], batch.zn_err[e], batch.taus, batch.log_taus, batch.c_exp)
        rows = _shell_rows(ops, bounds, 4000, e)
        assert len(rows) >= Wp
        theta[e] = rows[:Wp]
    with pytest.warns(RuntimeWarning, match='k_logprob_pd_reduced_comp'):
        got = batch.ctx.logprob(theta.reshape(-1, 8))
    checks, worst, moves = batch.ctx.reduced_guard()
    assert moves == 1 and checks == 2 and worst > 2e-11
    assert batch.ctx.reduced_tiers == (0, E) and batch.ctx.variant == 'reduced_comp'
    assert batch.ctx.reduced_check(theta.reshape(-1, 8), got) <= 2e-12
    forced = bisip_amd.SpectraBatch('PolynomialDecomposition', tables, nwalkers=Wp, nsteps=4, poly_deg=6)
    forced.ctx.set_variant('reduced_comp')
    assert np.array_equal(forced.ctx.logprob(theta.reshape(-1, 8)), got)
    # a new prior box starts over: the mix is back
    wide = bounds.copy()
    wide[:, 1:] *= 1.01
    monkeypatch.setenv('BISIP_SHELL_WEIGHT', '0.01')
    batch.ctx.set_bounds(wide)
    monkeypatch.delenv('BISIP_SHELL_WEIGHT')
    assert batch.ctx.reduced_tiers[0] >= 1


@pytest.mark.parametrize('n_freq', [8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 21, 23, 29, 32, 35])
def test_persistent_kernel_steps_a_geometric_grid_like_the_launches_do(n_freq):
    """The persistent kernel reads a spectrum's records from LDS two frequencies at a time; on a geometric
    grid a block of sixteen frequencies is eight such pairs, and with several lanes per walker each lane takes
    whole quarters of a block (kernels.h: logprob_sums_grid).  Every block tail -- one to seven pairs, with or without a
    single frequency after them -- and every ensemble size (1, 2, 4 lanes per walker) gives the bits of the
    launch path."""
    import bisip_amd
    from bisip_amd.sampler import DeviceEnsembleSampler
    for model, kw, centre in [('PeltonColeCole', dict(n_modes=2), [1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]),
                              ('PeltonColeCole', dict(n_modes=1), [1.0, 0.3, -3.0, 0.5]),
                              ('Shin2015', {}, [0.5, 0.5, -14.0, -6.0, 0.5, 0.5])]:
        for E, Wp in ((3, 64), (2, 256), (1, 32), (1, 512)):
            ndim = len(centre)
            batch = bisip_amd.SpectraBatch(model, _tables(E, n_freq), nwalkers=Wp, nsteps=8, **kw)
            assert batch.ctx.loop_flags == 3
            rng = np.random.RandomState(n_freq)
            p0 = (np.array(centre) + 1e-4 * rng.randn(E, Wp, ndim)).reshape(E * Wp, ndim)
            chains = []
            for persistent in (True, False):
                s = DeviceEnsembleSampler(Wp, ndim, batch.ctx, rng='philox', seed=5, n_ensembles=E, persistent=persistent)
                s.run_mcmc(p0, 8)
                chains.append((s.get_chain(), s.get_log_prob(), s.last_path))
            assert chains[0][2] == 'persistent' and chains[1][2] == 'launch-per-half-step'
            assert np.array_equal(chains[0][0], chains[1][0]) and np.array_equal(chains[0][1], chains[1][1])
            assert np.isfinite(chains[0][1]).all()
            # and the bulk launch (one lane per walker) gives those bits too
            last = chains[0][0][-1].reshape(E * Wp, ndim)
            lp = batch.ctx.logprob(np.tile(last.reshape(E, Wp, ndim), (1, 40, 1)).reshape(-1, ndim))
            assert np.array_equal(lp.reshape(E, -1)[:, :Wp].ravel(), chains[0][1][-1].ravel())


# ----------------------------------------------------------------------------------
# chain kept in HBM + posterior moments on the device (get_param_mean / get_param_std,
# src/bisip/utils.py:55-85, per spectrum)
# ----------------------------------------------------------------------------------

MOMENT_TOL = 1e-12     # |d| <= tol * max(1, |value|): same numbers, different summation order


def _close(a, b, tol=MOMENT_TOL):
    assert a.shape == b.shape
    assert np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))), np.max(np.abs(a - b))


def test_device_chain_is_the_host_chain_and_moments_match_numpy():
    import bisip_amd
    E, Wp = 6, 32
    rng = np.random.RandomState(11)
    centre = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6])
    p0 = centre + 1e-3 * rng.randn(E, Wp, 7)
    runs = {}
    for where in ('host', 'device'):
        b = bisip_amd.SpectraBatch('PeltonColeCole', _tables(E), nwalkers=Wp, nsteps=40, n_modes=2)
        b.fit(p0, seed=5, thin_by=2, chain=where)
        runs[where] = b
    dev, host = runs['device'], runs['host']
    # summaries first: nothing has been copied to the host yet
    for discard, thin in ((0, 1), (10, 1), (7, 3), (39, 1)):
        flat = host.get_chain(discard=discard, thin=thin, flat=True)           # (E, n, ndim)
        _close(dev.get_param_mean(discard=discard, thin=thin), flat.mean(axis=1))
        _close(dev.get_param_std(discard=discard, thin=thin), flat.std(axis=1))
        _close(host.get_param_mean(discard=discard, thin=thin), flat.mean(axis=1))
    with pytest.raises(ValueError):
        dev.get_param_mean(discard=40)
    assert np.array_equal(dev.get_chain(), host.get_chain())                   # same chain, bit for bit
    assert np.array_equal(dev.get_log_prob(), host.get_log_prob())
    assert np.array_equal(dev.acceptance_fraction, host.acceptance_fraction)


@pytest.mark.parametrize('n,E,Wp,ndim,thin', [
    (37, 1, 100, 4, 1),          # one ensemble: samples split over many workgroups
    (5, 3, 7, 10, 2),            # fewer rows than lanes, strided samples
    (64, 40, 256, 7, 1),
    (3, 2500, 4, 5, 1),          # more ensembles than the split target
    (2, 2, 3, 16, 1),            # ndim = BISIP_MAX_NDIM
])
def test_chain_moments_entry_point(n, E, Wp, ndim, thin):
    import torch
    from bisip_amd import _hip
    rng = np.random.RandomState(n * 7 + E)
    full = rng.standard_normal((n * thin, E * Wp, ndim)) * rng.uniform(0.1, 50, ndim) + rng.uniform(-20, 20, ndim)
    t = torch.from_numpy(full).cuda()
    mean = torch.empty((E, ndim), dtype=torch.float64, device='cuda')
    std = torch.empty_like(mean)
    work = torch.empty(_hip.chain_moments_workspace(n, E, ndim), dtype=torch.float64, device='cuda')
    first = thin - 1
    _hip.chain_moments_dev(t.data_ptr() + 8 * first * E * Wp * ndim, n, thin * E * Wp * ndim, E, Wp, ndim,
                           mean.data_ptr(), std.data_ptr(), work.data_ptr(),
                           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    used = full[first::thin].reshape(n, E, Wp, ndim).transpose(1, 0, 2, 3).reshape(E, n * Wp, ndim)
    _close(mean.cpu().numpy(), used.mean(axis=1))
    _close(std.cpu().numpy(), used.std(axis=1))
    with pytest.raises(ValueError):
        _hip.chain_moments_dev(t.data_ptr(), n, 1, E, Wp, ndim, mean.data_ptr(), std.data_ptr(),
                               work.data_ptr(), 0)


@pytest.mark.parametrize('model,kw,n_freq', [('PeltonColeCole', dict(n_modes=2), 32), ('PeltonColeCole', dict(n_modes=1), 20),
                                             ('PolynomialDecomposition', dict(poly_deg=5), 20), ('Dias2000', {}, 27),
                                             ('Shin2015', {}, 48), ('PolynomialDecomposition', dict(poly_deg=5), 48),
                                             ('PolynomialDecomposition', dict(poly_deg=8), 64)])
def test_batch_forward_whole_blocks_per_spectrum(model, kw, n_freq):
    """n a multiple of 64 rows per spectrum: the batch takes the tiled / whole-row forward
    kernels with a per-block record pointer; same numbers as the per-(row, frequency) kernel
    used for ragged batches, and the oracle's.  (PolynomialDecomposition at 48 / 64 frequencies: the
    16-frequency tiles, whose records come through the constant address space -- kernels.h: eval_const.)"""
    import bisip_amd
    E, n = 5, 128
    batch = bisip_amd.SpectraBatch(model, _tables(E, n_freq), nwalkers=n, **kw)
    rng = np.random.RandomState(n_freq)
    lo, hi = batch.param_bounds
    theta = rng.uniform(lo, hi, (E, n, lo.size))
    Z = batch.forward(theta)                       # uniform blocks
    assert Z.shape == (E, n, 2, n_freq)
    ragged = batch.forward(theta[:, :37])          # 37 rows per spectrum: blocks straddle spectra
    assert np.array_equal(ragged, Z[:, :37])
    okw = {}
    if model == 'PolynomialDecomposition':
        okw = dict(taus=batch.taus, log_taus=batch.log_taus, c_exp=batch.c_exp)
    if model == 'PeltonColeCole':
        okw = dict(n_modes=batch.n_modes)
    for e in range(E):
        prob = oracle.OracleProblem(batch.model, batch.w[e], batch.zn[e], batch.zn_err[e], batch.param_bounds, **okw)
        assert_Z_close(Z[e], oracle.forward(prob, theta[e]))


def test_batch_polydecomp_follows_a_changed_prior_box():
    """The reduced form's expansion point depends on the prior box; a batch context keeps one
    per spectrum on the device and must refresh them all when the bounds change."""
    import bisip_amd
    E, n = 7, 64
    batch = bisip_amd.SpectraBatch('PolynomialDecomposition', _tables(E), nwalkers=n, poly_deg=6)
    rng = np.random.RandomState(5)
    for scale in (1.0, 1e-2, 0.4):
        for name in batch.param_names[1:]:
            batch.params[name] = [-scale, scale]
        lo, hi = batch.param_bounds
        theta = rng.uniform(lo, hi, (E, n, lo.size))
        # degree 6: the plain triangle for the narrow boxes, the compensated one where the shell
        # log-probability = 0 lies inside the box (host_precompute.cpp: reduced_center)
        assert batch.ctx.variant in ('reduced', 'reduced_comp')
        assert_logp_close(batch.log_prob(theta), _oracle_logp(batch, theta))


@pytest.mark.parametrize('n,E,Wp,ndim,thin', [(37, 1, 100, 4, 1), (5, 3, 7, 10, 2), (40, 12, 64, 7, 1), (2, 2, 3, 16, 1),
                                              (1, 1, 1, 1, 1)])
def test_chain_percentiles_entry_point(n, E, Wp, ndim, thin):
    """np.percentile(..., axis=0) of every (ensemble, parameter) column, default 'linear' rule,
    computed where the chain lies (gather -> segmented radix sort -> interpolate)."""
    import torch
    from bisip_amd import _hip
    rng = np.random.RandomState(n * 11 + E)
    full = rng.standard_normal((n * thin, E * Wp, ndim)) * rng.uniform(0.1, 50, ndim) + rng.uniform(-20, 20, ndim)
    full[0, 0, 0] = full[-1, -1, 0]                       # a tie
    t = torch.from_numpy(full).cuda()
    p = np.array([0.0, 2.5, 50.0, 33.3, 97.5, 100.0])
    nbytes = _hip.chain_percentiles_workspace(n, E, Wp, ndim, p.size)
    work = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    out = torch.empty((p.size, E, ndim), dtype=torch.float64, device='cuda')
    first = thin - 1
    _hip.chain_percentiles_dev(t.data_ptr() + 8 * first * E * Wp * ndim, n, thin * E * Wp * ndim, E, Wp, ndim, p,
                               out.data_ptr(), work.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    used = full[first::thin].reshape(n, E, Wp, ndim).transpose(1, 0, 2, 3).reshape(E, n * Wp, ndim)
    want = np.percentile(used, p, axis=1)                # (n_p, E, ndim)
    _close(out.cpu().numpy(), want, 1e-14)
    with pytest.raises(ValueError):
        _hip.chain_percentiles_dev(t.data_ptr(), n, thin * E * Wp * ndim, E, Wp, ndim, [101.0], out.data_ptr(),
                                   work.data_ptr(), nbytes, 0)


def test_batch_percentiles_from_the_device_chain():
    import bisip_amd
    E, Wp = 5, 32
    rng = np.random.RandomState(3)
    centre = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6])
    p0 = centre + 1e-3 * rng.randn(E, Wp, 7)
    runs = {}
    for where in ('host', 'device'):
        b = bisip_amd.SpectraBatch('PeltonColeCole', _tables(E), nwalkers=Wp, nsteps=30, n_modes=2)
        b.fit(p0, seed=9, chain=where)
        runs[where] = b
    for kw in (dict(), dict(discard=10, thin=2)):
        got = runs['device'].get_param_percentile(**kw)
        want = runs['host'].get_param_percentile(**kw)
        assert got.shape == (3, E, 7)
        _close(got, want, 1e-14)
    _close(runs['device'].get_param_percentile([16, 84], discard=5), runs['host'].get_param_percentile([16, 84], discard=5), 1e-14)


def test_a_survey_split_over_ranks_gives_the_same_chains():
    """Whole-replica sharding: spectrum e's chain, summaries and acceptance
    are the same whether the survey runs as one batch or as blocks of spectra on several ranks --
    the Philox stream is keyed by the spectrum's index in the SURVEY (bisip_ctx_set_spectrum_offset)
    and the default starts are drawn for the whole survey.  The ranks run one after the other here
    (one GPU); they never exchange anything while sampling."""
    import bisip_amd
    E, Wp = 7, 64
    tables = _tables(E)

    def run(rank, world, p0=None):
        b = bisip_amd.SpectraBatch('PeltonColeCole', tables, nwalkers=Wp, nsteps=30, n_modes=2,
                                   rank=rank, world=world)
        np.random.seed(5)
        first, last = b.spectrum_range
        b.fit(p0=None if p0 is None else p0[first:last], seed=None if p0 is None else 9, chain='device')
        out = (b.get_chain(), b.get_param_mean(discard=10), b.acceptance_fraction)
        b.close()
        return out

    centre = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6])
    for p0 in (None, centre + 1e-3 * np.random.RandomState(0).randn(E, Wp, 7)):
        whole = run(0, 1, p0)
        for world in (2, 3):
            parts = [run(r, world, p0) for r in range(world)]
            assert np.array_equal(np.concatenate([p[0] for p in parts], axis=1), whole[0])
            assert np.array_equal(np.concatenate([p[1] for p in parts], axis=0), whole[1])
            assert np.array_equal(np.concatenate([p[2] for p in parts], axis=0), whole[2])
    # the offset is validated
    b = bisip_amd.SpectraBatch('PeltonColeCole', tables[:2], nwalkers=Wp, nsteps=1, n_modes=2)
    with pytest.raises(ValueError):
        b.ctx.set_spectrum_offset(-1)
    with pytest.raises(ValueError):
        b.ctx.set_spectrum_offset(2 ** 31 - 2)
    b.close()


def test_batch_operands_do_not_depend_on_threads_or_on_shared_frequencies(monkeypatch):
    """A batch context's per-spectrum operands are computed in blocks of spectra on host threads, and
    the kernel sums are reused while consecutive spectra share their frequency list: spectrum e must
    get exactly the operands a context of its own gets -- with one thread or many, and when the
    frequency lists differ from spectrum to spectrum or alternate (reuse, recompute, reuse ...)."""
    import bisip_amd
    from bisip_amd import _hip
    E, Wp = 37, 16
    tables = _tables(E)
    for e in range(E):
        if e % 5 in (2, 3):                       # runs of equal and of different frequency lists
            tables[e] = tables[e].copy()
            tables[e][:, 0] *= 1.0 + 0.01 * (e // 5)
    rng = np.random.RandomState(8)
    got = {}
    for threads in ('1', '3', None):
        if threads is None:
            monkeypatch.delenv('BISIP_HOST_THREADS', raising=False)
        else:
            monkeypatch.setenv('BISIP_HOST_THREADS', threads)
        batch = bisip_amd.SpectraBatch('PolynomialDecomposition', tables, nwalkers=Wp, poly_deg=4, c_exp=0.7)
        lo, hi = batch.param_bounds
        theta = np.random.RandomState(8).uniform(lo, hi, (E, Wp, lo.size))
        got[threads] = batch.log_prob(theta)
        v_default = 'auto'                        # every spectrum of a batch runs what a context of its own would run
        # a wider box re-centres every spectrum's reduced form (the other threaded loop)
        wide = batch.param_bounds.copy()
        wide[:, 1:] *= 1.5
        batch.ctx.set_bounds(wide)
        v_wide = 'auto'
        got[threads, 'wide'] = batch.ctx.logprob(theta.reshape(-1, lo.size)).reshape(E, Wp)
        if threads is None:
            assert_logp_close(got[threads], _oracle_logp(batch, theta))
            for e in (0, 2, 3, 4, 12, 13, 36):
                for bounds, key, v in ((batch.param_bounds, None, v_default), (wide, (None, 'wide'), v_wide)):
                    single = _hip.HipContext(0, batch.w[e], batch.zn[e], batch.zn_err[e], bounds, poly_deg=4,
                                             c_exp=0.7, taus=batch.taus, log_taus=batch.log_taus, variant=v)
                    assert np.array_equal(single.logprob(theta[e]), got[key][e])
                    single.close()
        batch.close()
    for key in ('1', '3'):
        assert np.array_equal(got[key], got[None]) and np.array_equal(got[key, 'wide'], got[None, 'wide'])


def test_many_percentiles_of_few_long_columns(monkeypatch):
    """More than 8 percentiles: groups of 8 by selection when the columns are few (7 columns of a long chain),
    one segmented sort when they are many -- NumPy's doubles either way."""
    import torch
    from bisip_amd import _hip
    rng = np.random.RandomState(4)
    p = np.concatenate([np.linspace(0, 100, 21), [2.5, 97.5]])
    for n, E, Wp, ndim in ((900, 1, 50, 7), (30, 16, 40, 7)):            # 7 columns of 45,000 values; 112 columns of 1,200
        full = rng.standard_normal((n, E * Wp, ndim)) * rng.uniform(0.1, 5, ndim) + rng.uniform(-2, 2, ndim)
        t = torch.from_numpy(full).cuda()
        nbytes = _hip.chain_percentiles_workspace(n, E, Wp, ndim, p.size)
        work = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
        used = full.reshape(n, E, Wp, ndim).transpose(1, 0, 2, 3).reshape(E, n * Wp, ndim)
        want = np.percentile(used, p, axis=1)
        for force in (None, '1'):
            if force:
                monkeypatch.setenv('BISIP_PERCENTILE_SORT', force)
            else:
                monkeypatch.delenv('BISIP_PERCENTILE_SORT', raising=False)
            out = torch.full((p.size, E, ndim), float('nan'), dtype=torch.float64, device='cuda')
            _hip.chain_percentiles_dev(t.data_ptr(), n, E * Wp * ndim, E, Wp, ndim, p, out.data_ptr(), work.data_ptr(), nbytes,
                                       torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize('variant', ['reduced', 'reduced_comp'])
def test_bulk_batch_logprob_streams_like_the_single_spectrum_kernel(variant):
    """E x Wp >= 131072 rows with Wp a multiple of the workgroup: the batch takes the headline kernel's
    streaming structure (k_logprob_batch_reduced_stream).  Spectrum e's block must equal, bit for bit,
    what a context of that spectrum alone returns, and agree with the oracle on a sample; a ragged
    batch of the same spectra (Wp not a multiple) takes the general kernel and gives the same bits."""
    import torch
    from bisip_amd import _hip
    import bisip_amd
    E, Wp = 3, 65536
    batch = bisip_amd.SpectraBatch('PolynomialDecomposition', _tables(E), nwalkers=64, poly_deg=5)
    batch.ctx.set_variant(variant)
    lo, hi = batch.param_bounds
    rng = np.random.RandomState(17)
    theta = rng.uniform(lo, hi, (E, Wp, lo.size))
    theta[1, 5, 2] = hi[2]                                   # on a bound
    got = batch.log_prob(theta)
    assert got.shape == (E, Wp) and np.isneginf(got[1, 5]) and np.isfinite(np.delete(got.ravel(), Wp + 5)).all()
    pick = rng.choice(Wp, 300, replace=False)
    assert_logp_close(got[:, pick], _oracle_logp(batch, theta[:, pick]))
    for e in range(E):
        single = _hip.HipContext(0, batch.w[e], batch.zn[e], batch.zn_err[e], batch.param_bounds, poly_deg=5,
                                 c_exp=batch.c_exp, taus=batch.taus, log_taus=batch.log_taus, variant=variant)
        assert np.array_equal(single.logprob(theta[e]), got[e])
        single.close()
    ragged = batch.log_prob(theta[:, :Wp - 3])               # Wp - 3 walkers per spectrum: the general kernel
    assert np.array_equal(ragged, got[:, :Wp - 3])
    batch.close()


@pytest.mark.parametrize('model,kw,centre,Wp', [
    ('PeltonColeCole', dict(n_modes=2), [1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6], 48),       # one forward launch per spectrum
    ('PolynomialDecomposition', dict(poly_deg=4), [1.0, 0.005, -0.003, -0.001, 0.0005, 0.0002], 48),
    ('PeltonColeCole', dict(n_modes=2), [1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6], 64),       # whole 64-row blocks: one launch per pass
    ('Dias2000', {}, [1.0, 0.5, -8.0, 10.0, 0.5], 128),
])
def test_batch_model_percentiles_per_spectrum(model, kw, centre, Wp):
    """SpectraBatch.get_model_percentile: per spectrum, np.percentile over axis 0 of the model response
    over that spectrum's flattened chain (the reference's get_model_percentile, src/bisip/utils.py:17-35,
    which a survey would call file by file) -- computed spectrum by spectrum on the device, the same
    from a chain kept in HBM and from one copied to the host."""
    import bisip_amd
    from bisip_amd import _hip
    E = 5
    tables = _tables(E)
    p0 = np.asarray(centre) + 1e-4 * np.random.RandomState(1).randn(E, Wp, len(centre))
    got = {}
    for chain in ('device', 'host'):
        b = bisip_amd.SpectraBatch(model, tables, nwalkers=Wp, nsteps=40, **kw)
        b.fit(p0, seed=6, chain=chain)
        got[chain] = b.get_model_percentile([2.5, 50, 97.5], discard=10, thin=3)
        assert got[chain].shape == (3, E, 2, 32)
        if chain == 'host':
            flat = b.get_chain(discard=10, thin=3, flat=True)            # (E, n*Wp, ndim)
            okw = dict(kw)
            if model == 'PolynomialDecomposition':
                okw = dict(poly_deg=b.poly_deg, c_exp=b.c_exp, taus=b.taus, log_taus=b.log_taus)
            mid = {'PolynomialDecomposition': 0, 'PeltonColeCole': 1, 'Dias2000': 2}[model]
            for e in range(E):
                single = _hip.HipContext(mid, b.w[e], b.zn[e], b.zn_err[e], b.param_bounds, **okw)
                want = np.percentile(single.forward(flat[e]), [2.5, 50, 97.5], axis=0)
                np.testing.assert_allclose(got[chain][:, e], want, rtol=1e-13, atol=1e-15)
                single.close()
            with pytest.raises(ValueError):
                b.get_model_percentile(50, discard=40)
            assert b.get_model_percentile(50, discard=10).shape == (1, E, 2, 32)
        b.close()
    assert np.array_equal(got['device'], got['host'])


def test_forward_over_a_range_of_spectra():
    """bisip_forward_spectra_dev: rows of n consecutive spectra of a batch context in one launch -- each spectrum's
    block is what a context of that spectrum alone computes; ranges and ragged row counts are checked."""
    import torch
    import bisip_amd
    from bisip_amd import _hip
    E, rows = 6, 128
    b = bisip_amd.SpectraBatch('PeltonColeCole', _tables(E), nwalkers=64, n_modes=1)
    lo, hi = b.param_bounds
    th = torch.from_numpy(np.random.RandomState(2).uniform(lo, hi, (3, rows, lo.size))).cuda()
    Z = torch.empty((3, rows, 2, 32), dtype=torch.float64, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    b.ctx.forward_spectra_dev(2, 3, th.data_ptr(), 3 * rows, Z.data_ptr(), st)          # spectra 2, 3, 4
    torch.cuda.synchronize()
    for k, e in enumerate((2, 3, 4)):
        single = _hip.HipContext(1, b.w[e], b.zn[e], b.zn_err[e], b.param_bounds, n_modes=1)
        assert np.array_equal(single.forward(th[k].cpu().numpy()), Z[k].cpu().numpy())
        single.close()
    for first, count, W in ((4, 3, 3 * rows), (-1, 1, rows), (0, 0, rows), (0, 2, 2 * rows - 2), (0, 2, 2 * 100)):
        with pytest.raises(ValueError):
            b.ctx.forward_spectra_dev(first, count, th.data_ptr(), W, Z.data_ptr(), st)
    b.ctx.forward_spectrum_dev(5, th.data_ptr(), 100, Z.data_ptr(), st)                  # one spectrum: any row count
    torch.cuda.synchronize()
    b.close()


@pytest.mark.parametrize('n,E,Wp,ndim,kind', [(40, 12, 64, 7, 'normal'), (3, 16, 5, 4, 'normal'), (1, 64, 1, 1, 'normal'),
                                               (25, 10, 33, 8, 'ties'), (60, 9, 50, 9, 'signs'), (7, 70, 300, 1, 'tight'),
                                               # columns of more than 40,960 values select from memory, not from registers
                                               (700, 10, 64, 7, 'normal'), (650, 64, 70, 1, 'tight'), (41, 8, 1000, 8, 'signs'),
                                               # a NaN in a column makes that column's percentiles NaN, as in NumPy
                                               (40, 12, 64, 7, 'nans'), (700, 10, 64, 7, 'nans')])
def test_percentiles_by_selection_equal_the_sorted_ones(n, E, Wp, ndim, kind, monkeypatch):
    """With enough columns the percentiles come from a radix SELECTION of the two order statistics each
    needs (k_segmented_select) instead of a segmented sort: same doubles as the sort path (forced with
    BISIP_PERCENTILE_SORT=1), and NumPy's -- on negative and positive values, zeros, exact ties,
    infinities, columns that differ only in their last bits, and every rank from the minimum to the maximum."""
    import torch
    from bisip_amd import _hip
    rng = np.random.RandomState(n + E + ndim)
    full = rng.standard_normal((n, E * Wp, ndim)) * rng.uniform(0.1, 50, ndim) + rng.uniform(-20, 20, ndim)
    if kind == 'ties':
        full = np.round(full)                                  # many equal values, -0.0 and 0.0 among them
        full[full == 0] *= rng.choice([-1.0, 1.0], size=(full == 0).sum())
    elif kind == 'signs':
        full[::3] *= -1e-300
        full[1, 1, :] = np.inf
        full[2, 2, :] = -np.inf
        full[3, :, 0] = 0.0
    elif kind == 'tight':
        full = 1.0 + rng.randint(0, 1 << 20, size=full.shape) * 2.0 ** -52      # same exponent, same high mantissa bits
    elif kind == 'nans':
        full[5, 3, 0] = np.nan                                 # one NaN in column (ensemble 0, parameter 0)
        full[7, Wp + 1, 2] = -np.nan                           # sign bit set: sorts to the other end
        full[:, 2 * Wp:3 * Wp, 1] = np.nan                     # a column of nothing else
    assert E * ndim >= 64
    t = torch.from_numpy(full).cuda()
    p = np.array([0.0, 2.5, 50.0, 33.3, 97.5, 99.99, 100.0, 16.0])
    nbytes = _hip.chain_percentiles_workspace(n, E, Wp, ndim, p.size)
    work = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    outs = {}
    for mode in ('select', 'sort'):
        if mode == 'sort':
            monkeypatch.setenv('BISIP_PERCENTILE_SORT', '1')
        else:
            monkeypatch.delenv('BISIP_PERCENTILE_SORT', raising=False)
        out = torch.full((p.size, E, ndim), float('nan'), dtype=torch.float64, device='cuda')
        _hip.chain_percentiles_dev(t.data_ptr(), n, E * Wp * ndim, E, Wp, ndim, p, out.data_ptr(), work.data_ptr(), nbytes,
                                   torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs[mode] = out.cpu().numpy()
    assert np.array_equal(outs['select'], outs['sort'], equal_nan=True)
    used = full.reshape(n, E, Wp, ndim).transpose(1, 0, 2, 3).reshape(E, n * Wp, ndim)
    with np.errstate(invalid='ignore'):
        want = np.percentile(used, p, axis=1)
    ok = np.isfinite(want)
    assert np.array_equal(outs['select'][ok], want[ok])            # order statistics, weights and _lerp are NumPy's: same doubles
    assert np.array_equal(np.isnan(outs['select']), np.isnan(want)) or kind == 'signs'
    if kind == 'nans':
        assert np.isnan(want[:, 0, 0]).all() and np.isnan(want[:, 1, 2]).all() and np.isnan(want[:, 2, 1]).all()
        assert np.isnan(want).sum() == 3 * p.size


def test_forward_columns_and_their_percentiles():
    """bisip_forward_columns_dev writes the responses column-major (what the percentile kernels read);
    bisip_columns_percentiles_dev takes np.percentile of such columns.  Same values as forward() transposed,
    for whole and ragged row counts, a range of spectra and a single-spectrum context; NumPy's percentiles."""
    import torch
    import bisip_amd
    from bisip_amd import _hip
    E = 5
    b = bisip_amd.SpectraBatch('PeltonColeCole', _tables(E), nwalkers=64, n_modes=2)
    lo, hi = b.param_bounds
    st = torch.cuda.current_stream().cuda_stream
    for rows in (128, 77):
        th = torch.from_numpy(np.random.RandomState(rows).uniform(lo, hi, (3, rows, lo.size))).cuda()
        cols = torch.empty((3, 64, rows), dtype=torch.float64, device='cuda')
        b.ctx.forward_columns_dev(1, 3, th.data_ptr(), 3 * rows, cols.data_ptr(), st)            # spectra 1, 2, 3
        torch.cuda.synchronize()
        p = np.array([2.5, 50.0, 97.5])
        out = torch.empty((3, 3 * 64), dtype=torch.float64, device='cuda')
        _hip.columns_percentiles_dev(cols.data_ptr(), 3 * 64, rows, p, out.data_ptr(), st)
        torch.cuda.synchronize()
        for k, e in enumerate((1, 2, 3)):
            single = _hip.HipContext(1, b.w[e], b.zn[e], b.zn_err[e], b.param_bounds, n_modes=2)
            Z = single.forward(th[k].cpu().numpy())                                              # (rows, 2, N)
            assert np.array_equal(cols[k].cpu().numpy(), Z.reshape(rows, 64).T)
            assert np.array_equal(out.cpu().numpy().reshape(3, 3, 64)[:, k], np.percentile(Z.reshape(rows, 64), p, axis=0))
            if k == 0:       # a single-spectrum context takes the same call
                c1 = torch.empty((64, rows), dtype=torch.float64, device='cuda')
                single.forward_columns_dev(0, 1, th[k].data_ptr(), rows, c1.data_ptr(), st)
                torch.cuda.synchronize()
                assert np.array_equal(c1.cpu().numpy(), Z.reshape(rows, 64).T)
            single.close()
    with pytest.raises(ValueError):
        b.ctx.forward_columns_dev(3, 3, th.data_ptr(), 3 * rows, cols.data_ptr(), st)           # spectra 3..5 of 5
    with pytest.raises(ValueError):
        b.ctx.forward_columns_dev(0, 2, th.data_ptr(), 3 * 77, cols.data_ptr(), st)             # 231 rows over 2 spectra
    b.close()


# ----------------------------------------------------------------------------------
# the stored samples nearest to the shell logp = 0 (bisip_chain_shell_rows_dev) and the guard they serve
# ----------------------------------------------------------------------------------

@pytest.mark.parametrize('n,E,Wp,ndim,k,n_stride,kind', [
    (1, 1, 5000, 7, 192, 64, 'wide'),        # an initial ensemble: one sample, stride rows; split over workgroups
    (40, 1, 4096, 4, 256, 0, 'posterior'),   # one ensemble's chunk: clustered log-probabilities, split path
    (64, 1, 32768, 7, 256, 0, 'crossing'),   # 2M samples, a few hundred near the shell
    (37, 600, 64, 5, 6, 0, 'crossing'),      # a survey: one workgroup per ensemble
    (9, 3, 100, 7, 256, 0, 'wide'),          # few ensembles, k below the sample count
    (2, 2, 50, 3, 256, 8, 'wide'),           # fewer samples than k: every finite one is taken, the rest NaN
    (5, 1, 1000, 2, 16, 0, 'ties'),          # many equal log-probabilities at the k-th place
])
def test_shell_rows_selection_against_numpy(n, E, Wp, ndim, k, n_stride, kind):
    """Per ensemble the k stored samples of smallest |logp|: every sample nearer than the k-th (to the 24 bits
    the selection resolves) is there, k of them in all (ties != 0) or exactly the strictly nearer ones
    (ties == 0); every output row is a (theta, logp) pair of that ensemble; -inf / NaN never selected;
    unfilled slots are NaN rows; the stride rows are the first sample's evenly spaced walkers."""
    import torch
    from bisip_amd import _hip
    rng = np.random.RandomState(n * 31 + E + k)
    lp = {'wide': lambda: rng.standard_normal((n, E * Wp)) * 10.0 ** rng.uniform(-3, 6, (n, E * Wp)),
          'posterior': lambda: 384.5 - rng.chisquare(4, (n, E * Wp)) / 2,
          'crossing': lambda: 400.0 - np.abs(rng.standard_normal((n, E * Wp))) * np.linspace(3e5, 1, n)[:, None] ** 1.0,
          'ties': lambda: np.round(rng.standard_normal((n, E * Wp)) * 3.0) * 0.5 + 0.25}[kind]()
    lp.ravel()[rng.choice(lp.size, max(1, lp.size // 50), replace=False)] = -np.inf       # out-of-prior samples
    lp.ravel()[rng.choice(lp.size, 3, replace=False)] = np.nan
    chain = rng.standard_normal((n, E * Wp, ndim))
    d_chain, d_lp = torch.from_numpy(chain).cuda(), torch.from_numpy(lp).cuda()
    work = torch.empty(_hip.chain_shell_rows_workspace(E), dtype=torch.uint8, device='cuda')

    def key24(x):
        return (np.abs(x).view(np.uint64) & np.uint64(0x7fffffffffffffff)) >> np.uint64(39)
    for ties in (True, False):
        out = torch.full((E, k + n_stride, ndim + 1), 7.0, dtype=torch.float64, device='cuda')
        _hip.chain_shell_rows_dev(d_chain.data_ptr(), d_lp.data_ptr(), n, E, Wp, ndim, k, n_stride, out.data_ptr(),
                                  work.data_ptr(), torch.cuda.current_stream().cuda_stream, ties=ties)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for e in range(E):
            mine_lp = lp[:, e * Wp:(e + 1) * Wp].ravel()
            mine_th = chain[:, e * Wp:(e + 1) * Wp].reshape(-1, ndim)
            fin = np.isfinite(mine_lp)
            order = np.argsort(np.abs(mine_lp[fin]), kind='stable')
            want_abs = np.abs(mine_lp[fin])[order]
            rows = got[e, :k]
            filled = ~np.isnan(rows[:, ndim])
            assert np.isnan(rows[~filled]).all()
            sel_lp = rows[filled, ndim]
            assert np.isfinite(sel_lp).all()
            n_fin = int(fin.sum())
            if n_fin <= k:
                assert filled.sum() == n_fin and np.array_equal(np.sort(np.abs(sel_lp)), want_abs)
            else:
                thr = key24(want_abs[k - 1:k])[0]                  # the k-th nearest's 24-bit key
                strictly = want_abs[key24(want_abs) < thr]
                got_keys = key24(np.sort(np.abs(sel_lp)))
                assert np.array_equal(np.sort(np.abs(sel_lp))[:len(strictly)], strictly)       # all the nearer ones
                if ties:
                    assert filled.sum() == k and (got_keys[len(strictly):] == thr).all()
                else:
                    assert filled.sum() == len(strictly)
            # every selected row is a real sample of THIS ensemble: its theta goes with its log-probability
            lookup = {}
            for i in np.flatnonzero(fin):
                lookup.setdefault(mine_lp[i], []).append(i)
            for r in rows[filled]:
                assert any(np.array_equal(mine_th[i], r[:ndim]) for i in lookup[r[ndim]])
            for j in range(n_stride):
                w = Wp * j // n_stride
                assert np.array_equal(got[e, k + j, :ndim], chain[0, e * Wp + w])
                a, b = got[e, k + j, ndim], lp[0, e * Wp + w]
                assert a == b or (np.isnan(a) and np.isnan(b))
    with pytest.raises(ValueError):
        _hip.chain_shell_rows_dev(d_chain.data_ptr(), d_lp.data_ptr(), n, E, Wp, ndim, 0, 0, 1, work.data_ptr(), 0)


def test_batch_fit_is_guarded_per_spectrum(monkeypatch):
    """SpectraBatch.fit: every spectrum's stored samples nearest to the shell are measured before a chunk is
    kept.  Sixteen degree-6 designs whose plain estimates are let through (BISIP_AUTO_ERR_MAX, a test hook), the
    walkers of each started on its own shell logp = 0, where the plain triangle reads 4-8e-11: the batch ends on
    the compensated kernels, every stored log-probability of every spectrum is within 2e-11 of the yardstick,
    and the chains are those of a batch that ran the compensated kernels from the start."""
    import warnings
    import bisip_amd
    from bisip_amd import _hip
    from test_gpu_parity import _shell_rows
    E, Wp, nsteps = 16, 64, 30
    tables = _tables(E, 64)
    monkeypatch.setenv('BISIP_AUTO_ERR_MAX', '1e-9')
    batch = bisip_amd.SpectraBatch('PolynomialDecomposition', tables, nwalkers=Wp, nsteps=nsteps, poly_deg=6)
    monkeypatch.delenv('BISIP_AUTO_ERR_MAX')
    assert batch.ctx.reduced_tiers == (E, 0) and batch.ctx.variant == 'reduced'
    bounds = np.array(batch.param_bounds)
    p0 = np.empty((E, Wp, 8))
    for e in range(E):
        ops = _hip.polydecomp_operands(batch.w[e], batch.zn[e], batch.zn_err[e], batch.taus, batch.log_taus, batch.c_exp)
        rows = _shell_rows(ops, bounds, 4000, e)
        assert len(rows) >= Wp
        p0[e] = rows[:Wp]
    with warnings.catch_warnings():
        warnings.simplefilter('error', RuntimeWarning)
        batch.fit(p0, seed=3)
    g = batch._sampler.guard_
    assert batch.ctx.variant == 'reduced_comp' and batch.ctx.reduced_tiers == (0, E)
    assert g['escalations'] == 1 and g['reruns'] == 1 and g['rejected'] > 2e-11 and g['worst'] <= 2e-11
    chain, lp = batch.get_chain(), batch.get_log_prob()           # (nsteps, E, Wp, ndim), (nsteps, E, Wp)
    for e in range(E):
        exact = _hip.polydecomp_reduced_reference(batch.w[e], batch.zn[e], batch.zn_err[e], batch.taus, batch.log_taus,
                                                  batch.c_exp, np.ascontiguousarray(chain[:, e].reshape(-1, 8)))
        rel = np.abs(lp[:, e].ravel() - exact) / np.maximum(1.0, np.abs(exact))
        assert rel.max() <= 2e-11, (e, rel.max())
    forced = bisip_amd.SpectraBatch('PolynomialDecomposition', tables, nwalkers=Wp, nsteps=nsteps, poly_deg=6)
    forced.ctx.set_variant('reduced_comp')
    forced.fit(p0, seed=3)
    assert np.array_equal(forced.get_chain(), chain) and np.array_equal(forced.get_log_prob(), lp)
    batch.close()
    forced.close()
