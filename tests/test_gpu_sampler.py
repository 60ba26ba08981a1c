"""Device-resident stretch move (bisip_stretch_*_dev) against the host-loop sampler.

Both drivers consume the same RNG stream; the device path must reproduce the host
path's chain BIT FOR BIT (positions, log-probs, acceptance counts), for every model
and formulation, including odd ensemble sizes, chunked runs and the eval+apply
(sharded) kernels driven single-rank."""

import numpy as np
import pytest

from conftest import assert_logp_close, case_id, golden_cases
from test_gpu_parity import make_ctx

pytestmark = pytest.mark.gpu

CASES = [('case04_', 'PolynomialDecomposition', 'reduced'),
         ('case04_', 'PolynomialDecomposition', 'collapsed'),
         ('case13_', 'PolynomialDecomposition', 'reduced'),
         ('case15_', 'PeltonColeCole', 'auto'),
         ('case16_', 'PeltonColeCole', 'auto'),
         ('case17_', 'PeltonColeCole', 'auto'),
         ('case18_', 'Dias2000', 'auto'),
         ('case19_', 'Shin2015', 'auto')]


def _case(prefix):
    return [p for p in golden_cases() if prefix in p][0]


def _start(g, W, seed):
    rng = np.random.RandomState(seed)
    i0 = int(g['n_prior'])
    centre = g['theta'][i0]
    lo, hi = g['bounds']
    p0 = centre + 1e-3 * (hi - lo) * rng.randn(W, lo.size)
    return np.clip(p0, lo + 1e-6 * (hi - lo), hi - 1e-6 * (hi - lo))


@pytest.mark.parametrize('prefix,model,variant', CASES)
def test_device_sampler_equals_host_sampler(prefix, model, variant):
    from bisip_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
    g = np.load(_case(prefix))
    ctx = make_ctx(g, model, variant)
    ndim = g['bounds'].shape[1]
    for W, nsteps, chunk in [(32, 60, None), (33, 25, 7), (256, 12, 5)]:
        if W < 2 * ndim:
            continue
        p0 = _start(g, W, 100 + W)
        np.random.seed(99)
        host = EnsembleSampler(W, ndim, ctx.logprob)
        host.run_mcmc(p0, nsteps)
        np.random.seed(99)
        dev = DeviceEnsembleSampler(W, ndim, ctx, chunk=chunk)
        dev.run_mcmc(p0, nsteps)
        assert np.array_equal(dev.get_chain(), host.get_chain())
        assert np.array_equal(dev.get_log_prob(), host.get_log_prob())
        assert np.array_equal(dev.acceptance_fraction, host.acceptance_fraction)
        assert 0.05 < dev.acceptance_fraction.mean() < 0.95
        # continuing a run appends and keeps the streams aligned
        host.run_mcmc(None, 5)
        dev.run_mcmc(None, 5)
        assert np.array_equal(dev.get_chain(), host.get_chain())
        # a second run from a NEW initial state: acceptances and iterations both keep accumulating
        p1 = _start(g, W, 200 + W)
        host.run_mcmc(p1, 6)
        dev.run_mcmc(p1, 6)
        assert np.array_equal(dev.get_chain(), host.get_chain())
        assert np.array_equal(dev.acceptance_fraction, host.acceptance_fraction)
    ctx.close()


def test_eval_apply_path_single_rank():
    """The sharded kernels (eval -> [all-gather] -> apply) driven with world=1 give the
    same state as the fused half-step kernel."""
    import torch
    from bisip_amd.sampler import DeviceEnsembleSampler, HipStretchBackend

    class SplitBackend(HipStretchBackend):
        calls = 0

        def run(self, st, n_steps):            # per-iteration driver instead of the fused C loop
            W = st['coords'].shape[0]
            for k in range(n_steps):
                self.half(st, k, 0, (W + 1) // 2)
                self.half(st, k, 1, W // 2)

        def half(self, st, k, h, n_slots):
            SplitBackend.calls += 1
            # two "ranks" worth of blocks evaluated one after the other, then applied
            world = 3
            pad = -(-n_slots // world)
            gathered = self.zeros((world * pad, st['coords'].shape[1] + 2), torch.float64)
            base, extra = divmod(n_slots, world)
            lo = 0
            for r in range(world):
                hi = lo + base + (1 if r < extra else 0)
                self.eval(st, k, h, n_slots, lo, hi, gathered[r * pad:(r + 1) * pad])
                lo = hi
            self.apply(st, k, h, n_slots, gathered, pad, world)

    g = np.load(_case('case16_'))
    ctx = make_ctx(g, 'PeltonColeCole')
    W, ndim = 40, 7
    p0 = _start(g, W, 5)
    np.random.seed(3)
    a = DeviceEnsembleSampler(W, ndim, ctx)
    a.run_mcmc(p0, 30)
    np.random.seed(3)
    b = DeviceEnsembleSampler(W, ndim, backend=SplitBackend(ctx), persistent=False)
    b.run_mcmc(p0, 30)
    assert SplitBackend.calls == 60           # the eval/apply kernels really ran
    assert np.array_equal(a.get_chain(), b.get_chain())
    assert np.array_equal(a.get_log_prob(), b.get_log_prob())
    assert np.array_equal(a.acceptance_fraction, b.acceptance_fraction)
    # thinning: every 3rd iteration stored, same stream
    np.random.seed(3)
    c = DeviceEnsembleSampler(W, ndim, ctx)
    c.run_mcmc(p0, 10, thin_by=3)
    assert np.array_equal(c.get_chain(), a.get_chain()[2::3])
    assert np.array_equal(c.get_log_prob(), a.get_log_prob()[2::3])
    np.random.seed(3)
    d = DeviceEnsembleSampler(W, ndim, backend=SplitBackend(ctx), persistent=False)
    d.run_mcmc(p0, 10, thin_by=3)
    assert np.array_equal(d.get_chain(), c.get_chain())
    ctx.close()


def test_device_sampler_against_oracle_replay():
    import oracle
    from bisip_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
    g = np.load(_case('case10_'))   # PolynomialDecomposition, synthetic N=32 (metric shape)
    ctx = make_ctx(g, 'PolynomialDecomposition')
    prob = oracle.OracleProblem.from_golden(g, 'PolynomialDecomposition')
    W, ndim = 64, 7
    p0 = _start(g, W, 8)
    np.random.seed(21)
    dev = DeviceEnsembleSampler(W, ndim, ctx)
    dev.run_mcmc(p0, 200)
    np.random.seed(21)
    ref = EnsembleSampler(W, ndim, lambda t: oracle.logprob(prob, t))
    ref.run_mcmc(p0, 200)
    assert np.array_equal(dev.get_chain(), ref.get_chain())
    assert_logp_close(dev.get_log_prob(), ref.get_log_prob())


def test_out_of_prior_start_and_nan_detection():
    from bisip_amd.sampler import DeviceEnsembleSampler
    g = np.load(_case('case15_'))
    ctx = make_ctx(g, 'PeltonColeCole')
    lo, hi = g['bounds']
    p0 = _start(g, 32, 1)
    p0[0, 0] = hi[0] + 0.01       # one walker starts outside: logp -inf, must move in or stay
    np.random.seed(4)
    s = DeviceEnsembleSampler(32, 4, ctx)
    s.run_mcmc(p0, 50)
    lp = s.get_log_prob()
    assert np.isneginf(lp[0, 0]) or np.isfinite(lp[0, 0])
    assert np.isfinite(lp[-1, 1:]).all()
    with pytest.raises(ValueError):
        bad = p0.copy()
        bad[1, 1] = np.nan
        s.run_mcmc(bad, 2)
    ctx.close()


def test_non_finite_start_of_a_big_ensemble_is_refused():
    """A big single ensemble is uploaded without a second finiteness pass: _start_from relies on
    walkers_independent() returning False for any NaN / inf (its fast Gram-matrix branch cannot say True for
    them).  Through run_mcmc, 32,768 walkers: one NaN and one inf each raise before anything reaches the device."""
    from bisip_amd.sampler import DeviceEnsembleSampler, walkers_independent
    g = np.load(_case('case15_'))
    ctx = make_ctx(g, 'PeltonColeCole')
    W = 32768
    p0 = _start(g, W, 3)
    assert walkers_independent(p0)
    np.random.seed(4)
    s = DeviceEnsembleSampler(W, 4, ctx, rng='philox', seed=1)
    for poison in (np.nan, np.inf, -np.inf):
        bad = p0.copy()
        bad[W // 2 + 5, 2] = poison
        assert not walkers_independent(bad)
        with pytest.raises(ValueError):
            s.run_mcmc(bad, 2)
        assert s._dev is None or np.isfinite(s._dev['coords'].cpu().numpy()).all()
    s.run_mcmc(p0, 2)                                        # and a finite start still runs
    assert np.isfinite(s.get_chain()).all()
    ctx.close()


def test_independence_test_of_a_big_ensemble_runs_on_the_device():
    """emcee's initial-state test (condition number of the centred, column-scaled positions) for ensembles of
    16,384 walkers and more: the device forms the shifted sums and second moments of the uploaded ensemble
    (bisip_ensemble_gram_dev), the host decides from ndim (ndim + 3) / 2 numbers -- the same answers as the host
    test: a spanning ensemble runs, linearly dependent columns and a constant column raise emcee's error, and the
    moments are NumPy's to rounding."""
    import torch
    from bisip_amd import _hip
    from bisip_amd.sampler import DeviceEnsembleSampler, gram_from_shifted_sums, walkers_independent
    g = np.load(_case('case04_'))
    ctx = make_ctx(g, 'PolynomialDecomposition')
    ndim = g['bounds'].shape[1]
    W = 40001
    p0 = _start(g, W, 8)
    # the moments themselves
    t = torch.from_numpy(p0).cuda()
    out = torch.empty(ndim + ndim * (ndim + 1) // 2, dtype=torch.float64, device='cuda')
    work = torch.empty(_hip.ensemble_gram_workspace(W, ndim), dtype=torch.float64, device='cuda')
    _hip.ensemble_gram_dev(t.data_ptr(), W, ndim, out.data_ptr(), work.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    G = gram_from_shifted_sums(out.cpu().numpy(), W, ndim)
    c = p0 - p0.mean(axis=0)
    want = c.T @ c
    assert np.abs(G - want).max() <= 1e-9 * np.abs(want).max()
    with pytest.raises(RuntimeError):
        _hip.ensemble_gram_dev(t.data_ptr(), W, 9, out.data_ptr(), work.data_ptr(), 0)
    # through run_mcmc
    s = DeviceEnsembleSampler(W, ndim, ctx, rng='philox', seed=3)
    s.run_mcmc(p0, 2)
    assert s.timing['check_s'] < 0.5 * 1e-3 * W / 16384 and np.isfinite(s.get_log_prob()).all()      # no pass over the ensemble on the host
    for spoil in ('dependent', 'constant', 'nearly'):
        bad = p0.copy()
        if spoil == 'dependent':
            bad[:, 3] = 2.0 * bad[:, 1] - bad[:, 2]
        elif spoil == 'constant':
            bad[:, 4] = 0.25                 # (a value whose mean over W walkers is exact: the centred column is exactly zero)
        else:
            bad[:, 3] = bad[:, 1] * (1.0 + 1e-11 * np.random.RandomState(1).randn(W))      # cond ~ 1e11: the moments cannot say, the singular values do
        assert not walkers_independent(bad)
        with pytest.raises(ValueError, match='condition number'):
            s.run_mcmc(bad, 2)
    s.run_mcmc(p0, 2)
    ctx.close()


def test_nan_in_the_initial_log_probability_raises_before_the_run():
    """emcee raises 'Probability function returned NaN' for the initial state before it samples.  The
    device sampler computes and checks the initial log-probabilities on the device without a host
    round trip; the flag's copy to the host starts at once and is looked at as soon as the first
    chunk is enqueued -- not after millions of iterations -- and the host's random stream is put back."""
    import time
    from bisip_amd.sampler import DeviceEnsembleSampler
    g = np.load(_case('case15_'))
    ctx = make_ctx(g, 'PeltonColeCole')
    p0 = _start(g, 32, 1)
    np.random.seed(4)
    s = DeviceEnsembleSampler(32, 4, ctx, persistent=False)
    s.run_mcmc(p0, 5)                                        # kernels loaded, allocator warm
    real = s.backend.logprob

    def poisoned(coords, out):
        real(coords, out)
        out[3] = float('nan')
    s.backend.logprob = poisoned
    state = s._random.get_state()
    t0 = time.perf_counter()
    with pytest.raises(ValueError, match='returned NaN'):
        s.run_mcmc(p0, 400000)                               # ~4 s of half-steps if it ran to the end
    assert time.perf_counter() - t0 < 1.5
    after = s._random.get_state()
    assert state[2] == after[2] and np.array_equal(state[1], after[1])
    s.backend.logprob = real
    s.run_mcmc(p0, 5)                                        # and the sampler is usable again
    assert np.isfinite(s.get_log_prob()).all()
    ctx.close()


def test_reset_forgets_the_acceptance_counts_on_the_device_too():
    """run, reset(), run(None, n) -- which emcee allows: the acceptance fraction is that of the second
    run alone (the device counter restarts with it), never above 1."""
    from bisip_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
    import oracle
    g = np.load(_case('case15_'))
    ctx = make_ctx(g, 'PeltonColeCole')
    p0 = _start(g, 32, 2)
    for kw in (dict(persistent=False), dict(persistent=True)):
        np.random.seed(9)
        s = DeviceEnsembleSampler(32, 4, ctx, **kw)
        s.run_mcmc(p0, 300)
        s.reset()
        s.run_mcmc(None, 40)
        np.random.seed(9)
        h = EnsembleSampler(32, 4, ctx.logprob)
        h.run_mcmc(p0, 300)
        h.reset()
        h.run_mcmc(None, 40)
        assert s.get_chain().shape == (40, 32, 4)
        assert np.array_equal(s.get_chain(), h.get_chain())
        assert np.array_equal(s.acceptance_fraction, h.acceptance_fraction)
        assert 0.0 <= s.acceptance_fraction.min() and s.acceptance_fraction.max() <= 1.0
    ctx.close()


@pytest.mark.parametrize('W,world', [(33, 2), (33, 3), (10, 8), (10, 64), (64, 5), (4096, 8), (4097, 7)])
def test_sharded_c_loop_with_every_rank_simulated_on_one_gpu(W, world):
    """The C half-step loop of the sharded sampler (bisip_stretch_run_sharded_dev: eval -> all-gather ->
    apply) has only ever met ONE rank on hardware.  Its multi-rank arithmetic -- which slots a rank owns,
    the pad of its slab, where the slab lies in the gather buffer, odd ensembles whose two halves differ,
    more ranks than slots (empty shards) -- is the same code with every rank's block evaluated on this
    device in turn and the all-gather left out (bisip_stretch_run_sharded_sim_dev): the chain must be the
    fused single-GPU chain, bit for bit, in both random-stream modes."""
    from bisip_amd.sampler import DeviceEnsembleSampler
    g = np.load(_case('case15_'))
    ctx = make_ctx(g, 'PeltonColeCole')
    p0 = _start(g, W, 3)
    for rng in ('numpy', 'philox'):
        runs = {}
        for name, kw in (('fused', dict(persistent=False)), ('simulated', dict(sharded_loop=f'simulate:{world}', persistent=False))):
            np.random.seed(21)
            s = DeviceEnsembleSampler(W, 4, ctx, rng=rng, seed=5, chunk=7, **kw)
            s.run_mcmc(p0, 15, thin_by=1)
            runs[name] = (s.get_chain(), s.get_log_prob(), s.acceptance_fraction, s.last_path)
        assert runs['fused'][3] == 'launch-per-half-step' and runs['simulated'][3] == f'sharded-simulated-{world}'
        for a, b in zip(runs['fused'][:3], runs['simulated'][:3]):
            assert np.array_equal(a, b), (rng, W, world)
    ctx.close()


def test_philox_draw_kernel_matches_contract():
    """bisip_stretch_draw_dev against the NumPy statement of the philox contract."""
    import torch
    from bisip_amd.sampler import affine_splits
    from numpy_stretch_backend import philox_stream
    g = np.load(_case('case16_'))
    ctx = make_ctx(g, 'PeltonColeCole')
    # 32 / 33: the flat launch shape of small ensembles; 4096 and 65536: 32-bit modular products;
    # 65537 and 70001: past 2^16 walkers the products need 64 bits (reduced through a quotient estimated in double,
    # perm_inverse); a million and three million walkers: products up to 2^43
    for W, n, step0, seed in [(32, 9, 0, 1), (33, 5, 1000, 0xdeadbeefcafe), (4096, 3, 7, 42), (65536, 2, 3, 9),
                              (65537, 2, 11, 5), (70001, 2, 0, 77), (1048576, 1, 5, 3), (3000001, 1, 2, 8)]:
        perm = affine_splits(seed, W, step0, n)
        nh = (W + 1) // 2
        dperm = torch.from_numpy(perm).cuda()
        bufs = {k: torch.empty((n, 2, nh), dtype=dt, device='cuda')
                for k, dt in (('active', torch.int32), ('partner', torch.int32), ('zz', torch.float64),
                              ('factor', torch.float64), ('logu', torch.float64))}
        ctx.stretch_draw_dev(W, 2.0, seed, step0, n, dperm.data_ptr(), bufs['active'].data_ptr(),
                             bufs['partner'].data_ptr(), bufs['zz'].data_ptr(),
                             bufs['factor'].data_ptr(), bufs['logu'].data_ptr(),
                             torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want = philox_stream(W, 7, 2.0, seed, step0, perm)
        assert np.array_equal(bufs['active'].cpu().numpy(), want['active'])
        assert np.array_equal(bufs['partner'].cpu().numpy(), want['partner'])
        assert np.array_equal(bufs['zz'].cpu().numpy(), want['zz'])           # pure arithmetic
        np.testing.assert_allclose(bufs['factor'].cpu().numpy(), want['factor'], rtol=1e-15, atol=1e-15)
        np.testing.assert_allclose(bufs['logu'].cpu().numpy(), want['logu'], rtol=1e-15, atol=1e-15)
        # every step is a balanced partition and partners come from the other half
        act = want['active']
        for k in range(n if W < 1000000 else 0):
            halves = [set(act[k, 0, :nh]), set(act[k, 1, :W // 2])]
            assert halves[0] | halves[1] == set(range(W)) and not (halves[0] & halves[1])
            assert set(want['partner'][k, 0, :nh]) <= halves[1]
            assert set(want['partner'][k, 1, :W // 2]) <= halves[0]
        z = want['zz'][:, 0, :]
        assert z.min() >= 0.5 and z.max() <= 2.0
    ctx.close()


def test_philox_mode_chain_replay_and_posterior():
    from bisip_amd.sampler import DeviceEnsembleSampler
    from numpy_stretch_backend import NumpyStretchBackend
    g = np.load(_case('case16_'))
    ctx = make_ctx(g, 'PeltonColeCole')
    W, ndim = 64, 7
    p0 = _start(g, W, 2)
    np.random.seed(5)
    dev = DeviceEnsembleSampler(W, ndim, ctx, rng='philox', seed=2024, chunk=64)
    dev.run_mcmc(p0, 150)
    # (a) replay: same contract evaluated in NumPy around the GPU log-probability
    np.random.seed(5)
    rep = DeviceEnsembleSampler(W, ndim, backend=NumpyStretchBackend(ctx.logprob), rng='philox',
                                seed=2024, chunk=50)
    rep.run_mcmc(p0, 150)
    assert np.array_equal(dev.get_chain(), rep.get_chain())
    assert np.array_equal(dev.acceptance_fraction, rep.acceptance_fraction)
    # (b) chunking and seeds: counter-based stream does not depend on the chunk size
    np.random.seed(5)
    dev2 = DeviceEnsembleSampler(W, ndim, ctx, rng='philox', seed=2024, chunk=17)
    dev2.run_mcmc(p0, 150)
    assert np.array_equal(dev.get_chain(), dev2.get_chain())
    np.random.seed(5)
    dev3 = DeviceEnsembleSampler(W, ndim, ctx, rng='philox', seed=2025)
    dev3.run_mcmc(p0, 150)
    assert not np.array_equal(dev.get_chain(), dev3.get_chain())
    # (c) posterior agreement between the two RNG modes (long runs, same target)
    np.random.seed(6)
    a = DeviceEnsembleSampler(W, ndim, ctx, rng='philox', seed=1)
    a.run_mcmc(p0, 3000)
    np.random.seed(7)
    b = DeviceEnsembleSampler(W, ndim, ctx, rng='numpy')
    b.run_mcmc(p0, 3000)
    fa, fb = a.get_chain(discard=1000, flat=True), b.get_chain(discard=1000, flat=True)
    sd = fb.std(axis=0)
    assert np.all(np.abs(fa.mean(0) - fb.mean(0)) < 0.25 * sd)
    assert np.all(np.abs(fa.std(0) / sd - 1) < 0.25)
    ctx.close()


@pytest.mark.parametrize('prefix,model,variant', CASES)
def test_persistent_kernel_equals_launch_per_half_step(prefix, model, variant):
    """The one-launch-per-chunk persistent kernel (workgroup per ensemble, state in LDS, 4 / 2 / 1
    lanes per walker depending on the ensemble size) reproduces the launch-per-half-step path
    bit for bit, for both random streams."""
    from bisip_amd.sampler import DeviceEnsembleSampler
    g = np.load(_case(prefix))
    ctx = make_ctx(g, model, variant)
    ndim = g['bounds'].shape[1]
    for W, nsteps, chunk, thin, rng in [(32, 80, None, 1, 'philox'), (33, 30, 7, 1, 'numpy'),
                                        (64, 24, 5, 3, 'numpy'), (510, 6, None, 1, 'philox'),
                                        (700, 4, None, 2, 'philox')]:
        if W < 2 * ndim or W * (ndim + 1) * 8 > 65536:
            continue
        p0 = _start(g, W, 300 + W)
        out = []
        for persistent in (True, False):
            np.random.seed(17)
            s = DeviceEnsembleSampler(W, ndim, ctx, rng=rng, seed=99, chunk=chunk, persistent=persistent)
            s.run_mcmc(p0, nsteps, thin_by=thin)
            s.run_mcmc(None, 4, thin_by=thin)       # continuation keeps the counters aligned
            out.append(s)
        a, b = out
        assert a.last_path == 'persistent' and b.last_path == 'launch-per-half-step'
        assert np.array_equal(a.get_chain(), b.get_chain())
        assert np.array_equal(a.get_log_prob(), b.get_log_prob())
        assert np.array_equal(a.acceptance_fraction, b.acceptance_fraction)
    ctx.close()


@pytest.mark.parametrize('prefix,model', [('case15_', 'PeltonColeCole'), ('case17_', 'PeltonColeCole'),
                                          ('case18_', 'Dias2000'), ('case19_', 'Shin2015')])
def test_lanes_per_slot_and_record_staging_do_not_change_the_chain(prefix, model, monkeypatch):
    """The tuning knobs of the sampler kernels -- lanes per slot (1 / 2 / 4), records read through
    the scalar cache or staged in LDS with the pipelined two-frequency loop -- select different
    instruction streams for the same arithmetic: every combination must give the same chain,
    persistent kernel and launch path alike."""
    from bisip_amd.sampler import DeviceEnsembleSampler
    g = np.load(_case(prefix))
    ctx = make_ctx(g, model)
    ndim = g['bounds'].shape[1]
    W = 96
    p0 = _start(g, W, 41)

    def run(persistent):
        s = DeviceEnsembleSampler(W, ndim, ctx, rng='philox', seed=5, persistent=persistent)
        s.run_mcmc(p0, 25)
        return s.get_chain(), s.get_log_prob()

    ref = run(False)
    for lanes in ('1', '2', '4'):
        monkeypatch.setenv('BISIP_STRETCH_LANES', lanes)
        for staging_off in (False, True):
            if staging_off:
                monkeypatch.setenv('BISIP_NO_LDS_STAGING', '1')
            else:
                monkeypatch.delenv('BISIP_NO_LDS_STAGING', raising=False)
            for persistent in (True, False):
                chain, logp = run(persistent)
                assert np.array_equal(chain, ref[0]), (lanes, staging_off, persistent)
                assert np.array_equal(logp, ref[1]), (lanes, staging_off, persistent)
    ctx.close()


@pytest.mark.parametrize('model,kw,W', [('PolynomialDecomposition', dict(poly_deg=5), 262144),
                                        ('PolynomialDecomposition', dict(poly_deg=5, variant='collapsed'), 131073),
                                        ('PolynomialDecomposition', dict(poly_deg=0), 131072),
                                        ('PolynomialDecomposition', dict(poly_deg=10), 140001),
                                        ('PeltonColeCole', dict(n_modes=1), 262144),
                                        ('PeltonColeCole', dict(n_modes=2), 262145),
                                        ('PeltonColeCole', dict(n_modes=3), 131072),
                                        ('Dias2000', {}, 150000), ('Shin2015', {}, 131074)])
def test_big_ensemble_samples_on_a_packed_state_and_keeps_the_chain(model, kw, W, monkeypatch, tmp_path):
    """A single ensemble of 131,072 walkers and more (one lane per slot) samples a chunk on a packed state --
    one aligned 64-byte row per walker: theta, padding, the log-probability (bisip_stretch_run_dev packs before the
    chunk's first half-step and unpacks after its last; ndim <= 7) -- so that a walker and its log-probability are one
    line to read and one to write.  Only the layout differs: the chain, the log-probabilities, the acceptance counts
    and the final state equal those of the plain layout (BISIP_NO_PACKED_STATE=1) bit for bit -- rows of 2 ... 7
    doubles, and 10 / 12 (no packed form: the plain kernel both times), even and odd ensembles (a last workgroup
    with dead lanes, halves of different sizes), thinning (iterations with and without a chain row), chunked
    runs (packed and unpacked again at every chunk).
    And the Philox stream of such a chunk is DRAWN IN PLACE by the half-step launches (bisip_stretch_run_philox_dev: no
    stream arrays, no draw kernel) -- the same entries from the same counters: the same chain as with the arrays
    (BISIP_NO_INLINE_DRAW=1), over several chunks (the counters carry on where the last chunk stopped) and through a
    second run_mcmc call."""
    import bisip_amd
    from bisip_amd.sampler import DeviceEnsembleSampler
    from bisip_amd.synthetic import write_spectrum_file
    path = write_spectrum_file(str(tmp_path / 's.csv'), 32, 1)
    m = getattr(bisip_amd, model)(path, nwalkers=W, nsteps=4, **kw)
    ctx = m._context()
    lo, hi = m.param_bounds
    ndim = lo.size
    centre = {'PolynomialDecomposition': np.r_[1.0, 0.004, np.zeros(ndim - 2)] if ndim > 1 else np.r_[1.0],
              'PeltonColeCole': np.r_[1.0, np.full((ndim - 1) // 3, 0.3), np.full((ndim - 1) // 3, -5.0), np.full((ndim - 1) // 3, 0.5)],
              'Dias2000': np.r_[1.0, 0.25, -10.0, 5.0, 0.5], 'Shin2015': np.r_[0.5, 0.5, -14.0, -6.0, 0.3, 0.45]}[model]
    p0 = centre + 1e-3 * (hi - lo) * np.random.RandomState(2).randn(W, ndim)
    p0 = np.clip(p0, lo + 1e-9 * (hi - lo), hi - 1e-9 * (hi - lo))

    packs = ndim <= 7

    def run(stream):
        s = DeviceEnsembleSampler(W, ndim, ctx, rng='philox', seed=9, persistent=False, live_dangerously=True, chunk=4)
        s.run_mcmc(p0, 3, thin_by=2)
        assert s.last_path == 'launch-per-half-step' and s.last_stream == stream
        s.run_mcmc(None, 1, thin_by=3)
        assert s.last_stream == stream
        return s.get_chain(), s.get_log_prob(), s.acceptance_fraction, s._coords
    monkeypatch.setenv('BISIP_NO_PACKED_STATE', '1')
    ref = run('arrays')
    monkeypatch.delenv('BISIP_NO_PACKED_STATE')
    monkeypatch.setenv('BISIP_NO_INLINE_DRAW', '1')
    packed = run('arrays')
    monkeypatch.delenv('BISIP_NO_INLINE_DRAW')
    got = run('in place' if packs else 'arrays')
    assert 0.05 < ref[2].mean() < 0.95
    for x, y, z in zip(got, ref, packed):
        assert np.array_equal(x, y) and np.array_equal(z, y)


def test_run_philox_entry_refuses_what_it_does_not_serve(monkeypatch):
    """bisip_stretch_run_philox_dev draws in place only where the packed-state half-step runs: anything else is
    refused BEFORE a launch (the caller then draws the arrays and takes bisip_stretch_run_dev), and so are bad
    arguments; bisip_stretch_philox_inline answers the same question without side effects."""
    import torch
    from bisip_amd import _hip
    from bisip_amd.sampler import affine_splits
    g = np.load(_case('case16_'))
    ctx = make_ctx(g, 'PeltonColeCole')
    ndim = g['bounds'].shape[1]
    assert ctx.stretch_philox_inline(262144) and not ctx.stretch_philox_inline(65536) and not ctx.stretch_philox_inline(1)
    monkeypatch.setenv('BISIP_NO_PACKED_STATE', '1')
    assert not ctx.stretch_philox_inline(262144)
    monkeypatch.delenv('BISIP_NO_PACKED_STATE')

    def args(W):
        a = _hip.StretchArgs()
        st = dict(coords=torch.zeros((W, ndim), dtype=torch.float64, device='cuda'), logp=torch.zeros(W, dtype=torch.float64, device='cuda'),
                  status=torch.zeros(1, dtype=torch.int32, device='cuda'))
        a.coords, a.logp, a.status = st['coords'].data_ptr(), st['logp'].data_ptr(), st['status'].data_ptr()
        a.n_slots = (W + 1) // 2
        return a, st
    W = 262144
    perm = torch.from_numpy(affine_splits(3, W, 0, 2)).cuda()
    a, keep = args(W)
    with pytest.raises(ValueError):
        ctx.stretch_run_philox_dev(a, W, 2, 1, 2.0, 3, 0, 0)                     # no splits
    with pytest.raises(ValueError):
        ctx.stretch_run_philox_dev(a, W, 3, 2, 2.0, 3, 0, perm.data_ptr())       # n_steps not a multiple of thin_by
    with pytest.raises(ValueError):
        ctx.stretch_run_philox_dev(a, W, 2, 1, 0.0, 3, 0, perm.data_ptr())       # a <= 0
    with pytest.raises(ValueError):
        ctx.stretch_run_philox_dev(a, W, 2, 1, 2.0, 3, (1 << 32) - 1, perm.data_ptr())   # the step counter has 32 bits
    small, keep2 = args(4096)
    with pytest.raises(RuntimeError, match='packed-state'):
        ctx.stretch_run_philox_dev(small, 4096, 2, 1, 2.0, 3, 0, perm.data_ptr())       # an ensemble the packed kernel does not take
    ctx.stretch_run_philox_dev(a, W, 0, 1, 2.0, 3, 0, perm.data_ptr())           # nothing to do: fine
    torch.cuda.synchronize()
    assert int(keep['status'].item()) == 0
    ctx.close()


def test_in_place_stream_of_a_chunk_whose_rows_are_selected_in_two_parts(monkeypatch, tmp_path):
    """A big chunk of a guarded run is enqueued in two parts (the guard's selection over the first 7/8 of its samples
    runs beside the last eighth): each part's launches draw from the counters of ITS iterations -- thinned, so that a
    part starts at a multiple of thin_by -- and the chain is that of the stream arrays."""
    import bisip_amd
    from bisip_amd.sampler import DeviceEnsembleSampler
    from bisip_amd.synthetic import write_spectrum_file
    path = write_spectrum_file(str(tmp_path / 's.csv'), 32, 2)
    W = 131072
    m = bisip_amd.PolynomialDecomposition(path, nwalkers=W, nsteps=16, poly_deg=4)
    lo, hi = m.param_bounds
    ctx = m._context()
    p0 = np.r_[1.0, 0.004, np.zeros(4)] + 1e-3 * (hi - lo) * np.random.RandomState(6).randn(W, 6)

    def run(stream):
        s = DeviceEnsembleSampler(W, 6, ctx, rng='philox', seed=33, chunk=32, live_dangerously=True)
        s.run_mcmc(p0, 16, thin_by=2)              # one chunk of 16 stored samples: parts of 14 and 2
        assert s.last_stream == stream and s.guard_['checks'] >= 2 and s.guard_['reruns'] == 0
        return s.get_chain(), s.get_log_prob(), s.acceptance_fraction
    monkeypatch.setenv('BISIP_NO_INLINE_DRAW', '1')
    ref = run('arrays')
    monkeypatch.delenv('BISIP_NO_INLINE_DRAW')
    got = run('in place')
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)


def test_chunk_sent_back_by_the_guard_draws_its_stream_in_place_again(monkeypatch, tmp_path):
    """A big ensemble draws its Philox stream inside the half-step launches, from counters: a chunk the guard sends
    back (the tier's own rows failed: the context has moved to the compensated kernel) starts again from its saved
    state AND from its first counter.  The measurement of the second chunk's rows is made to fail once (the move to
    the next tier done by hand, as bisip_ctx_reduced_guard_rows does it): chunks 0 stay the plain kernel's, chunks 1
    and 2 become the compensated kernel's -- the chain of a sampler that is switched by hand after chunk 0."""
    import bisip_amd
    from bisip_amd.sampler import DeviceEnsembleSampler
    from bisip_amd.synthetic import write_spectrum_file
    path = write_spectrum_file(str(tmp_path / 's.csv'), 32, 3)
    W, n = 131072, 6
    m = bisip_amd.PolynomialDecomposition(path, nwalkers=W, nsteps=n, poly_deg=5)
    lo, hi = m.param_bounds
    p0 = np.r_[1.0, 0.004, np.zeros(5)] + 1e-3 * (hi - lo) * np.random.RandomState(4).randn(W, 7)

    def fresh():
        ctx = m._context()
        ctx.set_bounds(m.param_bounds + 1e-9)
        ctx.set_bounds(m.param_bounds)
        ctx.set_variant('auto')
        assert ctx.kernel_name == 'k_logprob_pd_reduced'
        return ctx
    ctx = fresh()
    s = DeviceEnsembleSampler(W, 7, ctx, rng='philox', seed=21, chunk=2, live_dangerously=True)
    real = ctx.reduced_guard_rows
    calls = []

    def failing_once(theta, lp):
        calls.append(len(theta))
        if len(calls) == 3:                  # initial rows + chunk 0 pass; chunk 1's rows "fail"
            ctx.set_variant('reduced_comp')
            return 1e-9, True
        return real(theta, lp)
    monkeypatch.setattr(ctx, 'reduced_guard_rows', failing_once)
    s.run_mcmc(p0, n)
    assert s.last_stream == 'in place' and s.guard_['reruns'] == 1 and s.guard_['escalations'] == 1
    got = s.get_chain(), s.get_log_prob(), s.acceptance_fraction
    monkeypatch.undo()
    ctx = fresh()
    ref = DeviceEnsembleSampler(W, 7, ctx, rng='philox', seed=21, chunk=2, live_dangerously=True)
    ref.run_mcmc(p0, 2)
    ctx.set_variant('reduced_comp')
    ref.backend.logprob(ref._dev['coords'], ref._dev['logp'])       # as the guard does for the state it puts back
    ref.run_mcmc(None, n - 2)
    want = ref.get_chain(), ref.get_log_prob(), ref.acceptance_fraction
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('poly_deg', [2, 5, 7, 9, 10])
def test_compensated_persistent_kernel_with_and_without_staged_low_words(poly_deg, monkeypatch):
    """The compensated tier in the persistent kernel keeps its triangle in scalar registers and reads its low
    words (Rlo, elo) from LDS, where lane 0 put them (single spectrum) or the operand image was copied (a
    batch).  Same arithmetic as the launch path and as the bulk kernel: the chains are bit-equal with the
    staging on, with it off (BISIP_NO_LDS_STAGING=1: low words in registers) and from one launch per
    half-step; the stored log-probabilities are the bulk kernel's; ensembles of 1, 2 and 4 waves."""
    import bisip_amd
    from bisip_amd import _hip
    from bisip_amd.sampler import DeviceEnsembleSampler
    from bisip_amd.synthetic import synthetic_columns
    from bisip_amd.utils import columns_to_data
    d = columns_to_data(synthetic_columns(32, 3), 'mrad')
    per = np.log10(1. / d['w'])
    lt = np.linspace(np.floor(per.min() - 1), np.floor(per.max() + 1), 64)
    ndim = poly_deg + 2
    bounds = np.array([[0.9] + [-1.0] * (poly_deg + 1), [1.1] + [1.0] * (poly_deg + 1)])
    ctx = _hip.HipContext(0, d['w'], d['zn'], d['zn_err'], bounds, poly_deg=poly_deg, c_exp=1.0, taus=10 ** lt,
                          log_taus=np.array([lt ** i for i in range(poly_deg + 1)]), variant='reduced_comp')
    assert ctx.kernel_name == 'k_logprob_pd_reduced_comp'
    rng = np.random.RandomState(poly_deg)
    centre = np.r_[1.0, 0.005 * rng.randn(poly_deg + 1) / (1.0 + np.arange(poly_deg + 1)) ** 2]
    for W in (32, 200, 512):
        p0 = centre + 1e-4 * rng.randn(W, ndim)
        chains = {}
        for key, persistent, staging in (('staged', True, True), ('registers', True, False), ('launches', False, True)):
            if staging:
                monkeypatch.delenv('BISIP_NO_LDS_STAGING', raising=False)
            else:
                monkeypatch.setenv('BISIP_NO_LDS_STAGING', '1')
            s = DeviceEnsembleSampler(W, ndim, ctx, rng='philox', seed=7, persistent=persistent)
            s.run_mcmc(p0, 12, thin_by=2)
            chains[key] = (s.get_chain(), s.get_log_prob(), s.last_path)
        monkeypatch.delenv('BISIP_NO_LDS_STAGING', raising=False)
        assert chains['staged'][2] == 'persistent' and chains['registers'][2] == 'persistent'
        assert chains['launches'][2] == 'launch-per-half-step'
        for key in ('registers', 'launches'):
            assert np.array_equal(chains['staged'][0], chains[key][0]), (poly_deg, W, key)
            assert np.array_equal(chains['staged'][1], chains[key][1]), (poly_deg, W, key)
        last = chains['staged'][0][-1]
        assert np.array_equal(ctx.logprob(np.tile(last, (40, 1)))[:W], chains['staged'][1][-1])
        assert np.isfinite(chains['staged'][1]).all()
    ctx.close()
    # a batch: the image of every spectrum staged by its workgroup
    tables = [synthetic_columns(32, i) for i in range(5)]
    batch = bisip_amd.SpectraBatch('PolynomialDecomposition', tables, nwalkers=64, nsteps=6, poly_deg=poly_deg)
    batch.ctx.set_variant('reduced_comp')
    p0 = (centre + 1e-4 * rng.randn(5, 64, ndim)).reshape(-1, ndim)
    chains = {}
    for key, persistent, staging in (('staged', True, True), ('registers', True, False), ('launches', False, True)):
        if staging:
            monkeypatch.delenv('BISIP_NO_LDS_STAGING', raising=False)
        else:
            monkeypatch.setenv('BISIP_NO_LDS_STAGING', '1')
        s = DeviceEnsembleSampler(64, ndim, batch.ctx, rng='philox', seed=8, n_ensembles=5, persistent=persistent)
        s.run_mcmc(p0, 6)
        chains[key] = (s.get_chain(), s.get_log_prob())
    monkeypatch.delenv('BISIP_NO_LDS_STAGING', raising=False)
    for key in ('registers', 'launches'):
        assert np.array_equal(chains['staged'][0], chains[key][0]) and np.array_equal(chains['staged'][1], chains[key][1])
    batch.close()


def test_persistent_falls_back_when_ensemble_too_large():
    """Beyond one workgroup a single ensemble of up to 8,192 walkers and 7 parameters has the multi-workgroup
    persistent kernel; past either limit persistent=True quietly runs a launch per half-step."""
    from bisip_amd.sampler import DeviceEnsembleSampler
    g = np.load(_case('case15_'))
    ctx = make_ctx(g, 'PeltonColeCole')
    for W, want in ((4096, 'persistent-multi-workgroup'), (32770, 'launch-per-half-step')):
        p0 = _start(g, W, 9)
        np.random.seed(2)
        s = DeviceEnsembleSampler(W, 4, ctx, rng='philox', seed=5, persistent=True)
        s.run_mcmc(p0, 3)
        assert s.last_path == want
    ctx.close()
    assert ctx.group_walkers == 4096
    g = np.load(_case('case17_'))                # three modes: ndim 10, a row does not fit 64 bytes
    ctx = make_ctx(g, 'PeltonColeCole')
    s = DeviceEnsembleSampler(2048, 10, ctx, rng='philox', seed=5, persistent=True)
    s.run_mcmc(_start(g, 2048, 9), 3)
    assert s.last_path == 'launch-per-half-step'
    ctx.close()
    # left to itself the sampler takes the multi-workgroup kernel where it wins: the reduced PolynomialDecomposition
    # kernels (one lane per slot, one wave per workgroup, all XCDs beyond 32 workgroups) up to 32,768 walkers
    g = np.load(_case('case01_'))
    ctx = make_ctx(g, 'PolynomialDecomposition')
    ndim = g['bounds'].shape[1]
    if ndim <= 7:
        assert ctx.group_walkers == 32768
        for W, want in ((8192, 'persistent-multi-workgroup'), (32768, 'persistent-multi-workgroup'), (32770, 'launch-per-half-step')):
            s = DeviceEnsembleSampler(W, ndim, ctx, rng='philox', seed=5)
            s.run_mcmc(_start(g, W, 9), 3)
            assert s.last_path == want, (W, s.last_path)
    ctx.close()


@pytest.mark.parametrize('prefix,model,variant', [c for c in CASES if c[0] != 'case17_'] + [('case13_', 'PolynomialDecomposition', 'reduced_comp')])
def test_multi_workgroup_persistent_kernel_equals_launch_per_half_step(prefix, model, variant, monkeypatch):
    """One ensemble of 1,025 ... 8,192 walkers (BASELINE config 2: 4,096): k_stretch_group runs every iteration
    of a chunk in ONE launch -- several workgroups, the state in 64-byte rows in memory, a barrier of their own
    after every half-step -- and reproduces the launch-per-half-step path bit for bit: both random streams, odd
    ensembles (halves of different sizes, a last workgroup with dead slots), thinning, chunked runs,
    continuation, 1 / 2 / 4 / 8 lanes per slot (8: half a DPP row, this kernel's default up to 4,096 walkers), records
    staged in LDS or read through the scalar cache."""
    from bisip_amd.sampler import DeviceEnsembleSampler
    g = np.load(_case(prefix))
    ndim = g['bounds'].shape[1]
    if ndim > 7:
        pytest.skip('rows of more than 7 parameters do not fit the 64-byte state row')
    ctx = make_ctx(g, model, variant)
    for W, nsteps, chunk, thin, rng, lanes in [(2048, 12, None, 1, 'philox', None), (4096, 9, 4, 1, 'numpy', None),
                                               (4097, 6, None, 2, 'philox', '1'), (8192, 5, None, 1, 'philox', None),
                                               (1026, 8, 3, 1, 'philox', '2'), (3001, 6, None, 3, 'numpy', '4'),
                                               (1500, 7, None, 1, 'numpy', '8'), (4096, 5, 2, 2, 'philox', '8')]:
        p0 = _start(g, W, 500 + W)
        out = []
        for persistent in (True, False):
            if lanes is None:
                monkeypatch.delenv('BISIP_STRETCH_LANES', raising=False)
            else:
                monkeypatch.setenv('BISIP_STRETCH_LANES', lanes)
            if W == 1026 and persistent:
                monkeypatch.setenv('BISIP_NO_LDS_STAGING', '1')
            else:
                monkeypatch.delenv('BISIP_NO_LDS_STAGING', raising=False)
            np.random.seed(17)
            s = DeviceEnsembleSampler(W, ndim, ctx, rng=rng, seed=99, chunk=chunk, persistent=persistent)
            s.run_mcmc(p0, nsteps, thin_by=thin)
            s.run_mcmc(None, 3, thin_by=thin)       # continuation
            out.append(s)
        a, b = out
        assert a.last_path == 'persistent-multi-workgroup' and b.last_path == 'launch-per-half-step', (a.last_path, b.last_path)
        assert np.array_equal(a.get_chain(), b.get_chain()), W
        assert np.array_equal(a.get_log_prob(), b.get_log_prob()), W
        assert np.array_equal(a.acceptance_fraction, b.acceptance_fraction), W
        assert np.array_equal(a._coords, b._coords) and np.array_equal(a._lp, b._lp)
    ctx.close()


def test_samplers_on_two_threads_do_not_share_staging():
    """Two samplers run at the same time on two threads (own contexts, NumPy-order stream drawn on
    the host and staged in pinned memory, several chunks each so that the prefetch workers run
    too): each chain equals the chain of the same sampler run alone."""
    import threading
    from bisip_amd.sampler import DeviceEnsembleSampler
    g = np.load(_case('case15_'))

    def one(seed, W, nsteps, out):
        try:
            ctx = make_ctx(g, 'PeltonColeCole')
            p0 = _start(g, W, seed)
            s = DeviceEnsembleSampler(W, 4, ctx, rng='numpy', chunk=7)
            s._random.seed(seed)
            s.run_mcmc(p0, nsteps)
            out[seed] = (s.get_chain().copy(), s.get_log_prob().copy())
            ctx.close()
        except BaseException as exc:      # surfaced by the assertions below
            out[seed] = exc

    alone = {}
    one(3, 64, 60, alone)
    one(4, 256, 45, alone)
    for _ in range(3):
        both = {}
        threads = [threading.Thread(target=one, args=(3, 64, 60, both)),
                   threading.Thread(target=one, args=(4, 256, 45, both))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for seed in (3, 4):
            assert not isinstance(both[seed], BaseException), both[seed]
            assert np.array_equal(both[seed][0], alone[seed][0])
            assert np.array_equal(both[seed][1], alone[seed][1])


# ----------------------------------------------------------------------------------
# the device sampler's own guard of the QR-reduced kernels: no chunk of a failing tier is kept
# ----------------------------------------------------------------------------------

def _valley(prefix):
    import glob
    import os
    return np.load(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', prefix + '*.npz'))[0])


def _yardstick_rel(g, theta, lp):
    """|lp - exact| / max(1, |exact|) of stored log-probabilities against the library's binary128 yardstick
    (bisip_polydecomp_reduced_reference: 1e-13 from the 50-digit value on the valley fixtures)."""
    from bisip_amd import _hip
    exact = _hip.polydecomp_reduced_reference(g['w'], g['zn'], g['zn_err'], g['taus'], g['log_taus'], float(g['c_exp']),
                                              np.ascontiguousarray(theta))
    return np.abs(lp - exact) / np.maximum(1.0, np.abs(exact))


def test_device_fit_never_keeps_a_chunk_of_a_failing_tier(monkeypatch, tmp_path):
    """The default fit() samples PolynomialDecomposition with a QR-reduced kernel that AUTO chose from an
    ESTIMATE.  Here the estimate is made to pass (BISIP_AUTO_ERR_MAX, a test hook) on the worst-conditioned
    valley fixture's design -- degree 9, 64 frequencies, c = 0.5, where the plain triangle is 7e-9 off on the
    shell logp = 0 -- and the walkers start ON the fixture's valley / shell rows.  The device sampler measures
    rows of its own run against the binary128 yardstick before it keeps a chunk: the context ends on the
    compensated kernel, the chunk was run again from its saved initial state, EVERY stored log-probability is
    within 1e-10 of the exact value of the reference's formula, the chain is bit for bit the chain of a run
    that sampled with the compensated kernel from the start, and nothing warns."""
    import warnings
    import bisip_amd
    from bisip_amd.synthetic import write_spectrum_file
    g = _valley('valley06_')
    path = write_spectrum_file(str(tmp_path / 's.csv'), 64, 0)
    rows = g['theta'][np.all((g['bounds'][0] < g['theta']) & (g['theta'] < g['bounds'][1]), axis=1)]
    W = len(rows) - len(rows) % 2
    assert W >= 2 * rows.shape[1]
    p0 = np.ascontiguousarray(rows[:W])

    def model(variant):
        m = bisip_amd.PolynomialDecomposition(path, nwalkers=W, nsteps=240, poly_deg=9, c_exp=0.5, variant=variant)
        assert np.array_equal(m.data['w'], g['w']) and np.array_equal(m.data['zn'], g['zn'])     # the fixture's design
        return m
    m = model('auto')
    monkeypatch.setenv('BISIP_AUTO_ERR_MAX', '1e-6')
    ctx = m._context()
    monkeypatch.delenv('BISIP_AUTO_ERR_MAX')
    assert ctx.variant == 'reduced' and ctx.kernel_name == 'k_logprob_pd_reduced' and ctx.reduced_error > 1e-10
    # what the plain kernel would have stored: far outside the tolerance on the starting rows themselves
    ctx.reduced_guard(False)
    assert _yardstick_rel(g, p0, ctx.logprob(p0)).max() > 1e-10
    ctx.reduced_guard(True)
    for chunk in (None, 60):                      # one chunk; several (the guard's rows of chunk k are measured beside chunk k+1)
        ctx.set_bounds(m.param_bounds + 1e-9)     # a new box forgets what the guard found ...
        ctx.set_bounds(m.param_bounds)            # ... so each round starts on the plain tier again
        assert ctx.kernel_name == 'k_logprob_pd_reduced'
        np.random.seed(5)
        with warnings.catch_warnings():
            warnings.simplefilter('error', RuntimeWarning)
            if chunk is None:
                m.fit(p0=p0)
                s = m.sampler
            else:
                from bisip_amd.sampler import DeviceEnsembleSampler
                s = DeviceEnsembleSampler(W, p0.shape[1], ctx, chunk=chunk)
                s.run_mcmc(p0, 240)
        assert ctx.kernel_name == 'k_logprob_pd_reduced_comp' and ctx.variant == 'reduced_comp'
        assert s.guard_['escalations'] == 1 and s.guard_['reruns'] == 1 and s.guard_['rejected'] > 2e-11
        assert s.guard_['checks'] >= 2 and s.guard_['worst'] <= 2e-11
        chain, lp = s.get_chain(), s.get_log_prob()
        assert chain.shape == (240, W, p0.shape[1])
        rel = _yardstick_rel(g, chain.reshape(-1, chain.shape[-1]), lp.ravel())
        assert rel.max() <= 1e-10, rel.max()       # EVERY stored log-probability
        assert rel.max() <= 2e-11                  # (the compensated kernel's own bar on this design)
        ref = model('reduced_comp')
        np.random.seed(5)
        ref.fit(p0=p0)
        assert np.array_equal(chain, ref.get_chain()) and np.array_equal(lp, ref.sampler.get_log_prob())
        assert np.array_equal(s.acceptance_fraction, ref.sampler.acceptance_fraction)
    if chunk is None:
        assert m.reduced_check_ <= 2e-11


def test_guard_sends_back_the_chunk_in_which_the_walkers_reach_the_shell(monkeypatch):
    """A run that starts uniform in the prior box (|logp| ~ 1e5: every tier is accurate RELATIVELY) and crosses
    the shell logp = 0 some chunks later, on a degree-7 design whose plain triangle is ~1e-9 off there while
    its estimate is let through (BISIP_AUTO_ERR_MAX, a test hook).  The chunks before the crossing
    are the plain kernel's and stay; the chunk of the crossing is run again by the compensated kernel, from the
    state and the random stream it had started with; both random streams."""
    from bisip_amd.sampler import DeviceEnsembleSampler
    from test_gpu_parity import _pd_context
    n_freq, poly_deg, c_exp, idx = 48, 7, 0.5, 0
    monkeypatch.setenv('BISIP_AUTO_ERR_MAX', '1e-9')
    ctx, bounds, d, taus, log_taus = _pd_context(n_freq, poly_deg, c_exp, idx)
    monkeypatch.delenv('BISIP_AUTO_ERR_MAX')
    g = dict(w=d['w'], zn=d['zn'], zn_err=d['zn_err'], taus=taus, log_taus=log_taus, c_exp=c_exp)
    W, ndim, nsteps, chunk = 64, poly_deg + 2, 600, 50
    p0 = np.random.RandomState(3).uniform(bounds[0], bounds[1], (W, ndim))
    comp = _pd_context(n_freq, poly_deg, c_exp, idx, variant='reduced_comp')[0]
    plain = _pd_context(n_freq, poly_deg, c_exp, idx, variant='reduced')[0]
    for rng in ('numpy', 'philox'):
        ctx.set_bounds(bounds + 1e-9)
        ctx.set_bounds(bounds)
        assert ctx.variant == 'reduced'
        np.random.seed(11)
        s = DeviceEnsembleSampler(W, ndim, ctx, rng=rng, seed=7, chunk=chunk, persistent=False)
        s.run_mcmc(p0, nsteps)
        assert ctx.variant == 'reduced_comp' and s.guard_['escalations'] == 1 and s.guard_['reruns'] == 1
        chain, lp = s.get_chain(), s.get_log_prob()
        assert np.abs(lp).min() < 1.0                      # the walkers did cross the shell
        rel = _yardstick_rel(g, chain.reshape(-1, ndim), lp.ravel())
        assert rel.max() <= 2e-11, rel.max()
        # which chunk came back: its stored log-probabilities are the compensated kernel's, the chunks before it
        # are still the plain kernel's
        is_comp = np.array([np.array_equal(comp.logprob(chain[i]), lp[i]) for i in range(nsteps)])
        is_plain = np.array([np.array_equal(plain.logprob(chain[i]), lp[i]) for i in range(nsteps)])
        first = int(np.argmax(is_comp & ~is_plain))
        assert first % chunk == 0 and first > 0, first
        assert is_plain[:first].all() and is_comp[first:].all()
        # the same run with the guard switched off keeps the plain kernel and its error on the shell
        ctx.set_bounds(bounds + 1e-9)
        ctx.set_bounds(bounds)
        ctx.reduced_guard(False)
        np.random.seed(11)
        off = DeviceEnsembleSampler(W, ndim, ctx, rng=rng, seed=7, chunk=chunk, persistent=False)
        off.run_mcmc(p0, nsteps)
        ctx.reduced_guard(True)
        assert ctx.variant == 'reduced' and off.guard_['checks'] == 0
        assert np.array_equal(off.get_chain()[:first], chain[:first])          # identical until the chunk that came back
        assert _yardstick_rel(g, off.get_chain().reshape(-1, ndim), off.get_log_prob().ravel()).max() > 2e-11
    for c in (ctx, comp, plain):
        c.close()
