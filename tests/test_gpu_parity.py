"""GPU parity: the HIP kernels (through the C ABI) against the golden vectors the real
reference produced, and against the CPU oracle on fresh seeded inputs.

Tolerance: |dlogp| <= 1e-10*max(1,|logp|), |dZ| <= 1e-12*max(1,max|Z|);
-inf rows and row order must match exactly.
"""

import numpy as np
import pytest

from conftest import (assert_logp_close, assert_Z_close, case_id, case_model, extended_cases, golden_cases,
                      valley_cases)

pytestmark = pytest.mark.gpu

MODEL_IDS = {'PolynomialDecomposition': 0, 'PeltonColeCole': 1, 'Dias2000': 2, 'Shin2015': 3}


def make_ctx(g, model, variant='auto'):
    from bisip_amd import _hip
    kw = {}
    if model == 'PolynomialDecomposition':
        kw = dict(poly_deg=int(g['poly_deg']), c_exp=float(g['c_exp']), taus=g['taus'],
                  log_taus=g['log_taus'])
    if model == 'PeltonColeCole':
        kw = dict(n_modes=int(g['n_modes']))
    return _hip.HipContext(MODEL_IDS[model], g['w'], g['zn'], g['zn_err'], g['bounds'],
                           variant=variant, **kw)


def variants_for(g, model):
    if model != 'PolynomialDecomposition':
        return ['auto']
    v = ['auto', 'reduced', 'reduced_comp', 'collapsed']
    if int(g['poly_deg']) <= 7:
        v.append('faithful')
    if g['w'].size <= 64:
        v.append('wave')
    return v


@pytest.mark.parametrize('path', golden_cases(), ids=case_id)
def test_logprob_matches_reference_golden(path):
    g = np.load(path)
    model = case_model(path)
    for variant in variants_for(g, model):
        ctx = make_ctx(g, model, variant)
        got = ctx.logprob(g['theta'])
        err = assert_logp_close(got, g['logp'])
        print(f'{case_id(path)} [{ctx.kernel_name}] max rel err {err:.2e}')
        ctx.close()


@pytest.mark.parametrize('path', golden_cases(), ids=case_id)
def test_forward_matches_reference_golden(path):
    g = np.load(path)
    model = case_model(path)
    ctx = make_ctx(g, model)
    rows = np.all(np.isfinite(g['theta']), axis=1)
    got = ctx.forward(g['theta'][rows])
    assert_Z_close(got, g['Z'][rows])
    ctx.close()


@pytest.mark.parametrize('path', extended_cases(), ids=case_id)
def test_extended_shapes_match_reference_golden(path):
    """Reference outputs for shapes outside its tutorials (polynomial degree 0 / 8-10, exponents
    down to 0.21, 2N < P+2, 4-5 Cole-Cole modes) and forward() at every parameter's bounds."""
    g = np.load(path)
    model = case_model(path)
    for variant in variants_for(g, model):
        ctx = make_ctx(g, model, variant)
        err = assert_logp_close(ctx.logprob(g['theta']), g['logp'])
        print(f'{case_id(path)} [{ctx.kernel_name}] max rel err {err:.2e}')
        if variant == 'auto' and model == 'PolynomialDecomposition':
            # AUTO runs a QR-reduced kernel -- plain, or compensated on the nearly collinear
            # designs of degree 8-10 -- wherever the design has a triangle at all (2N >= P+2)
            square = 2 * g['w'].size >= int(g['poly_deg']) + 2
            assert ctx.kernel_name.startswith('k_logprob_pd_reduced') == square, ctx.kernel_name
            if square:
                assert ctx.reduced_error <= 1e-12
        ctx.close()
    ctx = make_ctx(g, model)
    rows = np.all(np.isfinite(g['theta']), axis=1)
    assert_Z_close(ctx.forward(g['theta'][rows]), g['Z'][rows])
    Zf = ctx.forward(g['theta_fwd_edges'])
    assert np.all(np.isfinite(Zf))
    assert_Z_close(Zf, g['Z_fwd_edges'])
    ctx.close()


# ----------------------------------------------------------------------------------
# fresh seeded inputs against the CPU oracle (sizes the oracle finishes in seconds)
# ----------------------------------------------------------------------------------

def _oracle_problem(g, model):
    import oracle
    return oracle.OracleProblem.from_golden(g, model)


@pytest.mark.parametrize('path', valley_cases(), ids=case_id)
def test_valley_rows_against_the_reference_and_the_exact_value(path):
    """Rows where the reference's own rounding exceeds the tolerance (fixtures: the real reference's
    value AND the exact value of its formula, 50 digits).  What BISIP_VARIANT_AUTO runs must be within
    1e-10 of the reference or of the exact value, row by row; the compensated kernel within 2e-11 of
    the exact value on every design -- also the one with terms 6e7 times the row sums (degree 9, 64
    frequencies, c = 0.5), where the reference itself is 8.5e-9 away: its operands come from a QR in
    binary128 (host_precompute.cpp: reduced_make_quad).  No per-design constant."""
    g = np.load(path)
    exact, ref = g['logp_exact'], g['logp']
    scale = np.maximum(1.0, np.abs(exact))
    for variant in ('auto', 'reduced_comp'):
        ctx = make_ctx(g, 'PolynomialDecomposition', variant)
        ctx.reduced_guard(False)
        got = ctx.logprob(g['theta'])
        to_exact, to_ref = np.abs(got - exact) / scale, np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
        print(f'{case_id(path)} [{ctx.kernel_name}]: {to_exact.max():.1e} from the exact value, {to_ref.max():.1e} from the '
              f'reference (which is {np.max(np.abs(ref - exact) / scale):.1e} from the exact value)')
        assert np.all(np.minimum(to_exact, to_ref) <= 1e-10)
        if variant == 'reduced_comp':
            assert to_exact.max() <= 2e-11
        assert to_exact.max() <= max(1e-10, np.max(np.abs(ref - exact) / scale) / 20)     # never worse than the reference
        # the library's own yardstick (what its estimates, checks and guard measure against) on the same rows
        mine = _yardstick(g)
        assert np.max(np.abs(mine - exact) / scale) <= 1e-12
        ctx.close()


@pytest.mark.parametrize('path', [p for p in valley_cases() if int(np.load(p)['poly_deg']) >= 6], ids=case_id)
def test_compensated_bulk_launch_with_operands_from_memory(path):
    """From degree 6 on, a bulk launch (>= 131072 rows) of a lone spectrum's compensated kernel reads its
    operands from memory through the scalar path, as a batch does (host.h:
    REDUCED_COMP_MEMORY_OPERANDS_FROM); below that size, and at lower degrees, they are kernel arguments.
    A row's value must not depend on the route: the fixture's valley rows inside a bulk batch, the bulk
    batch against itself in small launches, an 8-byte-aligned theta (the other staging path), and the
    exact values of the fixture."""
    import torch
    g = np.load(path)
    lo, hi = g['bounds']
    rng = np.random.RandomState(int(g['poly_deg']))
    W = 131072 + 257
    theta = rng.uniform(lo, hi, (W, lo.size))
    k = g['theta'].shape[0]
    theta[1000:1000 + k] = g['theta']
    ctx = make_ctx(g, 'PolynomialDecomposition', 'reduced_comp')
    ctx.reduced_guard(False)
    assert ctx.kernel_name == 'k_logprob_pd_reduced_comp'
    bulk = ctx.logprob(theta)
    pieces = np.concatenate([ctx.logprob(theta[i:i + 40000]) for i in range(0, W, 40000)])
    assert np.array_equal(bulk, pieces)
    exact = g['logp_exact']
    assert np.max(np.abs(bulk[1000:1000 + k] - exact) / np.maximum(1.0, np.abs(exact))) <= 2e-11
    dev = torch.from_numpy(np.concatenate([np.zeros((1, lo.size)), theta])).cuda()
    view = dev.reshape(-1)[1:1 + W * lo.size]                # starts 8 bytes in: the scalar staging path
    assert view.data_ptr() % 16 == 8
    view.copy_(torch.from_numpy(theta.reshape(-1)))
    out = torch.empty(W, dtype=torch.float64, device='cuda')
    ctx.logprob_dev(view.data_ptr(), W, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), bulk)
    # a new prior box: the device image follows the new expansion point
    wide = np.array([lo - 0.05 * (hi - lo), hi + 0.05 * (hi - lo)])
    ctx.set_bounds(wide)
    bulk2 = ctx.logprob(theta)
    pieces2 = np.concatenate([ctx.logprob(theta[i:i + 40000]) for i in range(0, W, 40000)])
    assert np.array_equal(bulk2, pieces2)
    assert np.max(np.abs(bulk2[1000:1000 + k] - exact) / np.maximum(1.0, np.abs(exact))) <= 2e-11
    ctx.close()


def _yardstick(g):
    from bisip_amd import _hip
    return _hip.polydecomp_reduced_reference(g['w'], g['zn'], g['zn_err'], g['taus'], g['log_taus'], float(g['c_exp']), g['theta'])


@pytest.mark.parametrize('path', golden_cases()[::2], ids=case_id)
def test_seeded_batch_against_oracle(path):
    import oracle
    g = np.load(path)
    model = case_model(path)
    rng = np.random.RandomState(4242)
    lo, hi = g['bounds']
    n = 3000
    theta = rng.uniform(lo, hi, (n, lo.size))
    # sprinkle out-of-prior rows: they must come back as -inf in place
    theta[::97, 0] = hi[0] + 1.0
    theta[5::211, -1] = lo[-1]
    want = oracle.logprob(_oracle_problem(g, model), theta, n_threads=4)
    for variant in variants_for(g, model):
        ctx = make_ctx(g, model, variant)
        assert_logp_close(ctx.logprob(theta), want)
        ctx.close()


@pytest.mark.parametrize('W', [0, 1, 2, 63, 64, 65, 255, 256, 257, 1000, 131071, 131072, 131073])
def test_ragged_batch_sizes(W):
    """Empty, sub-wave, wave/workgroup boundaries and the small/large launch switch
    (131072) -- each row's value must not depend on the batch it sits in."""
    import oracle
    path = [p for p in golden_cases() if 'PolynomialDecomposition_synthetic-N32-i0' in p][0]
    g = np.load(path)
    rng = np.random.RandomState(W)
    lo, hi = g['bounds']
    theta = rng.uniform(lo, hi, (W, lo.size))
    prob = _oracle_problem(g, 'PolynomialDecomposition')
    ref_rows = min(W, 512)
    want = oracle.logprob(prob, theta[:ref_rows])
    for variant in ('reduced', 'collapsed', 'faithful', 'wave'):
        ctx = make_ctx(g, 'PolynomialDecomposition', variant)
        got = ctx.logprob(theta)
        assert got.shape == (W,)
        assert_logp_close(got[:ref_rows], want)
        if W > 600:
            # batch-composition independence, bitwise: tail rows alone == tail rows in batch
            tail = ctx.logprob(theta[-300:])
            assert np.array_equal(tail, got[-300:])
        ctx.close()


@pytest.mark.parametrize('case,model', [('PolynomialDecomposition_synthetic-N32-i0', 'PolynomialDecomposition'),
                                        ('case16_', 'PeltonColeCole'), ('case18_', 'Dias2000')])
def test_big_host_buffer_calls_are_pipelined_and_keep_their_bits(case, model, monkeypatch):
    """bisip_logprob / bisip_forward on host buffers of 192 MB and more run as a pipeline over 64 MB chunks of rows
    (pinned double buffers, the rows up on one stream, the results back on another, the copies on the host
    threads; the device holds two chunks).  A row's value does not depend on the chunk it travels in: the results
    equal the single-launch path's (BISIP_NO_HOST_PIPELINE=1) bit for bit -- a size that ends inside a chunk, one
    just past the threshold; -inf rows; rows against the oracle."""
    import oracle
    path = [p for p in golden_cases() if case in p][0]
    g = np.load(path)
    lo, hi = g['bounds']
    ndim = lo.size
    prob = _oracle_problem(g, model)
    ctx = make_ctx(g, model)
    if model == 'PolynomialDecomposition':
        ctx.reduced_guard(False)
    row_bytes = 8 * (ndim + 1)
    for W in ((192 << 20) // row_bytes + 5, (64 << 20) // row_bytes // 256 * 256 * 4 + 777):
        theta = np.random.RandomState(W % 1000).uniform(lo, hi, (W, ndim))
        theta[::9973, 0] = hi[0] + 1.0                        # some rows outside the prior
        got = ctx.logprob(theta)
        monkeypatch.setenv('BISIP_NO_HOST_PIPELINE', '1')
        want = ctx.logprob(theta)
        monkeypatch.delenv('BISIP_NO_HOST_PIPELINE')
        assert np.array_equal(got, want) and np.isneginf(got[::9973]).all()
        pick = np.random.RandomState(1).choice(W, 300, replace=False)
        assert_logp_close(got[pick], oracle.logprob(prob, theta[pick]))
        del theta, got, want
    # forward: 16 N bytes come back per row
    N = g['w'].size
    Wf = (192 << 20) // (8 * ndim + 16 * N) + 33
    theta = np.random.RandomState(5).uniform(lo, hi, (Wf, ndim))
    Z = ctx.forward(theta)
    monkeypatch.setenv('BISIP_NO_HOST_PIPELINE', '1')
    Z1 = ctx.forward(theta)
    monkeypatch.delenv('BISIP_NO_HOST_PIPELINE')
    assert Z.shape == (Wf, 2, N) and np.array_equal(Z, Z1)
    assert_Z_close(Z[-50:], oracle.forward(prob, theta[-50:]))
    ctx.close()


def test_unaligned_device_pointer_path():
    """bisip_logprob_dev on a theta view that starts 8 B (not 16 B) aligned takes the
    scalar-load staging path; results must be identical."""
    import torch
    path = [p for p in golden_cases() if 'PolynomialDecomposition_synthetic-N32-i0' in p][0]
    g = np.load(path)
    ctx = make_ctx(g, 'PolynomialDecomposition', 'reduced')
    rng = np.random.RandomState(3)
    lo, hi = g['bounds']
    W = 5000
    theta = rng.uniform(lo, hi, (W + 1, lo.size))
    dev = torch.from_numpy(theta).cuda()
    out_a = torch.empty(W, dtype=torch.float64, device='cuda')
    out_b = torch.empty(W, dtype=torch.float64, device='cuda')
    view = dev[1:]                       # offset by 7 doubles = 56 B -> 8-B aligned only
    assert view.data_ptr() % 16 == 8
    st = torch.cuda.current_stream().cuda_stream
    ctx.logprob_dev(view.data_ptr(), W, out_a.data_ptr(), st)
    aligned = view.clone()
    assert aligned.data_ptr() % 16 == 0
    ctx.logprob_dev(aligned.data_ptr(), W, out_b.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b)
    assert_logp_close(out_a.cpu().numpy(), ctx.logprob(theta[1:]))
    ctx.close()


def test_full_size_properties():
    """BASELINE sizes (no oracle at this scale): permutation equivariance (walker index
    preserved bit-exactly), agreement of the three formulations, chunk invariance,
    -inf exactly where the prior says."""
    path = [p for p in golden_cases() if 'PolynomialDecomposition_synthetic-N64-i0' in p][0]
    g = np.load(path)   # cfg3 shape: N=64, S=128, P=5
    rng = np.random.RandomState(77)
    lo, hi = g['bounds']
    W = 65536
    theta = rng.uniform(lo, hi, (W, lo.size))
    bad = rng.rand(W) < 0.1
    theta[bad, 2] = 1.5
    ctx = make_ctx(g, 'PolynomialDecomposition', 'reduced')
    a = ctx.logprob(theta)
    assert np.array_equal(np.isneginf(a), bad)
    perm = rng.permutation(W)
    assert np.array_equal(ctx.logprob(theta[perm]), a[perm])
    assert np.array_equal(np.concatenate([ctx.logprob(theta[:30000]), ctx.logprob(theta[30000:])]), a)
    for variant in ('collapsed', 'faithful', 'wave'):
        ctx.set_variant(variant)
        assert_logp_close(ctx.logprob(theta), a, 1e-11)
    ctx.close()


def test_cfg2_shape_colecole_4096_walkers():
    """BASELINE config 2: single Cole-Cole, 32 synthetic frequencies, 4096 walkers."""
    import oracle
    path = [p for p in golden_cases() if 'case25_PeltonColeCole_synthetic-N32-i0' in p][0]
    g = np.load(path)
    from bisip_amd.synthetic import synthetic_theta
    theta = synthetic_theta(g['bounds'][0], g['bounds'][1], 4096)
    ctx = make_ctx(g, 'PeltonColeCole')
    want = oracle.logprob(_oracle_problem(g, 'PeltonColeCole'), theta, n_threads=4)
    assert_logp_close(ctx.logprob(theta), want)
    ctx.close()


def test_bounds_update_and_error_paths():
    from bisip_amd import _hip
    path = [p for p in golden_cases() if 'Dias2000_SIP-K389175' in p][0]
    g = np.load(path)
    ctx = make_ctx(g, 'Dias2000')
    th = g['theta'][:40].copy()
    base = ctx.logprob(th)
    assert np.isfinite(base).all()
    nb = g['bounds'].copy()
    nb[1, 3] = 25.0                                   # eta in [0, 25], as the tutorial does
    ctx.set_bounds(nb)
    got = ctx.logprob(th)
    outside = th[:, 3] >= 25.0
    assert outside.any() and (~outside).any()
    assert np.isneginf(got[outside]).all() and np.array_equal(got[~outside], base[~outside])
    with pytest.raises(ValueError):
        ctx.logprob(np.zeros((3, 4)))                 # wrong ndim
    with pytest.raises(RuntimeError):
        ctx.set_variant('reduced')                    # single-formulation model
    with pytest.raises(ValueError):
        _hip.HipContext(2, g['w'], g['zn'], g['zn_err'], g['bounds'][:, :4])
    ctx.close()


# ----------------------------------------------------------------------------------
# the drop-in class surface on the GPU
# ----------------------------------------------------------------------------------

def test_model_classes_drop_in():
    import bisip_amd
    path = bisip_amd.DataFiles()['SIP-K389175']
    g = np.load([p for p in golden_cases() if 'case04_' in p][0])
    m = bisip_amd.PolynomialDecomposition(path, nwalkers=32, nsteps=50)
    d = m.data
    args = (m.forward, m.param_bounds, d['w'], d['zn'], d['zn_err'])
    # emcee's per-walker call:  log_prob_fn(theta, *args) -> float
    i = int(g['n_prior'])
    one = m._log_probability(g['theta'][i], *args)
    assert isinstance(one, float)
    assert abs(one - g['logp'][i]) <= 1e-10 * max(1, abs(g['logp'][i]))
    # vectorised call: (n, ndim) -> (n,)
    assert_logp_close(m._log_probability(g['theta'], *args), g['logp'])
    assert_logp_close(m.log_prob(g['theta']), g['logp'])
    # forward: (ndim,) -> (2,N), (n,ndim) -> (n,2,N)
    rows = np.all(np.isfinite(g['theta']), axis=1)
    assert m.forward(g['theta'][i], d['w']).shape == (2, 20)
    assert_Z_close(m.forward(g['theta'][rows], d['w']), g['Z'][rows])
    # likelihood alone ignores the prior box
    outside = g['theta'][i].copy()
    outside[0] = 1.2
    assert np.isneginf(m._log_probability(outside, *args))
    assert np.isfinite(m._log_likelihood(outside, m.forward, d['w'], d['zn'], d['zn_err']))
    ll = m._log_likelihood(g['theta'][i], m.forward, d['w'], d['zn'], d['zn_err'])
    assert abs(ll - g['logp'][i]) <= 1e-10 * max(1, abs(g['logp'][i]))
    # forward on another frequency grid (denser, for plotting)
    w2 = np.logspace(5, -2, 50)
    assert m.forward(g['theta'][i], w2).shape == (2, 50)
    # the likelihood never touches the box of the context fit() / _log_probability use
    assert np.array_equal(m._context().logprob(g['theta']), m.log_prob(g['theta']))
    assert m._context().variant == 'reduced' and m._context(prior=False).variant == 'collapsed'
    # a model pinned to a reduced formulation: the likelihood alone has no box to centre it in
    mr = bisip_amd.PolynomialDecomposition(path, variant='reduced_comp')
    assert mr._context().variant == 'reduced_comp' and mr._context(prior=False).variant == 'collapsed'
    assert abs(mr._log_likelihood(outside, mr.forward, d['w'], d['zn'], d['zn_err'])
               - m._log_likelihood(outside, m.forward, d['w'], d['zn'], d['zn_err'])) <= 1e-9


def test_foreign_model_callable():
    """The reference's _log_likelihood / _log_probability take ANY callable f(theta, x) -> (2,N)
    (src/bisip/models.py:59-76).  A callable that is not the model's own forward runs on the
    host, row by row as in the reference, and the residual reduction on the device
    (bisip_loglike_z); checked against the reference's own formula in NumPy."""
    import bisip_amd
    path = bisip_amd.DataFiles()['SIP-K389175']
    m = bisip_amd.PeltonColeCole(path, nwalkers=32, nsteps=10, n_modes=1)
    d = m.data
    calls = []

    def debye(theta, w):      # a user model with the Pelton parameter vector, c forced to 1
        calls.append(1)
        r0, m1, lt, _ = theta
        z = r0 * (1 - m1 * (1 - 1 / (1 + 1j * w * np.exp(lt))))
        return np.array([z.real, z.imag])

    rng = np.random.RandomState(3)
    lo, hi = m.param_bounds
    theta = rng.uniform(lo, hi, (70, 4))
    theta[5, 1] = 1.5          # outside the prior
    theta[9, 0] = hi[0]        # on the bound

    def ref_ll(t):
        sigma2 = d['zn_err'] ** 2
        return -0.5 * np.sum((d['zn'] - debye(t, d['w'])) ** 2 / sigma2 + 2 * np.log(sigma2))

    want_ll = np.array([ref_ll(t) for t in theta])
    calls.clear()
    got = m._log_likelihood(theta, debye, d['w'], d['zn'], d['zn_err'])
    assert len(calls) == 70
    assert_logp_close(got, want_ll)
    one = m._log_likelihood(theta[0], debye, d['w'], d['zn'], d['zn_err'])
    assert isinstance(one, float) and abs(one - want_ll[0]) <= 1e-10 * max(1, abs(want_ll[0]))
    inside = np.logical_and(lo < theta, theta < hi).all(axis=1)
    want_lp = np.where(inside, want_ll, -np.inf)
    calls.clear()
    got_lp = m._log_probability(theta, debye, m.param_bounds, d['w'], d['zn'], d['zn_err'])
    assert len(calls) == inside.sum() == 68      # out-of-prior rows never reach f
    assert_logp_close(got_lp, want_lp)
    # the context entry point itself: shapes, empty batch, bad shape
    ctx = m._context(prior=False)
    assert ctx.loglike_z(np.empty((0, 2, 20))).shape == (0,)
    with pytest.raises(ValueError):
        ctx.loglike_z(np.zeros((3, 2, 19)))
    # with the model's own forward handed in as Z the two routes agree to rounding
    Z = m.forward(theta, d['w'])
    assert_logp_close(ctx.loglike_z(Z), m._log_likelihood(theta, m.forward, d['w'], d['zn'], d['zn_err']))


def test_fit_with_the_chain_kept_on_the_device():
    """fit(chain='device'): the stored samples stay in HBM; the reference's summary calls
    (get_param_mean / _std / _percentile, src/bisip/utils.py:37-85) are answered there and agree
    with NumPy on the chain that get_chain() copies out; the chain itself is the host-chain run's."""
    import bisip_amd
    path = bisip_amd.DataFiles()['SIP-K389175']
    runs = {}
    for chain in ('host', 'device'):
        m = bisip_amd.PeltonColeCole(path, nwalkers=300, nsteps=120, n_modes=1)
        np.random.seed(21)
        m.fit(rng='philox', chain=chain)
        runs[chain] = m
    dev, host = runs['device'], runs['host']
    assert dev.sampler.chain_on_device and not host.sampler.chain_on_device
    with pytest.warns(UserWarning):
        dev.get_param_mean()
    mean, std = dev.get_param_mean(discard=40, thin=2), dev.get_param_std(discard=40, thin=2)
    pct = dev.get_param_percentile([2.5, 50, 97.5], discard=40, thin=2)
    med = dev.get_param_percentile(50, discard=40)
    flat = dev.get_chain(discard=40, thin=2, flat=True)                 # copied out now
    assert np.array_equal(flat, host.get_chain(discard=40, thin=2, flat=True))
    assert np.allclose(mean, flat.mean(axis=0), rtol=1e-12, atol=1e-14)
    assert np.allclose(std, flat.std(axis=0), rtol=1e-9, atol=1e-14)
    assert np.allclose(pct, np.percentile(flat, [2.5, 50, 97.5], axis=0), rtol=1e-13, atol=1e-15)
    assert med.shape == (4,) and np.allclose(med, np.percentile(dev.get_chain(discard=40, flat=True), 50, axis=0), rtol=1e-13)
    # model-space percentiles: forward + column sort where the chain lies == the same kernels fed from the host
    mp = dev.get_model_percentile([2.5, 50, 97.5], discard=40, thin=2)
    assert mp.shape == (3, 2, 20) and np.array_equal(mp, host.get_model_percentile([2.5, 50, 97.5], chain=flat))
    assert np.allclose(mp, np.percentile(dev.forward(flat, dev.data['w']), [2.5, 50, 97.5], axis=0), rtol=1e-13, atol=1e-15)
    assert dev.get_model_percentile(50, discard=40, thin=2).shape == (2, 20)
    # a caller's own array still goes through NumPy
    assert np.array_equal(dev.get_param_mean(flat), flat.mean(axis=0))
    # get_chain() copied the samples out; they are still on the device for the summaries,
    # also after the run is continued
    assert np.array_equal(dev.get_param_mean(discard=40, thin=2), mean)
    dev.sampler.run_mcmc(None, 30)
    assert dev.get_chain().shape == (150, 300, 4) and np.array_equal(dev.get_chain()[:120], host.get_chain())
    assert np.allclose(dev.get_param_std(discard=100), dev.get_chain(discard=100, flat=True).std(axis=0), rtol=1e-9)
    # the stored samples nearest to logp = 0 (what fit() measures a reduced kernel on) are found where the
    # chain lies, the same rows as from the host copy
    r_dev, l_dev = dev.sampler.rows_nearest_zero_logp(64)
    lp_all = dev.sampler.get_log_prob(flat=True)
    assert r_dev.shape == (64, 4) and np.abs(l_dev).max() <= np.sort(np.abs(lp_all))[63]
    with pytest.raises(ValueError):
        dev.fit(chain='somewhere')


def test_fit_measures_its_reduced_kernel_on_the_final_ensemble(monkeypatch):
    """AUTO picks a QR-reduced kernel from an estimate on probe rows; fit() then measures that kernel on
    the ensemble it ends with, against long double, keeps the result in reduced_check_ and warns past
    the parity tolerance (forced here by a tolerance no double arithmetic meets)."""
    import warnings
    import bisip_amd
    path = bisip_amd.DataFiles()['SIP-K389175']
    m = bisip_amd.PolynomialDecomposition(path, nwalkers=64, nsteps=200)
    np.random.seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter('error', RuntimeWarning)
        m.fit()
    # the worse of two measurements: the final ensemble (around the mode: 1e-14) and the stored samples
    # nearest to logp = 0 -- this run starts uniform in the prior box and every walker crosses it on its
    # way in: there the plain triangle of this degree-5 design is 1e-11 off, within the bar AUTO holds it to
    assert m._context().variant == 'reduced' and 0.0 <= m.reduced_check_ <= 2e-11
    ctx = m._context()
    coords, lp = m.sampler._coords.copy(), m.sampler._lp.copy()
    final = ctx.reduced_check(coords, lp)
    assert final <= 1e-12 and m.reduced_check_shell_ is not None and m.reduced_check_ == max(final, m.reduced_check_shell_)
    rows, rlp = m.sampler.rows_nearest_zero_logp(256)
    chain_lp = m.sampler.get_log_prob(flat=True)
    assert rows.shape == (256, 7) and np.abs(rlp).max() <= np.sort(np.abs(chain_lp))[255]
    assert ctx.reduced_check(rows, rlp) == m.reduced_check_shell_
    # rows outside the prior are the prior's business; a wrong log-probability is found
    lp[5] += 1e-6 * abs(lp[5])
    assert 0.5e-6 < ctx.reduced_check(coords, lp) < 2e-6
    coords[5, 0] = 2.0                               # now outside the box: not looked at
    assert ctx.reduced_check(coords, lp) <= 1e-12
    monkeypatch.setattr(bisip_amd.PolynomialDecomposition, '_LOGP_TOL', 1e-30)
    with pytest.warns(RuntimeWarning, match="variant='reduced_comp'"):
        m.fit()
    # models without a reduced form: nothing to check, nothing raised
    cc = bisip_amd.PeltonColeCole(path, nwalkers=32, nsteps=20)
    cc.fit()
    assert cc.reduced_check_ is None
    with pytest.raises(RuntimeError):
        cc._context().reduced_check(cc.sampler._coords, cc.sampler._lp)
    # a batch of spectra measures every ensemble
    from bisip_amd.synthetic import synthetic_columns
    b = bisip_amd.SpectraBatch('PolynomialDecomposition', [synthetic_columns(32, i) for i in range(5)], nwalkers=32, nsteps=50)
    b.fit(seed=1)
    assert 0.0 <= b.reduced_check_ <= 1e-11
    b.close()


def test_model_percentiles_on_the_device():
    """get_model_percentile (src/bisip/utils.py:17-35: forward() over the chain, np.percentile over
    axis 0) as ONE library call -- forward and the per-(part, frequency) percentiles both on the
    device -- against NumPy's percentile of the batched forward, and of the oracle's forward."""
    import bisip_amd
    import oracle
    path = bisip_amd.DataFiles()['SIP-K389175']
    for cls, kw in ((bisip_amd.PolynomialDecomposition, {}), (bisip_amd.PeltonColeCole, dict(n_modes=2))):
        m = cls(path, nwalkers=32, nsteps=10, **kw)
        lo, hi = m.param_bounds
        rng = np.random.RandomState(8)
        chain = rng.uniform(lo, hi, (5003, lo.size))
        p = [2.5, 50, 97.5, 0, 100, 33.3]
        got = m.get_model_percentile(p, chain)
        assert got.shape == (6, 2, 20)
        Z = m.forward(chain, m.data['w'])
        assert np.array_equal(got, np.percentile(Z, p, axis=0))      # the same order statistics, weights and interpolation
        d = m.data
        okw = dict(taus=m.taus, log_taus=m.log_taus, c_exp=m.c_exp) if cls is bisip_amd.PolynomialDecomposition else dict(n_modes=2)
        prob = oracle.OracleProblem(cls.__name__, d['w'], d['zn'], d['zn_err'], m.param_bounds, **okw)
        assert_Z_close(got, np.percentile(oracle.forward(prob, chain), p, axis=0))
        # a scalar percentile gives (2, N), as np.percentile does
        assert m.get_model_percentile(50, chain).shape == (2, 20)
        assert np.array_equal(m.get_model_percentile(50, chain), got[1])
        # NumPy's behaviour at the edges: a percentile outside [0, 100] is a ValueError, and a NaN in the
        # chain (a NaN response in every column) makes every percentile NaN
        with pytest.raises(ValueError):
            m.get_model_percentile([50, 101], chain)
        bad = chain.copy()
        bad[7, 0] = np.nan
        with np.errstate(invalid='ignore'):
            assert np.isnan(m.get_model_percentile([2.5, 50], bad)).all()


def test_fit_keeps_numpys_stream_by_default_and_recommends_philox_for_big_ensembles():
    """fit()'s default random stream is NumPy's, in emcee's order, at EVERY ensemble size: np.random.seed pins
    the chain (the reference's quirk), bit for bit the host sampler's.  From 2048
    walkers on a UserWarning recommends rng='philox' (the device stream, 3-4x faster there); rng='auto'
    picks by size without a word; an explicit rng never warns."""
    import warnings
    import bisip_amd
    path = bisip_amd.DataFiles()['SIP-K389175']
    small = bisip_amd.PolynomialDecomposition(path, nwalkers=32, nsteps=20)
    np.random.seed(4)
    with warnings.catch_warnings():
        warnings.simplefilter('error', UserWarning)
        small.fit()
    assert small.sampler.rng == 'numpy'
    host = bisip_amd.PolynomialDecomposition(path, nwalkers=32, nsteps=20)
    np.random.seed(4)
    host.fit(sampler='host')
    assert np.array_equal(small.get_chain(), host.get_chain())
    big = bisip_amd.PolynomialDecomposition(path, nwalkers=2048, nsteps=10)
    centre = np.array([1.0, 0.005, -0.003, -0.001, 0.0005, 0.0002, 0.00001])
    p0 = centre + 1e-4 * np.random.RandomState(1).randn(2048, 7)
    np.random.seed(4)
    with pytest.warns(UserWarning, match="rng='philox'"):
        big.fit(p0)
    assert big.sampler.rng == 'numpy' and big.get_chain().shape == (10, 2048, 7)
    default_chain = big.get_chain()
    host = bisip_amd.PolynomialDecomposition(path, nwalkers=2048, nsteps=10)
    np.random.seed(4)
    host.fit(p0, sampler='host')
    assert np.array_equal(default_chain, host.get_chain())          # the seed pins a big run too
    with warnings.catch_warnings():
        warnings.simplefilter('error', UserWarning)
        np.random.seed(4)
        big.fit(p0, rng='auto')
        assert big.sampler.rng == 'philox'
        first = big.get_chain()
        np.random.seed(4)                    # the Philox key comes from the seeded global state
        big.fit(p0, rng='philox')
        assert np.array_equal(big.get_chain(), first) and not np.array_equal(first, default_chain)
        np.random.seed(4)
        big.fit(p0, rng='numpy')
        assert big.sampler.rng == 'numpy' and np.array_equal(big.get_chain(), default_chain)


def test_faithful_variant_fit_uses_the_host_loop():
    """The stretch-move kernels exist for the reduced / collapsed formulations; a model pinned
    to `faithful` samples through the host loop around that very kernel, and the C entry
    refuses instead of silently evaluating another formulation."""
    import bisip_amd
    from bisip_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
    path = bisip_amd.DataFiles()['SIP-K389175']
    m = bisip_amd.PolynomialDecomposition(path, nwalkers=32, nsteps=12, variant='faithful')
    np.random.seed(4)
    m.fit()
    assert isinstance(m.sampler, EnsembleSampler)
    lp = m.sampler.get_log_prob()[-1]
    assert np.array_equal(lp, m._context().logprob(m.get_chain()[-1]))   # the faithful kernel's bits
    dev = DeviceEnsembleSampler(32, 7, m._context(), persistent=False)
    with pytest.raises(RuntimeError, match='faithful'):
        dev.run_mcmc(m.p0, 2)


def test_fit_runs_and_matches_oracle_replay():
    """model.fit() on the GPU reproduces, step for step, the same sampler driven by the
    CPU oracle (accept decisions and positions identical; log-probs to tolerance)."""
    import bisip_amd
    import oracle
    from bisip_amd.sampler import EnsembleSampler
    path = bisip_amd.DataFiles()['SIP-K389175']
    m = bisip_amd.PeltonColeCole(path, nwalkers=32, nsteps=300, n_modes=2)
    np.random.seed(42)
    m.fit()
    chain = m.get_chain()
    assert chain.shape == (300, 32, 7) and m.fitted
    d = m.data
    prob = oracle.OracleProblem('PeltonColeCole', d['w'], d['zn'], d['zn_err'], m.param_bounds,
                                n_modes=2)
    np.random.seed(42)
    p0 = np.random.uniform(*m.param_bounds, (32, 7))
    assert np.array_equal(p0, m.p0)
    ref = EnsembleSampler(32, 7, lambda t: oracle.logprob(prob, t))
    ref.run_mcmc(p0, 300)
    assert np.array_equal(chain, ref.get_chain())
    assert_logp_close(m.sampler.get_log_prob(), ref.get_log_prob())
    # reference workflow after the fit
    flat = m.get_chain(discard=100, thin=2, flat=True)
    assert flat.shape == (100 * 32, 7)
    assert m.get_param_mean(flat).shape == (7,) and m.get_param_std(flat).shape == (7,)
    pct = m.get_model_percentile([2.5, 50, 97.5], flat)
    assert pct.shape == (3, 2, 20)
    with pytest.warns(UserWarning):
        m.get_param_mean()


# ----------------------------------------------------------------------------------
# unusual shapes: frequency counts that are not multiples of anything, extreme
# polynomial degrees / mode counts -- checked against the (golden-pinned) oracle
# ----------------------------------------------------------------------------------

def _synthetic_problem(n_freq, idx=1):
    from bisip_amd.synthetic import synthetic_columns
    from bisip_amd.utils import columns_to_data
    return columns_to_data(synthetic_columns(n_freq, idx), 'mrad')


@pytest.mark.parametrize('n_freq,poly_deg,c_exp', [(1, 0, 1.0), (2, 5, 1.0), (1, 3, 0.5), (3, 1, 0.5), (17, 2, 1.0), (20, 7, 0.8),
                                                     (33, 5, 1.0), (100, 3, 0.3), (200, 5, 1.0),
                                                     (12, 10, 1.0)])
def test_polydecomp_unusual_shapes(n_freq, poly_deg, c_exp):
    import oracle
    from bisip_amd import _hip
    d = _synthetic_problem(n_freq)
    per = np.log10(1. / d['w'])
    lt = np.linspace(np.floor(per.min() - 1), np.floor(per.max() + 1), max(2 * n_freq, 2))
    taus, log_taus = 10 ** lt, np.array([lt ** i for i in range(poly_deg + 1)])
    bounds = np.array([[0.9] + [-1.0] * (poly_deg + 1), [1.1] + [1.0] * (poly_deg + 1)])
    prob = oracle.OracleProblem('PolynomialDecomposition', d['w'], d['zn'], d['zn_err'], bounds,
                                taus=taus, log_taus=log_taus, c_exp=c_exp)
    rng = np.random.RandomState(n_freq * 100 + poly_deg)
    theta = rng.uniform(bounds[0], bounds[1], (700, poly_deg + 2))
    theta[:350, 1:] *= 1e-3          # a cloud where the fit is decent (small |logp|)
    want = oracle.logprob(prob, theta, n_threads=4)
    variants = ['reduced', 'collapsed'] + (['faithful'] if poly_deg <= 7 else []) + \
        (['wave'] if n_freq <= 64 else [])
    for v in variants:
        ctx = _hip.HipContext(0, d['w'], d['zn'], d['zn_err'], bounds, poly_deg=poly_deg, c_exp=c_exp,
                              taus=taus, log_taus=log_taus, variant=v)
        assert_logp_close(ctx.logprob(theta), want)
        if v == 'collapsed':
            assert_Z_close(ctx.forward(theta[:50]), oracle.forward(prob, theta[:50]))
        ctx.close()


@pytest.mark.parametrize('model,n_modes,n_freq', [('PeltonColeCole', 4, 7), ('PeltonColeCole', 5, 45),
                                                   ('PeltonColeCole', 1, 1), ('Dias2000', 0, 5),
                                                   ('Dias2000', 0, 130), ('Shin2015', 0, 3),
                                                   ('Shin2015', 0, 77)])
def test_other_models_unusual_shapes(model, n_modes, n_freq):
    import oracle
    from bisip_amd import _hip
    from bisip_amd.batch import default_params
    d = _synthetic_problem(n_freq, 2)
    bounds = np.array(list(default_params(model, n_modes=n_modes).values()), float).T
    kw = dict(n_modes=n_modes) if model == 'PeltonColeCole' else {}
    prob = oracle.OracleProblem(model, d['w'], d['zn'], d['zn_err'], bounds, **kw)
    rng = np.random.RandomState(n_freq)
    theta = rng.uniform(bounds[0], bounds[1], (900, bounds.shape[1]))
    ctx = _hip.HipContext(MODEL_IDS[model], d['w'], d['zn'], d['zn_err'], bounds, **kw)
    assert_logp_close(ctx.logprob(theta), oracle.logprob(prob, theta, n_threads=4))
    assert_Z_close(ctx.forward(theta[:64]), oracle.forward(prob, theta[:64]))
    ctx.close()


@pytest.mark.parametrize('model,n_modes,n_freq', [('PeltonColeCole', 2, 32), ('PeltonColeCole', 3, 21),
                                                   ('PeltonColeCole', 1, 1), ('Dias2000', 0, 22),
                                                   ('Shin2015', 0, 23), ('Dias2000', 0, 2)])
def test_lanes_per_walker_do_not_change_bits(model, n_modes, n_freq):
    """Launches of <= 16384 walkers use 4 lanes per walker, <= 32768 two, larger ones one
    (dispatch_logprob.hip:lanes_per_walker); frequency counts that are not multiples of the lane count
    leave a partial last round.  The running sums are handed from lane to lane in frequency
    order, so every regime must return the SAME BITS for the same row."""
    from bisip_amd import _hip
    from bisip_amd.batch import default_params
    d = _synthetic_problem(n_freq, 3)
    bounds = np.array(list(default_params(model, n_modes=n_modes).values()), float).T
    kw = dict(n_modes=n_modes) if model == 'PeltonColeCole' else {}
    ctx = _hip.HipContext(MODEL_IDS[model], d['w'], d['zn'], d['zn_err'], bounds, **kw)
    rng = np.random.RandomState(n_freq + 17)
    theta = rng.uniform(bounds[0], bounds[1], (140000, bounds.shape[1]))
    theta[5] = bounds[1]                               # an out-of-prior row inside a lane group
    one = ctx.logprob(theta)                           # 1 lane per walker (and the 256-lane workgroups)
    mid = ctx.logprob(theta[:40000])                   # 1 lane, 64-lane workgroups
    two = ctx.logprob(theta[:20000])                   # 2 lanes
    four = ctx.logprob(theta[:5000])                   # 4 lanes
    tiny = ctx.logprob(theta[:3])                      # 4 lanes, one partial wave
    assert np.isneginf(one[5]) and np.all(np.isfinite(one[:5]))
    assert np.array_equal(mid, one[:40000])
    assert np.array_equal(two, one[:20000])
    assert np.array_equal(four, one[:5000])
    assert np.array_equal(tiny, one[:3])
    ctx.close()


@pytest.mark.parametrize('n_freq,poly_deg,c_exp', [(5, 8, 0.3805172729737878), (3, 10, 1.0), (48, 9, 0.21152054037418078),
                                                     (4, 6, 0.5), (2, 10, 0.3)])
def test_reduced_form_on_ill_conditioned_designs(n_freq, poly_deg, c_exp):
    """Few frequencies / high polynomial degree: the weighted design matrix is (nearly) rank
    deficient and its least-squares solution is astronomically large.  The reduced form
    expands about that solution CLAMPED INTO THE PRIOR BOX (host_precompute.h:reduced_center),
    otherwise the cancellation inside R (bhat - b) costs up to 1e-2 of the log-probability
    (found by benchmarks/fuzz_parity.py).  Also: moving the box moves the expansion point."""
    import oracle
    from bisip_amd import _hip
    d = _synthetic_problem(n_freq, 5)
    S = 2 * n_freq
    per = np.log10(1. / d['w'])
    lt = np.linspace(np.floor(per.min() - 1), np.floor(per.max() + 1), S)
    taus, log_taus = 10 ** lt, np.array([lt ** i for i in range(poly_deg + 1)])
    bounds = np.array([[0.9] + [-1.0] * (poly_deg + 1), [1.1] + [1.0] * (poly_deg + 1)])
    rng = np.random.RandomState(n_freq * 31 + poly_deg)
    ctx = _hip.HipContext(0, d['w'], d['zn'], d['zn_err'], bounds, poly_deg=poly_deg, c_exp=c_exp,
                          taus=taus, log_taus=log_taus, variant='reduced')
    for box in (bounds, bounds * np.r_[1.0, np.full(poly_deg + 1, 1e-3)], np.array([bounds[0] + 0.3 * (bounds[1] - bounds[0]), bounds[1]])):
        ctx.set_bounds(box)
        prob = oracle.OracleProblem('PolynomialDecomposition', d['w'], d['zn'], d['zn_err'], box,
                                    taus=taus, log_taus=log_taus, c_exp=c_exp)
        theta = rng.uniform(box[0], box[1], (4000, poly_deg + 2))
        assert_logp_close(ctx.logprob(theta), oracle.logprob(prob, theta, n_threads=4))
    ctx.close()


def _pd_context(n_freq, poly_deg, c_exp, idx=5, variant='auto'):
    from bisip_amd import _hip
    d = _synthetic_problem(n_freq, idx)
    per = np.log10(1. / d['w'])
    lt = np.linspace(np.floor(per.min() - 1), np.floor(per.max() + 1), 2 * n_freq)
    bounds = np.array([[0.9] + [-1.0] * (poly_deg + 1), [1.1] + [1.0] * (poly_deg + 1)])
    taus, log_taus = 10 ** lt, np.array([lt ** i for i in range(poly_deg + 1)])
    ctx = _hip.HipContext(0, d['w'], d['zn'], d['zn_err'], bounds, poly_deg=poly_deg, c_exp=c_exp,
                          taus=taus, log_taus=log_taus, variant=variant)
    return ctx, bounds, d, taus, log_taus


def test_auto_variant_follows_the_reduced_kernels_error_estimate():
    """AUTO = the plain QR-reduced kernel where a host-side emulation of its double arithmetic
    stays within 1e-12 of long double on probe rows (bisip_ctx_reduced_error), else the
    compensated reduced kernel under the same test, else the per-frequency form; the estimates
    are refreshed when the prior box changes."""
    ctx, *_ = _pd_context(32, 5, 1.0, 0)             # the headline shape
    assert ctx.variant == 'reduced' and ctx.kernel_name == 'k_logprob_pd_reduced'
    # 5e-13 = 9e-12 absolute on the shell logp = 0, counted at a twentieth (reduced_center); 1e-15 elsewhere
    assert ctx.reduced_error < 1e-12
    ctx.close()
    picked = []
    for n_freq, poly_deg, c_exp in [(32, 8, 1.0), (48, 7, 0.5), (80, 10, 0.5), (21, 10, 0.5), (64, 9, 0.3),
                                    (32, 6, 1.0), (20, 10, 1.0), (33, 10, 0.5)]:
        ctx, *_ = _pd_context(n_freq, poly_deg, c_exp, 0)
        assert ctx.variant in ('reduced', 'reduced_comp'), (n_freq, poly_deg, c_exp, ctx.variant)
        assert ctx.reduced_error <= 1e-12
        assert ctx.kernel_name == ('k_logprob_pd_reduced' if ctx.variant == 'reduced' else 'k_logprob_pd_reduced_comp')
        picked.append(ctx.variant)
        est_auto = ctx.reduced_error
        ctx.set_variant('reduced')                    # each tier is still available on request
        est_plain = ctx.reduced_error
        ctx.set_variant('reduced_comp')
        est_comp = ctx.reduced_error
        assert ctx.kernel_name == 'k_logprob_pd_reduced_comp'
        assert est_auto == (est_plain if est_plain <= 1e-12 else est_comp)
        assert est_comp <= 1e-12
        ctx.close()
    assert 'reduced_comp' in picked                   # nearly collinear designs need the compensated sums
    ctx, bounds, *_ = _pd_context(3, 10, 1.0)         # fewer data rows than unknowns: no triangle
    assert ctx.variant == 'collapsed'
    ctx.close()
    ctx, bounds, *_ = _pd_context(32, 8, 1.0, 0)
    before = ctx.reduced_error
    ctx.set_bounds(bounds * np.r_[1.0, np.full(9, 1e-4)])       # the estimate follows the prior box
    after = ctx.reduced_error
    assert np.isfinite(after) and after != before
    assert ctx.variant in ('reduced', 'reduced_comp') and after <= 1e-12
    ctx.close()


@pytest.mark.parametrize('n_freq,poly_deg,c_exp,idx', [(32, 10, 0.5, 0), (33, 10, 0.5, 3), (20, 10, 0.5, 7), (80, 10, 0.5, 1),
                                                         (32, 8, 0.5, 2), (64, 9, 1.0, 4), (48, 6, 0.3, 5)])
def test_posterior_valley_walkers_on_nearly_collinear_designs(n_freq, poly_deg, c_exp, idx):
    """Where an ensemble sampler's walkers actually sit on a design of degree 6-10: draws from the
    Gaussian posterior of the linear model, spread along the flat valley of chi^2.  There every
    row of R (bhat - b), and every per-frequency residual of the reference's own sum, cancels by
    many orders of magnitude.  Every formulation must stay within the parity tolerance of the
    oracle, and the compensated reduced kernel -- what AUTO runs when the plain one's estimate is
    too large -- must agree with a 50-digit evaluation far better than that."""
    import oracle
    from bisip_amd import _hip
    ctx, bounds, d, taus, log_taus = _pd_context(n_freq, poly_deg, c_exp, idx)
    n = poly_deg + 2
    ops = _hip.polydecomp_operands(d['w'], d['zn'], d['zn_err'], taus, log_taus, c_exp)
    bls = ops['bhat'].astype(np.longdouble)
    rng = np.random.RandomState(poly_deg * 100 + n_freq)
    rows = []
    for scale in (1.0, 3.0):
        z = scale * rng.randn(400, n)
        db = np.array([np.linalg.solve(ops['R'], zi) for zi in z])
        b = bls[None, :] + db
        t = np.concatenate([b[:, :1], b[:, 1:] / b[:, :1]], axis=1).astype(np.float64)
        rows.append(t[np.all((bounds[0] < t) & (t < bounds[1]), axis=1)])
    theta = np.concatenate(rows)
    if len(theta) < 20:
        pytest.skip('the posterior of this design lies outside the default prior box')
    prob = oracle.OracleProblem('PolynomialDecomposition', d['w'], d['zn'], d['zn_err'], bounds,
                                taus=taus, log_taus=log_taus, c_exp=c_exp)
    want = oracle.logprob(prob, theta)
    assert np.all(np.isfinite(want)) and np.ptp(want) < 1e4         # all near the mode
    # the exact value of the reference's formula: the library's host-only yardstick (nothing rounded to
    # double on the way; pinned by 50-digit arithmetic in tests/test_host_logic.py)
    exact = _hip.polydecomp_reduced_reference(d['w'], d['zn'], d['zn_err'], taus, log_taus, c_exp, theta)
    errs = {}
    for variant in ('auto', 'reduced', 'reduced_comp', 'collapsed'):
        ctx.set_variant(variant)
        got = ctx.logprob(theta)
        errs[variant] = (assert_logp_close(got, want),
                         float(np.max(np.abs(got - exact) / np.maximum(1.0, np.abs(exact)))))
        # the library's own after-the-fact check measures exactly that distance
        mine = ctx.reduced_check(theta, got)
        assert abs(mine - errs[variant][1]) <= 1e-3 * errs[variant][1] + 1e-17, (variant, mine, errs[variant][1])
    print(n_freq, poly_deg, c_exp, len(theta), {k: ('%.1e' % a, '%.1e' % b) for k, (a, b) in errs.items()})
    assert errs['reduced_comp'][1] <= 1e-12
    ctx.set_variant('auto')
    assert ctx.variant in ('reduced', 'reduced_comp') and errs['auto'][1] <= 2e-12
    ctx.close()


def _shell_rows(ops, bounds, n_rows, seed):
    """Rows ON the shell log-probability = 0 of the linear model: b = b_ls + s R^-1 z with s such that
    rest + s^2 |z|^2 = 2 lconst.  There the tolerance's denominator max(1, |logp|) is 1 and the absolute
    error of a chi^2 of several hundred counts."""
    n = ops['R'].shape[0]
    z = np.random.RandomState(seed).randn(n_rows, n)
    sc = np.sqrt((2.0 * ops['lconst'] - ops['rest']) / (z * z).sum(axis=1))
    db = np.linalg.solve(ops['R'], (z * sc[:, None]).T).T
    b = ops['bhat'][None, :] + db
    t = np.concatenate([b[:, :1], b[:, 1:] / b[:, :1]], axis=1)
    return np.ascontiguousarray(t[np.all((bounds[0] < t) & (t < bounds[1]), axis=1)])


def test_guard_rows_entry_measures_and_moves_only_what_auto_chose(monkeypatch):
    """bisip_ctx_reduced_guard_rows -- the guard for rows a caller brings from the device: measures them as
    bisip_ctx_reduced_check does; moves a context that chose its tier itself ('auto') past 2e-11 and says so; only
    measures a forced variant or a context whose guard is switched off; has nothing to measure once the
    per-frequency form runs; refuses models without a reduced form."""
    from bisip_amd import _hip
    n_freq, poly_deg, c_exp, idx = 64, 6, 1.0, 2
    monkeypatch.setenv('BISIP_SHELL_WEIGHT', '0')
    ctx, bounds, d, taus, log_taus = _pd_context(n_freq, poly_deg, c_exp, idx)
    forced, *_ = _pd_context(n_freq, poly_deg, c_exp, idx, variant='reduced')
    monkeypatch.delenv('BISIP_SHELL_WEIGHT')
    ops = _hip.polydecomp_operands(d['w'], d['zn'], d['zn_err'], taus, log_taus, c_exp)
    theta = _shell_rows(ops, bounds, 3000, 5)[:512]
    ctx.reduced_guard(False)
    plain = ctx.logprob(theta)                      # what the plain kernel says on the shell (no guard: nothing moves)
    far = ctx.reduced_check(theta, plain)
    assert far > 2e-11 and ctx.variant == 'reduced'
    assert ctx.reduced_guard_rows(theta, plain) == (far, False) and ctx.variant == 'reduced'      # guard off: measured only
    ctx.reduced_guard(True)
    rows = np.vstack([theta, np.full((3, theta.shape[1]), np.nan)])                                   # unfilled slots are skipped
    worst, moved = ctx.reduced_guard_rows(rows, np.r_[plain, np.nan, np.nan, np.nan])
    assert worst == far and moved and ctx.variant == 'reduced_comp' and ctx.reduced_guard()[2] == 1
    with pytest.warns(RuntimeWarning, match='k_logprob_pd_reduced_comp'):      # the next host-buffer call reports the move
        comp = ctx.logprob(theta)
    worst, moved = ctx.reduced_guard_rows(theta, comp)
    assert worst <= 2e-12 and not moved and ctx.variant == 'reduced_comp'
    assert forced.reduced_guard_rows(theta, forced.logprob(theta)) == (far, False) and forced.variant == 'reduced'
    forced.set_variant('collapsed')
    assert forced.reduced_guard_rows(theta, forced.logprob(theta)) == (0.0, False)
    with pytest.raises(ValueError):
        ctx.reduced_guard_rows(theta, comp[:-1])
    cc = make_ctx(np.load([p for p in golden_cases() if 'case15_' in p][0]), 'PeltonColeCole')
    with pytest.raises(RuntimeError):
        cc.reduced_guard_rows(np.zeros((2, 4)), np.zeros(2))
    for c in (ctx, forced, cc):
        c.close()


def test_logprob_guard_moves_a_context_off_a_reduced_kernel_that_fails_on_its_batch(monkeypatch):
    """bisip_logprob -- what log_prob() and emcee's vectorised callback call -- measures the QR-reduced
    kernel on rows of the caller's own batch (first call, then every 2^n-th) and a context on 'auto' that
    is more than 2e-11 off moves to the compensated kernel and evaluates the batch again: no fit() needed.
    The batch: rows on the shell logp = 0 of a degree-6 design, where the plain triangle's cancellation
    shows as an absolute error.  The context: built with the shell probes of the estimate switched off
    (BISIP_SHELL_WEIGHT=0, i.e. round 2's estimate), so that its estimate passes and AUTO starts on the
    plain kernel -- with them on, AUTO picks the compensated kernel by itself."""
    import warnings
    from bisip_amd import _hip
    n_freq, poly_deg, c_exp, idx = 64, 6, 1.0, 2
    monkeypatch.setenv('BISIP_SHELL_WEIGHT', '0')
    ctx, bounds, d, taus, log_taus = _pd_context(n_freq, poly_deg, c_exp, idx)
    monkeypatch.delenv('BISIP_SHELL_WEIGHT')
    assert ctx.variant == 'reduced' and ctx.reduced_error <= 1e-12            # the estimate passed
    ops = _hip.polydecomp_operands(d['w'], d['zn'], d['zn_err'], taus, log_taus, c_exp)
    theta = _shell_rows(ops, bounds, 6000, 5)
    assert len(theta) > 1000
    # what the plain kernel does on these rows (guard off), measured by the library's own checker
    ctx.reduced_guard(False)
    plain = ctx.logprob(theta)
    assert np.abs(plain).max() < 1.0                                          # on the shell
    far = ctx.reduced_check(theta[::max(1, len(theta) // 256)], plain[::max(1, len(theta) // 256)])
    assert far > 2e-11, far
    ctx.reduced_guard(True)
    with pytest.warns(RuntimeWarning, match='k_logprob_pd_reduced_comp'):
        got = ctx.logprob(theta)
    checks, worst, moves = ctx.reduced_guard()
    assert ctx.variant == 'reduced_comp' and moves == 1 and checks == 2 and worst == far
    assert ctx.reduced_check(theta, got) <= 2e-12
    ref = _pd_context(n_freq, poly_deg, c_exp, idx, variant='reduced_comp')[0]
    assert np.array_equal(got, ref.logprob(theta))                            # the batch was re-evaluated
    ref.close()
    # the choice holds for later calls (no new warning), until the prior box changes
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        for _ in range(9):
            assert np.array_equal(ctx.logprob(theta[:50]), got[:50])
    assert ctx.reduced_guard()[2] == 1 and ctx.reduced_guard()[0] >= 4        # calls 2, 4, 8 were measured too
    ctx.close()
    # with the shell probes in the estimate AUTO never starts on the plain kernel here
    ctx2 = _pd_context(n_freq, poly_deg, c_exp, idx)[0]
    assert ctx2.variant == 'reduced_comp'
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        assert ctx2.reduced_check(theta, ctx2.logprob(theta)) <= 2e-12
    assert ctx2.reduced_guard()[2] == 0
    ctx2.close()
    # a variant the caller forced is measured, named in a warning past the tolerance, and left alone
    monkeypatch.setenv('BISIP_SHELL_WEIGHT', '0')
    ctx3 = _pd_context(48, 9, 1.0, 0, variant='reduced')[0]
    monkeypatch.delenv('BISIP_SHELL_WEIGHT')
    d3, taus3, log_taus3 = _pd_context(48, 9, 1.0, 0)[2:]
    ops3 = _hip.polydecomp_operands(d3['w'], d3['zn'], d3['zn_err'], taus3, log_taus3, 1.0)
    rows3 = _shell_rows(ops3, bounds=np.array([[0.9] + [-1.0] * 10, [1.1] + [1.0] * 10]), n_rows=20000, seed=6)
    if len(rows3) > 256:
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter('always')
            ctx3.logprob(rows3)
        assert ctx3.variant == 'reduced' and ctx3.reduced_guard()[2] == 0
        if ctx3.reduced_guard()[1] > 1e-10:
            assert any("variant='auto'" in str(w.message) for w in rec)
    ctx3.close()


def test_shared_reciprocal_is_switched_off_by_boxes_it_would_overflow_in():
    """ColeCole<D> (D >= 2) and Shin take ONE reciprocal per group of denominators of a frequency when
    the prior box keeps their products normal (bound_flags, kernels.h: rcp_batch_n).  Default boxes and a
    box widened to tau = e^60 s still do; a box that lets c reach 30 (denominators that overflow) must
    fall back to one reciprocal per term -- a NaN here would be the shared reciprocal at work."""
    import oracle
    from bisip_amd import _hip
    from bisip_amd.batch import default_params
    d = _synthetic_problem(32, 1)
    rng = np.random.RandomState(8)
    for n_modes, widen in ((2, None), (4, None), (4, 'tau'), (2, 'c'), (3, 'c')):
        bounds = np.array(list(default_params('PeltonColeCole', n_modes=n_modes).values()), float).T
        D = n_modes
        if widen == 'tau':
            bounds[1, 1 + D:1 + 2 * D] = 60.0
        if widen == 'c':
            bounds[1, 1 + 2 * D:] = 30.0
        theta = rng.uniform(bounds[0], bounds[1], (3000, bounds.shape[1]))
        if widen == 'tau':
            theta[:500, 1 + D:1 + 2 * D] = rng.uniform(55, 60, (500, D))     # denominators ~ 2^200 each
            theta[:500, 1 + 2 * D:] = rng.uniform(0.95, 1.0, (500, D))
        if widen == 'c':
            theta[:500, 1 + 2 * D:] = rng.uniform(25, 30, (500, D))           # e^2 overflows
            theta[:500, 1 + D:1 + 2 * D] = rng.uniform(3, 5, (500, D))
        prob = oracle.OracleProblem('PeltonColeCole', d['w'], d['zn'], d['zn_err'], bounds, n_modes=n_modes)
        with np.errstate(all='ignore'):
            want = oracle.logprob(prob, theta, n_threads=4)
        ctx = _hip.HipContext(1, d['w'], d['zn'], d['zn_err'], bounds, n_modes=n_modes)
        got = ctx.logprob(theta)
        ok = np.isfinite(want)
        assert ok.mean() > 0.8 and np.isfinite(got[ok]).all(), (n_modes, widen)
        err = np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))
        assert err.max() <= 1e-10, (n_modes, widen, err.max())
        # the same rows through set_bounds on a context created with the default box
        ctx.close()
    # Shin: R next to 0 inside the open box (1/R clamped at 2^110: the partner element and -- one reciprocal
    # serves the four |y|^2 of a pair of frequencies -- the neighbouring frequency must survive; both R tiny:
    # the product of four clamped denominators is still a normal number)
    bounds = np.array(list(default_params('Shin2015').values()), float).T
    theta = rng.uniform(bounds[0], bounds[1], (2000, 6))
    theta[:300, 0] = 10.0 ** rng.uniform(-300, -60, 300)
    theta[300:600, 1] = 10.0 ** rng.uniform(-300, -60, 300)
    theta[600:700, :2] = 10.0 ** rng.uniform(-300, -30, (100, 2))
    prob = oracle.OracleProblem('Shin2015', d['w'], d['zn'], d['zn_err'], bounds)
    with np.errstate(all='ignore'):
        want = oracle.logprob(prob, theta, n_threads=4)
    ctx = _hip.HipContext(3, d['w'], d['zn'], d['zn_err'], bounds)
    assert_logp_close(ctx.logprob(theta), want)
    ctx.close()
    # Shin in a box widened x3 (campaign seed 46, case 623): R2 < 0 and n1 -> 2 make Q (iw)^n and 1/R cancel to
    # a thousandth of either at w = 5744.46 -- forward() amplifies the rounding of the power a thousandfold and
    # read 2.4e-12 of max|Z| with one fused exponent; with the reference's roundings it stays inside 1e-12
    row = np.array([[0.51204783, -0.27030474, -16.63525857, -7.76820352, 1.99925033, -0.89533083]])
    wide = np.array([[-1.0, -1.0, -17.0, -9.0, -1.0, -1.0], [2.0, 2.0, -11.0, -3.0, 2.0, 2.0]])
    w = d['w'].copy()
    w[5] = 5744.464998942009
    near = row + 1e-6 * rng.uniform(-1, 1, (400, 6))
    prob = oracle.OracleProblem('Shin2015', w, d['zn'], d['zn_err'], wide)
    ctx = _hip.HipContext(3, w, d['zn'], d['zn_err'], wide)
    assert ctx.loop_flags == 0
    rows = np.concatenate([row, near])
    want_Z = oracle.forward(prob, rows)
    assert np.abs(want_Z[0, :, 5]).max() > 100.0            # the cancellation is there
    assert_Z_close(ctx.forward(rows), want_Z)
    assert_logp_close(ctx.logprob(rows), oracle.logprob(prob, rows))
    ctx.close()
    # Dias: frequencies 2k and 2k+1 share ONE reciprocal (of the product of their denominators) while the box
    # keeps that product a normal number -- the reference's box does, also with delta -> 0 and m -> 1 (tau'
    # clamped at 1e50); a box that lets tau reach e^40 does not and takes a reciprocal per frequency.  Odd and
    # even numbers of frequencies (a last unpaired one), one / two / four lanes per walker: the same bits.
    for n_freq in (32, 33, 5):
        dd = _synthetic_problem(n_freq, 3)
        bounds = np.array(list(default_params('Dias2000').values()), float).T
        theta = rng.uniform(bounds[0], bounds[1], (200000, 5))
        theta[:300, 4] = 10.0 ** rng.uniform(-120, -20, 300)            # delta -> 0: tau' at its clamp
        theta[300:600, 1] = 1.0 - 10.0 ** rng.uniform(-16, -3, 300)     # m -> 1
        prob = oracle.OracleProblem('Dias2000', dd['w'], dd['zn'], dd['zn_err'], bounds)
        with np.errstate(all='ignore'):
            want = oracle.logprob(prob, theta[:3000], n_threads=4)
        ctx = _hip.HipContext(2, dd['w'], dd['zn'], dd['zn_err'], bounds)
        assert ctx.loop_flags == 1
        got = ctx.logprob(theta)
        assert_logp_close(got[:3000], want)
        for rows in (100, 6000):
            assert np.array_equal(ctx.logprob(theta[:rows]), got[:rows]), (n_freq, rows)
        wide = bounds.copy()
        wide[1, 2] = 40.0
        ctx.set_bounds(wide)
        assert ctx.loop_flags == 0
        assert_logp_close(ctx.logprob(theta[:3000]), want)
        ctx.close()


@pytest.mark.parametrize('model,n_modes', [('PeltonColeCole', 1), ('PeltonColeCole', 2), ('Shin2015', 0)])
def test_frequency_pairs_share_a_reciprocal_in_every_loop(model, n_modes, monkeypatch):
    """ColeCole<1>, ColeCole<2> and Shin take the denominators of frequencies 2k and 2k+1 from ONE
    reciprocal (kernels.h: rcp_joint) in their FAST loops -- the direct loop of off-grid spectra and the
    stepped loop of geometric grids -- and a last unpaired frequency from its own group.  Odd and even numbers
    of frequencies, block tails of every kind, one / two / four lanes per walker: the same bits; the oracle's
    value within the parity tolerance; exponents at the edge of what BOUNDS_FAST admits (denominators
    ~2^200 each, a product of four ~2^800) included."""
    import oracle
    from bisip_amd import _hip
    from bisip_amd.batch import default_params
    kw = dict(n_modes=n_modes) if model == 'PeltonColeCole' else {}
    rng = np.random.RandomState(31 + n_modes)
    for n_freq in (1, 2, 3, 5, 18, 21, 32, 33):
        d = _synthetic_problem(n_freq, 4)
        bounds = np.array(list(default_params(model, **kw).values()), float).T
        if model == 'PeltonColeCole':
            bounds[1, 1 + n_modes:1 + 2 * n_modes] = 60.0              # tau up to e^60 s: still FAST (y <= 110)
        theta = rng.uniform(bounds[0], bounds[1], (140000, bounds.shape[1]))
        if model == 'PeltonColeCole':
            theta[:400, 1 + n_modes:1 + 2 * n_modes] = rng.uniform(55, 60, (400, n_modes))
            theta[:400, 1 + 2 * n_modes:] = rng.uniform(0.95, 1.0, (400, n_modes))
        prob = oracle.OracleProblem(model, d['w'], d['zn'], d['zn_err'], bounds, **kw)
        with np.errstate(all='ignore'):
            want = oracle.logprob(prob, theta[:2000], n_threads=4)
        for grid in (True, False):
            if grid:
                monkeypatch.delenv('BISIP_NO_GRID', raising=False)
            else:
                monkeypatch.setenv('BISIP_NO_GRID', '1')
            ctx = _hip.HipContext(MODEL_IDS[model], d['w'], d['zn'], d['zn_err'], bounds, **kw)
            assert ctx.loop_flags & 1, (n_freq, grid)                   # the FAST loop
            one = ctx.logprob(theta)                                    # one lane per walker
            assert_logp_close(one[:2000], want)
            for rows in (40000, 20000, 5000, 3):                        # 64-lane workgroups, two lanes, four lanes
                assert np.array_equal(ctx.logprob(theta[:rows]), one[:rows]), (model, n_modes, n_freq, grid, rows)
            ctx.close()
    monkeypatch.delenv('BISIP_NO_GRID', raising=False)


def test_unsupported_shapes_fail_loudly():
    from bisip_amd import _hip
    d = _synthetic_problem(8)
    bounds6 = np.array([[0.9] + [0.0] * 18, [1.1] + [1.0] * 18])
    with pytest.raises(ValueError, match='ndim'):
        _hip.HipContext(1, d['w'], d['zn'], d['zn_err'], bounds6, n_modes=6)          # ndim 19 > 16
    lt = np.linspace(-6, 2, 16)
    with pytest.raises(RuntimeError, match='poly_deg'):                                # status -4
        _hip.HipContext(0, d['w'], d['zn'], d['zn_err'], bounds6[:, :13], poly_deg=11, taus=10 ** lt,
                        log_taus=np.array([lt ** i for i in range(12)]))
    bad = d['zn_err'].copy()
    bad[0, 0] = 0.0
    with pytest.raises(ValueError, match='zn_err'):
        _hip.HipContext(2, d['w'], d['zn'], bad, np.array([[0.9, 0, -20, 0, 0], [1.1, 1, 0, 150, 1.0]]))


def test_device_entry_points_are_graph_capturable():
    """bisip_logprob_dev / bisip_forward_dev do no allocation and no synchronisation, so a
    caller can capture them in a hipGraph and replay."""
    import torch
    path = [p for p in golden_cases() if 'case16_' in p][0]
    g = np.load(path)
    ctx = make_ctx(g, 'PeltonColeCole')
    theta = torch.from_numpy(np.ascontiguousarray(g['theta'][:64])).cuda()
    out = torch.zeros(64, dtype=torch.float64, device='cuda')
    Z = torch.zeros((64, 2, g['w'].size), dtype=torch.float64, device='cuda')
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        ctx.logprob_dev(theta.data_ptr(), 64, out.data_ptr(), side.cuda_stream)   # warm-up outside capture
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            ctx.logprob_dev(theta.data_ptr(), 64, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            ctx.forward_dev(theta.data_ptr(), 64, Z.data_ptr(), torch.cuda.current_stream().cuda_stream)
    want = ctx.logprob(g['theta'][:64])
    for rep in range(3):
        theta.copy_(torch.from_numpy(np.ascontiguousarray(g['theta'][:64][::-1].copy() if rep % 2 else g['theta'][:64])))
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        ref = want[::-1] if rep % 2 else want
        assert np.array_equal(np.isfinite(got), np.isfinite(ref))
        assert np.array_equal(got[np.isfinite(ref)], ref[np.isfinite(ref)])
    ctx.close()


# ----------------------------------------------------------------------------------
# corners of the prior box: parameters a hair inside their bounds (tiny delta, m -> 1,
# c -> 0 and 1, extreme log_tau), where the forward models produce huge or tiny
# intermediates.  The kernels must agree with the oracle there too (or be non-finite
# together).
# ----------------------------------------------------------------------------------

@pytest.mark.parametrize('model,n_modes', [('PeltonColeCole', 1), ('PeltonColeCole', 3), ('Dias2000', 0),
                                           ('Shin2015', 0), ('PolynomialDecomposition', 0)])
def test_prior_box_corners(model, n_modes):
    import itertools
    import oracle
    from bisip_amd import _hip
    from bisip_amd.batch import default_params
    d = _synthetic_problem(32, 4)
    bounds = np.array(list(default_params(model, n_modes=n_modes, poly_deg=3).values()), float).T
    lo, hi = bounds
    nd = lo.size
    kw, okw = {}, {}
    if model == 'PeltonColeCole':
        kw = okw = dict(n_modes=n_modes)
    if model == 'PolynomialDecomposition':
        per = np.log10(1. / d['w'])
        lt = np.linspace(np.floor(per.min() - 1), np.floor(per.max() + 1), 64)
        okw = dict(taus=10 ** lt, log_taus=np.array([lt ** i for i in range(4)]), c_exp=1.0)
        kw = dict(poly_deg=3, **okw)
    rows = []
    rng = np.random.RandomState(12)
    for eps in (1e-3, 1e-9, 1e-15):
        for _ in range(150):
            corner = rng.randint(0, 3, nd)            # 0: near lo, 1: near hi, 2: interior
            t = np.where(corner == 0, eps, np.where(corner == 1, 1 - eps, rng.uniform(0.05, 0.95, nd)))
            rows.append(lo + t * (hi - lo))
    theta = np.array(rows)
    theta = np.minimum(np.maximum(theta, np.nextafter(lo, hi)), np.nextafter(hi, lo))  # strictly inside
    prob = oracle.OracleProblem(model, d['w'], d['zn'], d['zn_err'], bounds, **okw)
    with np.errstate(all='ignore'):
        want = oracle.logprob(prob, theta, n_threads=4)
    ctx = _hip.HipContext(MODEL_IDS[model], d['w'], d['zn'], d['zn_err'], bounds, **kw)
    got = ctx.logprob(theta)
    both = np.isfinite(want) & np.isfinite(got)
    assert both.mean() > 0.9, 'most corner rows should be finite in both'
    # wherever the reference is finite the kernel must be finite and agree
    assert np.isfinite(got[np.isfinite(want)]).all()
    err = np.abs(got[both] - want[both]) / np.maximum(1.0, np.abs(want[both]))
    assert err.max() <= 1e-10, err.max()
    print(f'{model}: {both.sum()}/{len(want)} finite, max rel err {err.max():.2e}')
    ctx.close()


def test_plain_c_consumer(tmp_path):
    """examples/c_abi_demo.c: the boundary is usable from plain C (no Python, no torch)."""
    import os
    import subprocess
    import oracle
    from conftest import ROOT
    exe = str(tmp_path / 'c_abi_demo')
    libdir = os.path.join(ROOT, 'bisip_amd')
    subprocess.check_call(['gcc', '-std=c99', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'examples', 'c_abi_demo.c'), '-L', libdir, '-lbisip_hip',
                           f'-Wl,-rpath,{libdir}', '-Wl,-rpath,/opt/rocm/lib', '-lm', '-o', exe])
    out = subprocess.check_output([exe], text=True)
    vals = {line.split()[0]: float(line.split()[1]) for line in out.strip().splitlines()}
    # same problem through the oracle
    N = 8
    th0 = np.array([1.0, 0.3, -2.0, 0.5])
    w = 2 * 3.14159265358979323846 * 1000.0 / 4.0 ** np.arange(N)
    z = th0[0] * (1 - th0[1] * (1 - 1 / (1 + (1j * w * np.exp(th0[2])) ** th0[3])))
    zn = np.array([z.real, z.imag])
    zn_err = np.array([np.full(N, 0.01), np.full(N, 0.002)])
    bounds = np.array([[0.9, 0, -15, 0], [1.1, 1, 5, 1.0]])
    prob = oracle.OracleProblem('PeltonColeCole', w, zn, zn_err, bounds, n_modes=1)
    theta = np.array([[1.0, 0.3, -2.0, 0.5], [1.0, 0.3, -2.0, 0.6], [1.05, 0.2, -3.0, 0.4],
                      [1.2, 0.3, -2.0, 0.5], [1.0, 1.0, -2.0, 0.5]])
    want = oracle.logprob(prob, theta)
    got = np.array([vals[f'logp[{i}]'] for i in range(5)])
    assert_logp_close(got, want, 1e-9)     # the C demo builds its spectrum with its own libm calls
    assert np.isneginf(got[3]) and np.isneginf(got[4])
    assert abs(got[0] - vals['const']) < 1e-6


@pytest.mark.parametrize('script,expect', [
    ('quickstart.py', ['parameters', 'acceptance']),
    ('batch_of_spectra.py', ['spectrum   0', 'acceptance']),
    ('multi_gpu_logprob.py', ['1 rank(s): 1048576 log-probabilities']),
    ('multi_gpu_sampler.py', ['1 rank(s), driver sharded:', 'posterior mean']),
    ('multi_gpu_sampler.py rccl', ['1 rank(s), driver sharded-rccl', 'posterior mean']),
])
def test_python_examples_run(script, expect):
    """examples/*.py as a user would start them on one GPU (the multi-GPU ones with a single rank:
    the same code path -- RCCL communicator, C half-step loop -- as under torch.distributed.run)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29500 + os.getpid() % 2000))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    script, *argv = script.split()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', script), *argv], env=env, text=True,
                       capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for needle in expect:
        assert needle in r.stdout, r.stdout[-2000:]


def test_polydecomp_random_spectra_fuzz():
    """Random problems (frequencies, spectrum, errors over four decades, degree, exponent):
    every formulation against the oracle, on walkers clustered at the least-squares optimum
    -- where y - Z cancels hardest and the QR-reduced form has the least slack."""
    import oracle
    from bisip_amd import _hip
    rng = np.random.RandomState(20260101)
    worst = {}
    for trial in range(24):
        N = int(rng.randint(4, 48))
        P = int(rng.randint(1, 7))
        c_exp = float(rng.choice([1.0, 0.5, rng.uniform(0.2, 1.0)]))
        w = np.sort(2 * np.pi * 10 ** rng.uniform(-2, 4, N))[::-1].copy()
        per = np.log10(1. / w)
        lt = np.linspace(np.floor(per.min() - 1), np.floor(per.max() + 1), 2 * N)
        taus, log_taus = 10 ** lt, np.array([lt ** i for i in range(P + 1)])
        bounds = np.array([[0.5] + [-5.0] * (P + 1), [1.5] + [5.0] * (P + 1)])
        # a plausible spectrum: a random polynomial model + noise, errors over 4 decades
        th_true = np.r_[rng.uniform(0.9, 1.1), rng.randn(P + 1) * 10.0 ** -np.arange(2, P + 3)]
        err = 10 ** rng.uniform(-4.5, -1.0, (2, N))
        prob0 = oracle.OracleProblem('PolynomialDecomposition', w, np.zeros((2, N)), err, bounds,
                                     taus=taus, log_taus=log_taus, c_exp=c_exp)
        zn = oracle.forward(prob0, th_true[None, :])[0] + err * rng.randn(2, N)
        prob = oracle.OracleProblem('PolynomialDecomposition', w, zn, err, bounds, taus=taus,
                                    log_taus=log_taus, c_exp=c_exp)
        # least-squares centre through the linear structure (forward of unit vectors)
        base = oracle.forward(prob, np.r_[1.0, np.zeros(P + 1)][None, :])[0].ravel()
        cols = [base] + [oracle.forward(prob, np.r_[1.0, np.eye(P + 1)[p]][None, :])[0].ravel() - base
                         for p in range(P + 1)]
        A = np.array(cols).T / err.ravel()[:, None]
        b = np.linalg.lstsq(A, zn.ravel() / err.ravel(), rcond=None)[0]
        centre = np.r_[b[0], b[1:] / b[0]]
        theta = centre + 1e-5 * np.abs(centre) * rng.randn(200, P + 2)
        theta = np.vstack([theta, rng.uniform(bounds[0], bounds[1], (100, P + 2))])
        want = oracle.logprob(prob, theta, n_threads=4)
        for v in ('reduced', 'collapsed', 'faithful', 'wave'):
            ctx = _hip.HipContext(0, w, zn, err, bounds, poly_deg=P, c_exp=c_exp, taus=taus,
                                  log_taus=log_taus, variant=v)
            e = assert_logp_close(ctx.logprob(theta), want)
            worst[v] = max(worst.get(v, 0.0), e)
            ctx.close()
    print('worst relative error per formulation:', {k: f'{v:.2e}' for k, v in worst.items()})


# ----------------------------------------------------------------------------------
# the reference's compiled-function names on top of the HIP forward kernels
# (src/bisip/cython_funcs.pyx:49,64,75,96) -- one parameter vector per call
# ----------------------------------------------------------------------------------

@pytest.mark.parametrize('path', golden_cases()[::3], ids=case_id)
def test_cyth_named_functions_match_reference_golden(path):
    from bisip_amd import cython_funcs as cf
    g = np.load(path)
    model = case_model(path)
    w = g['w']
    for row in (0, 5, 41, 60):
        th = g['theta'][row]
        if not np.all(np.isfinite(th)):
            continue
        if model == 'PolynomialDecomposition':
            Z = cf.Decomp_cyth(w, g['taus'], g['log_taus'], float(g['c_exp']), R0=th[0], a=th[1:].copy())
        elif model == 'PeltonColeCole':
            D = int(g['n_modes'])
            Z = cf.ColeCole_cyth(w, th[0], th[1:1 + D].copy(), th[1 + D:1 + 2 * D].copy(),
                                 th[1 + 2 * D:].copy())
        elif model == 'Dias2000':
            Z = cf.Dias2000_cyth(w, *th)
        else:
            Z = cf.Shin2015_cyth(w, th[0:2].copy(), th[2:4].copy(), th[4:6].copy())
        assert Z.shape == (2, w.size) and Z.dtype == np.float64
        assert_Z_close(Z[None], g['Z'][row][None])


def test_bench_two_ranks_on_one_device():
    """`python bench.py --gpus 2` from a plain shell: the parent starts the two ranks itself; here
    both use cuda:0 over gloo (RCCL refuses two ranks on one device), each runs the real kernel on
    its own walkers, and the line reports 2 ranks seen, a per-rank kernel time and their sum rate."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo',
                        '--same-device', '--walkers', str(1 << 20), '--steps', '4', '--warmup', '1',
                        '--prime-seconds', '0.05'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['ranks_seen'] == 2 and rec['scaling'] == 'weak'
    assert rec['config']['global_walkers'] == 2 << 20 and rec['config']['kernel'] == 'k_logprob_pd_reduced'
    assert len(rec['roofline']['per_rank_kernel_ms']) == 2 and min(rec['roofline']['per_rank_kernel_ms']) > 0
    assert rec['value'] > 1e8 and 'cpu_baseline' not in rec       # rank-0-at-N=1 extras stay out
    # every rank checked its own shard against the oracle; the line carries the reduction
    par = rec['parity']
    assert par['ranks_checked'] == 2 and par['rows_per_rank'] == 4096 and par['neg_inf_rows_match'] is True
    assert par['max_rel_err_vs_oracle'] <= 1e-10
    # after the result line a SECOND group of two ranks ran BASELINE config 4's sharded stretch move (here
    # over gloo through host memory, each rank evaluating its half of every half-step with the
    # real kernels): every rank holds the same ensemble, and it is the single-GPU chain
    extra = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith('{"sampler_cfg4"')]
    assert len(extra) == 1, r.stderr[-3000:]
    sc = extra[0]['sampler_cfg4']
    assert sc['n_gpus'] == 2 and sc['walkers'] == 32768 and sc['driver'] == 'sharded'
    assert sc['state_identical_on_every_rank'] is True and sc['equals_single_gpu_fused_chain'] is True



@pytest.mark.parametrize('n_freq', [8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 19, 22, 27, 30, 32, 33, 47])
def test_grid_stepped_exponentials_agree_with_direct_ones_and_with_the_oracle(n_freq, monkeypatch):
    """On a geometric frequency grid (kernels.h: BOUNDS_GRID) ColeCole and Shin take one exponential per
    block of sixteen frequencies and term and step it by multiplication.  Same answer as one exponential per
    (frequency, term) -- the loop BISIP_NO_GRID=1 and every off-grid spectrum run -- to rounding, both
    within the parity tolerance of the oracle; and, as for every other loop, the same BITS whether one,
    two or four lanes evaluate a walker (launches of 100, 6,000 and 200,000 rows), for block tails of every
    kind (whole quarters, one to three frequencies after them, one or several blocks)."""
    import oracle
    from bisip_amd import _hip
    from bisip_amd.batch import default_params
    d = _synthetic_problem(n_freq, 2)
    assert _hip.frequency_grid_step(d['w']) is not None
    rng = np.random.RandomState(n_freq)
    for name, mid, kw in (('PeltonColeCole', 1, dict(n_modes=1)), ('PeltonColeCole', 1, dict(n_modes=2)),
                          ('PeltonColeCole', 1, dict(n_modes=3)), ('PeltonColeCole', 1, dict(n_modes=5)), ('Shin2015', 3, {})):
        bounds = np.array(list(default_params(name, **kw).values()), float).T
        theta = rng.uniform(bounds[0], bounds[1], (200000, bounds.shape[1]))
        got = {}
        for grid in (True, False):
            if grid:
                monkeypatch.delenv('BISIP_NO_GRID', raising=False)
            else:
                monkeypatch.setenv('BISIP_NO_GRID', '1')
            ctx = _hip.HipContext(mid, d['w'], d['zn'], d['zn_err'], bounds, **kw)
            # five modes: the steps' registers would spill in the persistent kernels; such models keep the direct loop
            assert ctx.loop_flags == (3 if grid and kw.get('n_modes', 2) <= 3 else 1)
            got[grid] = ctx.logprob(theta)
            for rows in (100, 6000):
                assert np.array_equal(ctx.logprob(theta[:rows]), got[grid][:rows]), (name, kw, rows, grid)
            ctx.close()
        monkeypatch.delenv('BISIP_NO_GRID', raising=False)
        prob = oracle.OracleProblem(name, d['w'], d['zn'], d['zn_err'], bounds, **kw)
        want = oracle.logprob(prob, theta[:3000], n_threads=4)
        scale = np.maximum(1.0, np.abs(want))
        assert np.max(np.abs(got[True][:3000] - want) / scale) <= 1e-10
        assert np.max(np.abs(got[False][:3000] - want) / scale) <= 1e-10
        assert np.max(np.abs(got[True] - got[False]) / np.maximum(1.0, np.abs(got[False]))) <= 2e-11


def test_grid_loop_needs_the_grid_and_the_default_box():
    """The stepped loop runs only where its premises hold: the bundled spectra (frequencies halved from
    6 kHz and then rounded in the files: 188.9 Hz for 187.5), a jittered grid and a 1-2-5 sequence keep one
    exponential per frequency (a series correction for rounded grids was built and measured: 1.1x in bulk
    launches, 0.7-0.9x on the ensembles of 32-256 walkers such spectra are fitted with -- not kept); so
    does a box widened past the shared-reciprocal limits, and set_bounds switches with the box.  A batch
    runs the stepped loop only if every spectrum is on a grid (each on its own)."""
    import bisip_amd
    from bisip_amd import _hip
    from bisip_amd.batch import default_params
    from bisip_amd.synthetic import synthetic_columns
    from bisip_amd.utils import load_data
    bounds = np.array(list(default_params('PeltonColeCole', n_modes=2).values()), float).T
    real = load_data(bisip_amd.DataFiles()['SIP-K389175'])
    ctx = _hip.HipContext(1, real['w'], real['zn'], real['zn_err'], bounds, n_modes=2)
    assert ctx.loop_flags == 1
    ctx.close()
    d = _synthetic_problem(32, 0)
    w = d['w'].copy()
    w[17] *= 1.0 + 3e-14
    ctx = _hip.HipContext(1, w, d['zn'], d['zn_err'], bounds, n_modes=2)
    assert ctx.loop_flags == 1
    ctx.close()
    w125 = 2 * np.pi * np.outer(10.0 ** np.arange(-2, 6), [1.0, 2.0, 5.0]).ravel()
    d125 = _synthetic_problem(24, 0)
    ctx = _hip.HipContext(1, w125, d125['zn'], d125['zn_err'], bounds, n_modes=2)
    assert ctx.loop_flags == 1
    ctx.close()
    ctx = _hip.HipContext(1, d['w'], d['zn'], d['zn_err'], bounds, n_modes=2)
    assert ctx.loop_flags == 3
    wide = bounds.copy()
    wide[1, 5:] = 30.0
    ctx.set_bounds(wide)
    assert ctx.loop_flags == 0
    ctx.set_bounds(bounds)
    assert ctx.loop_flags == 3
    ctx.close()
    # frequencies in either order (instruments write them descending: a negative step), and a sparse grid
    # (8 frequencies over 6 decades: steps of 2 in ln w): same value as in ascending order, to rounding
    import oracle
    rng = np.random.RandomState(12)
    theta = rng.uniform(bounds[0], bounds[1], (4000, 7))
    for dd in (d, _synthetic_problem(8, 1)):
        rev = dict(w=dd['w'][::-1].copy(), zn=dd['zn'][:, ::-1].copy(), zn_err=dd['zn_err'][:, ::-1].copy())
        assert _hip.frequency_grid_step(rev['w']) * _hip.frequency_grid_step(dd['w']) < 0     # one of them descends
        got = []
        for x in (dd, rev):
            ctx = _hip.HipContext(1, x['w'], x['zn'], x['zn_err'], bounds, n_modes=2)
            assert ctx.loop_flags == 3
            got.append(ctx.logprob(theta))
            ctx.close()
        want = oracle.logprob(oracle.OracleProblem('PeltonColeCole', rev['w'], rev['zn'], rev['zn_err'], bounds, n_modes=2), theta, n_threads=4)
        for g in got:
            assert np.max(np.abs(g - want) / np.maximum(1.0, np.abs(want))) <= 1e-10
    tables = [synthetic_columns(32, i) for i in range(3)]
    batch = bisip_amd.SpectraBatch('PeltonColeCole', tables, nwalkers=32, nsteps=2, n_modes=2)
    assert batch.ctx.loop_flags == 3
    # a batch that mixes spectra on a grid with one that is not: the loop is chosen per spectrum (its records
    # carry its own step, 0 off any grid), so every spectrum gets the bits of a context of its own -- for
    # whole-wave ensembles (64 rows per spectrum) and for ensembles that share waves (40 rows)
    tables[1] = tables[1].copy()
    tables[1][5, 0] *= 1.0 + 1e-9
    batch = bisip_amd.SpectraBatch('PeltonColeCole', tables, nwalkers=32, nsteps=2, n_modes=2)
    assert batch.ctx.loop_flags == 3
    for rows in (64, 40):
        th = rng.uniform(bounds[0], bounds[1], (3, rows, 7))
        got = batch.log_prob(th)
        for e in range(3):
            one = bisip_amd.SpectraBatch('PeltonColeCole', [tables[e]], nwalkers=32, nsteps=2, n_modes=2)
            assert one.ctx.loop_flags == (1 if e == 1 else 3)
            assert np.array_equal(one.log_prob(th[e:e + 1])[0], got[e]), (rows, e)
            one.close()
    batch.close()
