"""BASELINE.json's full-size sampler configurations on one MI355X.

cfg4: Debye decomposition (PolynomialDecomposition c_exp = 1, 40 relaxation times = the bundled
20-frequency grid, poly_deg 5), 32768 walkers -- here on ONE GPU: the fused half-step, the
sharded half-step driven from Python (eval -> all_gather -> apply) and the sharded half-step
driven by the C loop over RCCL must give the same chain bit for bit, and the stored
log-probabilities must be the oracle's values of the stored positions.
cfg5: one GPU's share of the batch -- 512 spectra x 256 walkers, double Cole-Cole, 32 frequencies.
"""

import socket

import numpy as np
import pytest

from conftest import assert_logp_close

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.fixture
def one_rank_rccl_group():
    """torch.distributed over RCCL with a single rank (what every rank of an N-GPU run sees)."""
    import torch
    import torch.distributed as dist
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def test_cfg4_full_size_fused_python_sharded_and_rccl_loop_agree(one_rank_rccl_group):
    import bisip_amd
    import oracle
    from bisip_amd.sampler import DeviceEnsembleSampler
    W, ndim, nsteps = 32768, 7, 20
    m = bisip_amd.PolynomialDecomposition(bisip_amd.DataFiles()['SIP-K389175'], nwalkers=W, nsteps=nsteps)
    assert m.taus.size == 40 and m.data['N'] == 20 and m.poly_deg == 5 and m.c_exp == 1.0
    ctx = m._context()
    rng = np.random.RandomState(2024)
    centre = np.array([1.0, 0.005, -0.003, -0.001, 0.0005, 0.0002, 0.00001])
    p0 = centre + 1e-4 * rng.randn(W, ndim)

    def run(**kw):
        np.random.seed(7)
        s = DeviceEnsembleSampler(W, ndim, ctx, rng='philox', seed=11, persistent=False, **kw)
        s.run_mcmc(p0, nsteps)
        return s

    fused = run()
    assert fused.last_path == 'launch-per-half-step'
    chain, logp = fused.get_chain(), fused.get_log_prob()
    assert chain.shape == (nsteps, W, ndim)
    results = {
        'python loop': run(distributed=True, force_sharded_path=True, sharded_loop='python'),
        'C loop, torch communicator': run(distributed=True, force_sharded_path=True, sharded_loop='rccl'),
        'C loop, own communicator (group)': run(distributed=True, force_sharded_path=True, sharded_loop='rccl-own'),
        'C loop, own communicator (no group)': run(force_sharded_path=True, sharded_loop='rccl'),
    }
    want_path = {'python loop': 'sharded'}
    for name, s in results.items():
        assert s.last_path == want_path.get(name, 'sharded-rccl'), name
        assert np.array_equal(s.get_chain(), chain), name
        assert np.array_equal(s.get_log_prob(), logp), name
        assert np.array_equal(s.acceptance_fraction, fused.acceptance_fraction), name
        s.close()
    assert 0.2 < fused.acceptance_fraction.mean() < 0.8
    assert not np.array_equal(chain[0], chain[-1])
    # the stored log-probabilities are the oracle's log-probabilities of the stored positions
    d = m.data
    prob = oracle.OracleProblem('PolynomialDecomposition', d['w'], d['zn'], d['zn_err'], m.param_bounds,
                                taus=m.taus, log_taus=m.log_taus, c_exp=1.0)
    pick = rng.choice(W, 512, replace=False)
    for k in (0, nsteps - 1):
        assert_logp_close(logp[k, pick], oracle.logprob(prob, chain[k, pick]))


def test_cfg5_full_size_batch_of_spectra():
    import oracle
    from bisip_amd.batch import SpectraBatch
    from bisip_amd.synthetic import synthetic_columns
    E, Wp, nsteps = 512, 256, 10
    spectra = [synthetic_columns(32, i) for i in range(E)]
    b = SpectraBatch('ColeCole', spectra, nwalkers=Wp, nsteps=nsteps, n_modes=2)
    assert b.N == 32 and b.ndim == 7 and b.n_spectra == E
    truth = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6])
    rng = np.random.RandomState(5)
    p0 = truth + 1e-3 * rng.randn(E, Wp, 7)
    b.fit(p0=p0, seed=99, chain='device')
    mean_dev, std_dev = b.get_param_mean(), b.get_param_std()
    pct_dev = b.get_param_percentile([2.5, 50, 97.5])
    chain = b.get_chain()                       # (nsteps, E, Wp, ndim), copied from the device now
    logp = b.get_log_prob()
    assert chain.shape == (nsteps, E, Wp, 7) and logp.shape == (nsteps, E, Wp)
    flat = b.get_chain(flat=True)               # (E, nsteps*Wp, ndim)
    assert np.allclose(mean_dev, flat.mean(axis=1), rtol=1e-12, atol=1e-14)
    assert np.allclose(std_dev, flat.std(axis=1), rtol=1e-9, atol=1e-14)
    assert np.allclose(pct_dev, np.percentile(flat, [2.5, 50, 97.5], axis=1), rtol=1e-13, atol=1e-15)
    acc = b.acceptance_fraction
    assert acc.shape == (E, Wp) and 0.1 < acc.mean() < 0.9
    # eight spectra spread over the batch: stored log-probs = oracle log-probs of the stored positions
    for e in (0, 1, 63, 64, 255, 300, 510, 511):
        prob = oracle.OracleProblem('PeltonColeCole', b.w[e], b.zn[e], b.zn_err[e], b.param_bounds, n_modes=2)
        for k in (0, nsteps - 1):
            assert_logp_close(logp[k, e], oracle.logprob(prob, chain[k, e]))
    # and the batch log-probability entry on the final positions agrees with the chain's
    last = np.ascontiguousarray(chain[-1])
    assert np.array_equal(b.log_prob(last), logp[-1])


def test_headline_full_size_properties():
    """The bench metric's own shape and size -- PolynomialDecomposition P = 5, 32 frequencies, 64
    relaxation times, W = 2^24 walkers resident in HBM (BASELINE.json `metric`) -- through
    properties that need no oracle at that size, plus the oracle on a sample:
    -inf exactly on the rows outside the prior; walker index preserved under a permutation of
    the rows, bit for bit; a launch over any split of the batch gives the same bits; the
    per-frequency formulation agrees to 1e-11; 8192 rows spread over the batch match the oracle."""
    import glob
    import os
    import torch
    import oracle
    from conftest import GOLDEN
    from test_gpu_parity import make_ctx, _oracle_problem
    from bisip_amd.synthetic import synthetic_theta
    g = np.load(glob.glob(os.path.join(GOLDEN, 'case10_*'))[0])
    assert g['w'].size == 32 and int(g['poly_deg']) == 5 and g['taus'].size == 64
    W = 1 << 24
    lo, hi = g['bounds']
    theta = synthetic_theta(lo, hi, W)                     # the bench's rows (seed 2024)
    rng = np.random.RandomState(5)
    bad = rng.rand(W) < 0.01
    theta[bad, 3] = np.where(rng.rand(int(bad.sum())) < 0.5, hi[3], lo[3] - 1.0)   # on a bound / outside
    ctx = make_ctx(g, 'PolynomialDecomposition')
    assert ctx.variant == 'reduced'
    dev = torch.device('cuda', 0)
    th = torch.from_numpy(theta).to(dev)
    out = torch.empty(W, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ctx.logprob_dev(th.data_ptr(), W, out.data_ptr(), stream)
    torch.cuda.synchronize()
    a = out.cpu().numpy()
    assert np.array_equal(np.isneginf(a), bad) and not np.isnan(a).any()
    # permutation equivariance, on the device
    perm = torch.randperm(W, device=dev)
    th_p = th[perm].contiguous()
    out_p = torch.empty_like(out)
    ctx.logprob_dev(th_p.data_ptr(), W, out_p.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.equal(out_p, out[perm])
    del th_p, out_p, perm
    # any split of the batch: same bits (ragged pieces, one smaller than a workgroup)
    cuts = [0, 100, 5_000_001, 5_000_064, W]
    out_s = torch.empty_like(out)
    for lo_i, hi_i in zip(cuts[:-1], cuts[1:]):
        ctx.logprob_dev(th.data_ptr() + 8 * 7 * lo_i, hi_i - lo_i, out_s.data_ptr() + 8 * lo_i, stream)
    torch.cuda.synchronize()
    assert torch.equal(out_s, out)
    # the per-frequency formulation
    ctx.set_variant('collapsed')
    ctx.logprob_dev(th.data_ptr(), W, out_s.data_ptr(), stream)
    torch.cuda.synchronize()
    assert_logp_close(out_s.cpu().numpy(), a, 1e-11)
    # the oracle on rows spread over the whole batch
    pick = np.unique(np.concatenate([np.arange(0, W, W // 8000), [W - 1], np.flatnonzero(bad)[:64]]))
    want = oracle.logprob(_oracle_problem(g, 'PolynomialDecomposition'), theta[pick], n_threads=8)
    assert_logp_close(a[pick], want)
    ctx.close()


def test_bench_cfg5_extra_on_one_rank(one_rank_rccl_group, capsys):
    """bench.py's multi-GPU extras print a BASELINE config 5 record (every rank its own block of spectra, no
    collective on the data path).  The pool gives one GPU: the same function on a one-rank RCCL group."""
    import json
    import sys
    import torch
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    bench.batch_extra(one_rank_rccl_group, torch, 0, 1, 0, spectra_per_gpu=64, walkers=64, iterations=400)
    err = capsys.readouterr().err
    rec = json.loads([ln for ln in err.splitlines() if ln.startswith('{"sampler_cfg5"')][-1])['sampler_cfg5']
    assert rec['n_gpus'] == 1 and rec['spectra'] == 64 and rec['iterations'] == 400 and rec['path'] == 'persistent'
    assert rec['finite_means'] and rec['walker_steps_per_s'] > 1e7 and rec['collectives_on_the_data_path'] == 0


def test_fp64_stream_probe_is_the_ceiling_bench_holds_the_kernels_to():
    """bench.py's `fp64_fma_stream`: a stream of independent v_fma_f64 (bisip_fp64_stream_probe_dev) reaches most of the
    nominal issue peak and never more, its result depends on nothing but its arguments, and bad arguments are
    refused before anything is launched."""
    import sys
    import torch
    from conftest import ROOT
    from bisip_amd import _hip
    sys.path.insert(0, ROOT)
    import bench
    rec = bench.fp64_stream_ceiling(torch, torch.cuda.current_stream(), seconds=0.1)
    assert 0.6 < rec['frac_of_nominal_peak'] < 1.02, rec
    assert rec['wave_instr_per_s'] == pytest.approx(rec['frac_of_nominal_peak'] * bench.VALU_PEAK_WAVE_INSTR_S)
    lanes = _hip.fp64_stream_probe_lanes()
    a, b = (torch.zeros(lanes, dtype=torch.float64, device='cuda') for _ in range(2))
    n = _hip.fp64_stream_probe_dev(a.data_ptr(), 3)
    assert n == lanes // 64 * 32 * 3
    _hip.fp64_stream_probe_dev(b.data_ptr(), 3)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and bool(torch.isfinite(a).all()) and float(a.abs().min()) > 0.0
    for bad in (0, -1, (1 << 20) + 1):
        with pytest.raises(ValueError):
            _hip.fp64_stream_probe_dev(a.data_ptr(), bad)
    with pytest.raises(ValueError):
        _hip.fp64_stream_probe_dev(0, 3)
