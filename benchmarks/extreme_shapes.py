#!/usr/bin/env python3
"""Frequency counts far beyond the reference's 20 (up to the library's limit of 4096): log-probability
and forward of every model / formulation against the oracle, and a short device sampler run against the
host loop -- the shapes where the kernels switch paths (records staged in LDS or not, tiled / whole-row
forward, register blocks).  Exits non-zero on the first violation.

    python benchmarks/extreme_shapes.py [--sizes 100,257,1024,4096]
"""
import argparse
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from bisip_amd import _hip
from bisip_amd.batch import default_params
from bisip_amd.synthetic import synthetic_columns
from bisip_amd.utils import columns_to_data
from bisip_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
NAMES = {'PolynomialDecomposition':0,'PeltonColeCole':1,'Dias2000':2,'Shin2015':3}
ap = argparse.ArgumentParser()
ap.add_argument('--sizes', default='100,257,1024,4096')
args = ap.parse_args()
rng = np.random.RandomState(5)
worst = 0
for N in [int(x) for x in args.sizes.split(',')]:
    d = columns_to_data(synthetic_columns(N, 3), 'mrad')
    for model in NAMES:
        kw, okw = {}, {}
        if model == 'PolynomialDecomposition':
            P = 5; per = np.log10(1./d['w']); lt = np.linspace(np.floor(per.min()-1), np.floor(per.max()+1), min(2*N, 512))
            kw = dict(poly_deg=P, c_exp=1.0, taus=10**lt, log_taus=np.array([lt**i for i in range(P+1)])); okw = dict(taus=kw['taus'], log_taus=kw['log_taus'], c_exp=1.0)
            params = default_params(model, poly_deg=P)
        elif model == 'PeltonColeCole':
            kw = okw = dict(n_modes=2); params = default_params(model, n_modes=2)
        else:
            params = default_params(model)
        bounds = np.array(list(params.values()), float).T
        theta = rng.uniform(bounds[0], bounds[1], (333, bounds.shape[1]))
        if model == 'PolynomialDecomposition': theta[:, 1:] *= 1e-3
        prob = oracle.OracleProblem(model, d['w'], d['zn'], d['zn_err'], bounds, **okw)
        want = oracle.logprob(prob, theta, n_threads=8)
        for variant in (['auto', 'collapsed', 'reduced_comp'] if model == 'PolynomialDecomposition' else ['auto']):
            ctx = _hip.HipContext(NAMES[model], d['w'], d['zn'], d['zn_err'], bounds, variant=variant, **kw)
            got = ctx.logprob(theta)
            e = float(np.max(np.abs(got-want)/np.maximum(1, np.abs(want))))
            Z = ctx.forward(theta[:70]); Zw = oracle.forward(prob, theta[:70])
            ez = float(np.max(np.abs(Z-Zw))/max(1.0, float(np.max(np.abs(Zw)))))
            worst = max(worst, e)
            # a short sampler run: device == host loop
            W = 2*bounds.shape[1]+6
            p0 = theta[:W]
            np.random.seed(1); h = EnsembleSampler(W, bounds.shape[1], ctx.logprob); h.run_mcmc(p0, 6)
            np.random.seed(1); dv = DeviceEnsembleSampler(W, bounds.shape[1], ctx); dv.run_mcmc(p0, 6)
            same = np.array_equal(h.get_chain(), dv.get_chain())
            print(f'N={N:5d} {model:24s} {variant:12s} {ctx.kernel_name:28s} logp err {e:.1e}  Z err {ez:.1e}  sampler {"same" if same else "DIFFERENT"} ({dv.last_path})', flush=True)
            assert e <= 1e-10 and ez <= 1e-12 and same
            ctx.close()
print('{"summary": true, "worst_logp_rel_err": %.3g}' % worst)
