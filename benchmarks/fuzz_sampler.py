#!/usr/bin/env python3
"""Randomised campaign for the device-resident stretch move: random model, ensemble size (odd
sizes, the smallest allowed, several hundred), iteration count, thinning, chunking and stream
mode; the chain of the persistent kernel, of the launch-per-half-step path and of the host
sampler around the same GPU log-probability must be IDENTICAL (bit for bit), and every stored
log-probability must equal the oracle's value of the stored position (tolerance 1e-10).

    python benchmarks/fuzz_sampler.py --cases 150 --seed 1 > gpurun_out/fuzz_sampler.jsonl
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'benchmarks'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=100)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    import oracle
    from bisip_amd import _hip
    from bisip_amd.batch import default_params
    from bisip_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
    from bisip_amd.synthetic import synthetic_columns
    from bisip_amd.utils import columns_to_data
    from numpy_stretch_backend import NumpyStretchBackend

    rng = np.random.RandomState(args.seed)
    names = ['PolynomialDecomposition', 'PeltonColeCole', 'Dias2000', 'Shin2015']
    bad = 0
    t0 = time.time()
    for case in range(args.cases):
        model = names[rng.randint(4)]
        N = int(rng.choice([3, 16, 20, 21, 32, 45]))
        idx = int(rng.randint(0, 500))
        cols = synthetic_columns(N, idx)
        if model in ('PeltonColeCole', 'Shin2015'):
            from fuzz_parity import frequencies_as_in_a_file
            cols = frequencies_as_in_a_file(cols, idx * 131 + N)
        d = columns_to_data(cols, 'mrad')
        kw, okw, variant = {}, {}, 'auto'
        if model == 'PolynomialDecomposition':
            P = int(rng.randint(0, 8))
            per = np.log10(1. / d['w'])
            lt = np.linspace(np.floor(per.min() - 1), np.floor(per.max() + 1), 2 * N)
            kw = dict(poly_deg=P, c_exp=float(rng.choice([1.0, 0.5])), taus=10 ** lt, log_taus=np.array([lt ** i for i in range(P + 1)]))
            okw = dict(taus=kw['taus'], log_taus=kw['log_taus'], c_exp=kw['c_exp'])
            variant = str(rng.choice(['reduced', 'collapsed']))
            params = default_params(model, poly_deg=P)
        elif model == 'PeltonColeCole':
            D = int(rng.randint(1, 4))
            kw = okw = dict(n_modes=D)
            params = default_params(model, n_modes=D)
        else:
            params = default_params(model)
        bounds = np.array(list(params.values()), float).T
        ndim = bounds.shape[1]
        W = int(rng.choice([2 * ndim, 2 * ndim + 1, 32, 33, 64, 100, 129, 256, 301, 700]))
        nsteps = int(rng.randint(1, 25))
        thin = int(rng.choice([1, 1, 2, 3]))
        chunk = None if rng.rand() < 0.5 else int(rng.randint(1, 9))
        mode = str(rng.choice(['numpy', 'philox']))
        seed = int(rng.randint(1, 2 ** 31 - 1))
        mid = 0.5 * (bounds[0] + bounds[1])
        p0 = mid + 0.2 * (bounds[1] - bounds[0]) * (rng.rand(W, ndim) - 0.5)
        if rng.rand() < 0.3:
            p0[rng.randint(W)] = bounds[1] + 1.0          # a walker that starts outside the prior
        ctx = _hip.HipContext(names.index(model), d['w'], d['zn'], d['zn_err'], bounds, variant=variant, **kw)
        rec = dict(case=case, model=model, N=N, W=W, ndim=ndim, nsteps=nsteps, thin=thin, chunk=chunk, rng=mode, variant=variant)
        chains = {}
        for name, pers in (('persistent', True), ('launches', False)):
            np.random.seed(seed % (2 ** 31))
            s = DeviceEnsembleSampler(W, ndim, ctx, rng=mode, seed=seed, chunk=chunk, persistent=pers, live_dangerously=True)
            s.run_mcmc(p0, nsteps, thin_by=thin)
            s.run_mcmc(None, 2, thin_by=thin)
            chains[name] = (s.get_chain(), s.get_log_prob(), s.acceptance_fraction, s.last_path)
        fits = W * (ndim + 1) * 8 <= 65536 and (W + 1) // 2 <= 1024
        problems = []
        if fits and chains['persistent'][3] != 'persistent':
            problems.append('persistent path not taken')
        for k in range(3):
            if not np.array_equal(chains['persistent'][k], chains['launches'][k]):
                problems.append(f'persistent != launches (item {k})')
        # the same stream driven from the host around the GPU log-probability
        np.random.seed(seed % (2 ** 31))
        if mode == 'numpy':
            h = EnsembleSampler(W, ndim, ctx.logprob, live_dangerously=True)
        else:
            h = DeviceEnsembleSampler(W, ndim, backend=NumpyStretchBackend(ctx.logprob), rng='philox', seed=seed,
                                      chunk=chunk, live_dangerously=True)
        h.run_mcmc(p0, nsteps, thin_by=thin)
        h.run_mcmc(None, 2, thin_by=thin)
        if not np.array_equal(h.get_chain(), chains['launches'][0]):
            problems.append('host-driven chain differs')
        if not np.array_equal(h.get_log_prob(), chains['launches'][1]):
            problems.append('host-driven log-prob differs')
        # stored log-probs are the oracle's value of the stored positions
        ch, lp = chains['launches'][0], chains['launches'][1]
        prob = oracle.OracleProblem(model, d['w'], d['zn'], d['zn_err'], bounds, **okw)
        want = oracle.logprob(prob, ch.reshape(-1, ndim), n_threads=8).reshape(lp.shape)
        fin = np.isfinite(want)
        if not np.array_equal(np.isneginf(lp), np.isneginf(want)) or np.any(np.isnan(lp)):
            problems.append('-inf pattern differs from the oracle')
        err = float(np.max(np.abs(lp[fin] - want[fin]) / np.maximum(1, np.abs(want[fin])))) if fin.any() else 0.0
        if err > 1e-10:
            problems.append(f'stored log-prob off by {err:.2e}')
        rec['oracle_err'] = float('%.3g' % err)
        rec['acceptance'] = round(float(chains['launches'][2].mean()), 3)
        if problems:
            bad += 1
            rec['problems'] = problems
        print(json.dumps(rec), flush=True)
        ctx.close()
    print(json.dumps(dict(summary=True, cases=args.cases, seed=args.seed, failures=bad, seconds=round(time.time() - t0, 1))))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
