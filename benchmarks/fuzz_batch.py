#!/usr/bin/env python3
"""Randomised campaign for batches of spectra (bisip_batch_create): random model, number of
spectra, walkers per spectrum (multiples of 64 and not, so both the wave-uniform and the
per-lane record paths run), frequencies; log-prob and forward against the oracle per
spectrum, the E-ensemble device sampler (persistent kernel and launches) against a NumPy
replay of the stream contract around the same GPU log-probability, device-resident chain
moments against NumPy.

    python benchmarks/fuzz_batch.py --cases 120 --seed 1 > gpurun_out/fuzz_batch.jsonl
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=100)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    import oracle
    import bisip_amd
    from bisip_amd.sampler import DeviceEnsembleSampler
    from bisip_amd.synthetic import synthetic_columns
    from numpy_stretch_backend import NumpyStretchBackend

    rng = np.random.RandomState(args.seed)
    names = ['PolynomialDecomposition', 'PeltonColeCole', 'Dias2000', 'Shin2015']
    bad = 0
    t0 = time.time()
    for case in range(args.cases):
        model = names[rng.randint(4)]
        N = int(rng.choice([4, 16, 20, 27, 32]))
        E = int(rng.choice([2, 3, 7, 16, 33]))
        kw = {}
        if model == 'PolynomialDecomposition':
            kw = dict(poly_deg=int(rng.randint(0, 7)), c_exp=float(rng.choice([1.0, 0.5])))
        elif model == 'PeltonColeCole':
            kw = dict(n_modes=int(rng.randint(1, 4)))
        first = int(rng.randint(0, 300))
        tables = [synthetic_columns(N, first + i) for i in range(E)]
        if model in ('PeltonColeCole', 'Shin2015'):
            from fuzz_parity import frequencies_as_in_a_file
            # per spectrum: a batch mixes grids, rounded grids and irregular frequencies, and runs ONE loop
            tables = [frequencies_as_in_a_file(t, (first + i) * 131 + N + case) for i, t in enumerate(tables)]
        ndim = len(bisip_amd.batch.default_params(model, n_modes=kw.get('n_modes', 1), poly_deg=kw.get('poly_deg', 5)))
        Wp = int(rng.choice([2 * ndim + (2 * ndim) % 2, 30, 64, 100, 128, 256]))
        Wp += Wp % 2
        nsteps = int(rng.randint(2, 12))
        thin = int(rng.choice([1, 2]))
        batch = bisip_amd.SpectraBatch(model, tables, nwalkers=Wp, nsteps=nsteps, **kw)
        lo, hi = batch.param_bounds
        rec = dict(case=case, model=model, N=N, E=E, Wp=Wp, nsteps=nsteps, thin=thin, **kw)
        problems = []
        okw = {}
        if model == 'PolynomialDecomposition':
            okw = dict(taus=batch.taus, log_taus=batch.log_taus, c_exp=batch.c_exp)
        if model == 'PeltonColeCole':
            okw = dict(n_modes=batch.n_modes)
        probs = [oracle.OracleProblem(batch.model, batch.w[e], batch.zn[e], batch.zn_err[e], batch.param_bounds, **okw)
                 for e in range(E)]
        # log-prob / forward, n rows per spectrum (n random: whole blocks or ragged)
        n = int(rng.choice([5, 64, 70, 128]))
        theta = rng.uniform(lo, hi, (E, n, ndim))
        theta[rng.randint(E), rng.randint(n), rng.randint(ndim)] = hi[0] + 5.0
        got = batch.log_prob(theta)
        want = np.array([oracle.logprob(probs[e], theta[e]) for e in range(E)])
        fin = np.isfinite(want)
        if not np.array_equal(np.isneginf(got), np.isneginf(want)):
            problems.append('-inf pattern')
        err = float(np.max(np.abs(got[fin] - want[fin]) / np.maximum(1, np.abs(want[fin]))))
        if err > 1e-10:
            problems.append(f'log-prob off by {err:.2e}')
        Z = batch.forward(theta)
        Zw = np.array([oracle.forward(probs[e], theta[e]) for e in range(E)])
        ez = float(np.max(np.abs(Z - Zw)) / max(1.0, float(np.max(np.abs(Zw)))))
        if ez > 1e-12:
            problems.append(f'forward off by {ez:.2e}')
        rec['logp_err'], rec['Z_err'] = float('%.3g' % err), float('%.3g' % ez)
        # sampler: persistent / launches / NumPy replay of the contract
        mid = 0.5 * (lo + hi)
        p0 = mid + 0.1 * (hi - lo) * (rng.rand(E, Wp, ndim) - 0.5)
        seed = int(rng.randint(1, 2 ** 31 - 1))
        batch.ctx.set_bounds(batch.param_bounds)
        chains = []
        for pers in (True, False):
            s = DeviceEnsembleSampler(Wp, ndim, batch.ctx, rng='philox', seed=seed, n_ensembles=E, persistent=pers,
                                      chain_on_device=pers, live_dangerously=True)
            s.run_mcmc(p0.reshape(E * Wp, ndim), nsteps, thin_by=thin)
            if pers:
                mean, std = s.param_moments(discard=nsteps // 3)
            chains.append((s.get_chain(), s.get_log_prob(), s.last_path))
        if chains[0][2] != 'persistent' and Wp * (ndim + 1) * 8 <= 65536:
            problems.append('persistent path not taken')
        if not (np.array_equal(chains[0][0], chains[1][0]) and np.array_equal(chains[0][1], chains[1][1])):
            problems.append('persistent != launches')
        rep = DeviceEnsembleSampler(Wp, ndim, backend=NumpyStretchBackend(batch.ctx.logprob, E), rng='philox', seed=seed,
                                    n_ensembles=E, live_dangerously=True)
        rep.run_mcmc(p0.reshape(E * Wp, ndim), nsteps, thin_by=thin)
        if not np.array_equal(rep.get_chain(), chains[1][0]):
            problems.append('NumPy replay differs')
        flat = chains[1][0][nsteps // 3:].reshape(-1, E, Wp, ndim).transpose(1, 0, 2, 3).reshape(E, -1, ndim)
        em = float(np.max(np.abs(mean - flat.mean(1)) / np.maximum(1, np.abs(flat.mean(1)))))
        es = float(np.max(np.abs(std - flat.std(1)) / np.maximum(1, np.abs(flat.std(1)))))
        if em > 1e-12 or es > 1e-12:
            problems.append(f'moments off: mean {em:.2e} std {es:.2e}')
        if problems:
            bad += 1
            rec['problems'] = problems
        print(json.dumps(rec), flush=True)
        batch.ctx.close()
    print(json.dumps(dict(summary=True, cases=args.cases, seed=args.seed, failures=bad, seconds=round(time.time() - t0, 1))))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
