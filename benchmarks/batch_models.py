#!/usr/bin/env python3
"""The batch-of-spectra sampler for every model (cfg5's shape: 512 spectra x 256 walkers, 32 frequencies):
microseconds per half-step and walker-steps/s, persistent kernel and one launch per half-step.

    python benchmarks/batch_models.py [--spectra 512] [--walkers 256] [--iterations 8000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument('--spectra', type=int, default=512)
ap.add_argument('--walkers', type=int, default=256)
ap.add_argument('--iterations', type=int, default=8000)
ap.add_argument('--only', default=None, help='substring of the model name')
args = ap.parse_args()

import bisip_amd
from bisip_amd.sampler import DeviceEnsembleSampler
from bisip_amd.synthetic import synthetic_columns

E, Wp, thin = args.spectra, args.walkers, 40
stored = args.iterations // thin
tables = [synthetic_columns(32, i) for i in range(E)]
MODELS = (('PolynomialDecomposition', dict(poly_deg=5), [1.0, 0.005, -0.003, -0.001, 0.0005, 0.0002, 0.00001]),
          ('PeltonColeCole', dict(n_modes=1), [1.0, 0.5, -5.0, 0.5]),
          ('PeltonColeCole', dict(n_modes=2), [1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]),
          ('PeltonColeCole', dict(n_modes=3), [1.0, 0.15, 0.3, 0.2, -1.5, -6.0, -12.0, 0.45, 0.5, 0.6]),
          ('Dias2000', {}, [1.0, 0.5, -8.0, 10.0, 0.5]),
          ('Shin2015', {}, [0.5, 0.5, -14.0, -6.0, 0.5, 0.5]))
for model, kw, centre in MODELS:
    if args.only and args.only not in model:
        continue
    b = bisip_amd.SpectraBatch(model, tables, nwalkers=Wp, nsteps=stored, **kw)
    nd = len(centre)
    p0 = np.asarray(centre) + 1e-4 * np.random.RandomState(0).randn(E * Wp, nd)
    row = {'model': model + (f" D={kw['n_modes']}" if 'n_modes' in kw else ''), 'kernel': b.ctx.kernel_name,
           'spectra': E, 'walkers_per_spectrum': Wp, 'iterations': stored * thin,
           'spectra_plain_comp': list(b.ctx.reduced_tiers)}
    for name, persistent in (('persistent', True), ('launch_per_half_step', False)):
        best = None
        for rep in range(3):
            s = DeviceEnsembleSampler(Wp, nd, b.ctx, rng='philox', seed=3, n_ensembles=E, chain_on_device=True, persistent=persistent)
            t0 = time.perf_counter()
            s.run_mcmc(p0, stored, thin_by=thin)
            dt = time.perf_counter() - t0
            us = (s.timing['enqueue_s'] + s.timing['drain_s']) / (stored * thin) / 2 * 1e6
            if best is None or dt < best[0]:
                best = (dt, us, s.last_path, float(s.acceptance_fraction.mean()))
            del s
        row[name] = {'path': best[2], 'seconds': round(best[0], 4), 'us_per_half_step': round(best[1], 2),
                     'walker_steps_per_s': float('%.4g' % (stored * thin * E * Wp / best[0])), 'acceptance': round(best[3], 3)}
    print(json.dumps(row), flush=True)
    b.close()
