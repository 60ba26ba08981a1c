#!/usr/bin/env python3
"""Discriminate the intermittent 20-35 ms stall of the first host->device copy of a run."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bisip_amd
from bisip_amd.synthetic import synthetic_columns
from bisip_amd import sampler as S

mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'
E, Wp = 512, 256
batch = bisip_amd.SpectraBatch('PeltonColeCole', [synthetic_columns(32, i) for i in range(E)], nwalkers=Wp, nsteps=10, n_modes=2)
p0 = (np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]) + 1e-3 * np.random.RandomState(0).randn(E, Wp, 7)).reshape(-1, 7)
batch.ctx.set_bounds(batch.param_bounds)
ups = []
for rep in range(20):
    s = S.DeviceEnsembleSampler(Wp, 7, batch.ctx, rng='philox', seed=3, n_ensembles=E, chain_on_device=True)
    if mode == 'sleep':
        time.sleep(0.05)
    if mode == 'short':
        s.run_mcmc(p0, 2, thin_by=10)
    else:
        s.run_mcmc(p0, 100, thin_by=10)
    ups.append(s.timing['upload_s'] * 1e3)
print(mode, os.environ.get('HSA_ENABLE_SDMA', '-'), 'upload ms:', ' '.join('%.1f' % u for u in ups), flush=True)
