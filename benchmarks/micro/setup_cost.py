#!/usr/bin/env python3
"""Where DeviceEnsembleSampler.run_mcmc's setup time goes for a cfg5-sized batch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bisip_amd
from bisip_amd.synthetic import synthetic_columns
from bisip_amd.sampler import DeviceEnsembleSampler, affine_splits, walkers_independent

E, Wp = 512, 256
batch = bisip_amd.SpectraBatch('PeltonColeCole', [synthetic_columns(32, i) for i in range(E)], nwalkers=Wp, n_modes=2)
rng = np.random.RandomState(0)
p0 = (np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]) + 1e-3 * rng.randn(E, Wp, 7)).reshape(-1, 7)
s = DeviceEnsembleSampler(Wp, 7, batch.ctx, rng='philox', seed=3, n_ensembles=E)
s.run_mcmc(p0, 2)
be = s.backend


def lap(name, fn, reps=3):
    dts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    print(f'{name:34s} ' + ' '.join(f'{1e3 * dt:8.3f}' for dt in dts) + ' ms')
    return out


lap('np.array(copy)', lambda: np.array(p0, dtype=np.float64, copy=True))
lap('walkers_independent x32', lambda: [walkers_independent(p0[e * Wp:(e + 1) * Wp]) for e in range(32)])
lap('_check_coords', lambda: s._check_coords(p0))
c = lap('be.tensor(coords)', lambda: be.tensor(p0, torch.float64))
lap('be.zeros naccept', lambda: be.zeros((E * Wp,), torch.int32))
lp = lap('be.empty logp', lambda: be.empty((E * Wp,), torch.float64))
lap('logprob launch+sync', lambda: (be.logprob(c, lp), be.synchronize()))
lap('logp .cpu().numpy()', lambda: lp.cpu().numpy())
lap('affine_splits 1000', lambda: affine_splits(3, Wp, 0, 1000))
lap('be.tensor(perm)', lambda: be.tensor(affine_splits(3, Wp, 0, 1000)))
lap('_upload_state', lambda: s._upload_state(p0))
for k in range(3):
    s2 = DeviceEnsembleSampler(Wp, 7, batch.ctx, rng='philox', seed=3, n_ensembles=E, chain_on_device=True)
    t0 = time.perf_counter()
    s2.run_mcmc(p0, 100, thin_by=10)
    print('run', round(1e3 * (time.perf_counter() - t0), 2), 'ms', {k: round(1e3 * v, 2) for k, v in s2.timing.items()})

