#!/usr/bin/env python3
"""_upload_state inside the cfg5 benchmark's own sequence (priming runs, then the timed run)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bisip_amd
from bisip_amd.synthetic import synthetic_columns
from bisip_amd import sampler as S

E, Wp = 512, 256
batch = bisip_amd.SpectraBatch('PeltonColeCole', [synthetic_columns(32, i) for i in range(E)], nwalkers=Wp, nsteps=10, n_modes=2)
p0 = (np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]) + 1e-3 * np.random.RandomState(0).randn(E, Wp, 7))
batch.ctx.set_bounds(batch.param_bounds)
orig = S.DeviceEnsembleSampler._upload_state


def timed_upload(self, coords, lp=None):
    be = self.backend
    marks = [('start', time.perf_counter())]
    W = self.nwalkers
    self._accepted_before = np.asarray(self._accepted, dtype=np.float64).copy()
    marks.append(('accepted copy', time.perf_counter()))
    c = be.tensor(coords, torch.float64); marks.append(('be.tensor', time.perf_counter()))
    dev = dict(coords=c, naccept=be.zeros((W,), torch.int32), status=be.zeros((1,), torch.int32)); marks.append(('zeros', time.perf_counter()))
    dev['logp'] = be.empty((W,), torch.float64); marks.append(('empty', time.perf_counter()))
    be.logprob(dev['coords'], dev['logp']); marks.append(('logprob launch', time.perf_counter()))
    be.synchronize(); marks.append(('synchronize', time.perf_counter()))
    lp0 = dev['logp'].cpu().numpy(); marks.append(('logp to host', time.perf_counter()))
    assert not np.any(np.isnan(lp0)); marks.append(('isnan', time.perf_counter()))
    self._dev = dev
    print('   upload:', ', '.join(f'{n} {1e3*(t - marks[i][1]):.2f}' for i, (n, t) in enumerate(marks[1:])), 'ms', flush=True)


S.DeviceEnsembleSampler._upload_state = timed_upload


def make(chain):
    return S.DeviceEnsembleSampler(Wp, 7, batch.ctx, rng='philox', seed=3, n_ensembles=E, chain_on_device=chain)


for chain in (True, False):
    print('chain_on_device', chain)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        make(chain).run_mcmc(p0.reshape(-1, 7), 2, thin_by=10)
    for rep in range(3):
        s = make(chain)
        t0 = time.perf_counter()
        s.run_mcmc(p0.reshape(-1, 7), 100, thin_by=10)
        print('  run', rep, 'seconds %.4f' % (time.perf_counter() - t0), {k: round(v, 4) for k, v in s.timing.items()}, flush=True)
