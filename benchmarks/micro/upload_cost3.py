#!/usr/bin/env python3
"""Which upload path stalls? 15 timed cfg5 runs per (chain mode, upload method); prints the upload_s
of every run and the allocator counters around the slow ones."""
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
p0 = (np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]) + 1e-3 * np.random.RandomState(0).randn(E, Wp, 7)).reshape(-1, 7)
batch.ctx.set_bounds(batch.param_bounds)
pinned_tensor = S.HipStretchBackend.tensor


def pageable_tensor(self, array, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(array), dtype=dtype)
    return t.to(self.device)


def stats():
    m = torch.cuda.memory_stats()
    return m.get('num_device_alloc', 0), m.get('num_device_free', 0), m.get('num_alloc_retries', 0)


for method, fn in (('pinned-scratch', pinned_tensor), ('pageable', pageable_tensor)):
    S.HipStretchBackend.tensor = fn
    for chain in (True, False):
        ups, marks = [], []
        for rep in range(15):
            s = S.DeviceEnsembleSampler(Wp, 7, batch.ctx, rng='philox', seed=3, n_ensembles=E, chain_on_device=chain)
            a = stats()
            s.run_mcmc(p0, 100, thin_by=10)
            b = stats()
            ups.append(s.timing['upload_s'] * 1e3)
            marks.append((b[0] - a[0], b[1] - a[1], b[2] - a[2]))
        print(method, 'chain_on_device', chain, 'upload ms:', ' '.join('%.1f' % u for u in ups))
        print('    device alloc/free/retry per run:', marks, flush=True)
