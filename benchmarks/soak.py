#!/usr/bin/env python3
"""Long runs, for what short tests cannot show: device / host memory that grows from run to run, a drifting
posterior, a stalled worker thread.  Twenty fits of 100,000 iterations each (NumPy-order stream, many chunks,
the prefetch thread at work), then twenty batch fits with the Philox stream drawn beside the kernels."""
import gc
import json
import os
import resource
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bisip_amd
from bisip_amd.synthetic import synthetic_columns


def rss_mb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


out = {}
m = bisip_amd.PolynomialDecomposition(bisip_amd.DataFiles()['SIP-K389175'], nwalkers=32, nsteps=2000, poly_deg=4)
means, mem, host = [], [], []
t0 = time.perf_counter()
for rep in range(20):
    np.random.seed(rep)
    m.fit(thin_by=50)                       # 100,000 iterations, 2000 stored
    means.append(m.get_param_mean(m.get_chain(discard=500, flat=True)))
    gc.collect()
    mem.append(torch.cuda.memory_allocated())
    host.append(rss_mb())
means = np.array(means)
out['single'] = {'fits': 20, 'iterations_each': 100000, 'seconds': round(time.perf_counter() - t0, 2),
                 'device_bytes_after_fit_2_and_20': [mem[1], mem[-1]], 'host_max_rss_mb_after_fit_2_and_20': [round(host[1], 1), round(host[-1], 1)],
                 'posterior_mean_r0_min_max': [float(means[:, 0].min()), float(means[:, 0].max())],
                 'spread_over_fits_in_posterior_sigmas': float(np.max(np.ptp(means, axis=0) / m.get_param_std(m.get_chain(discard=500, flat=True))))}
E_, Wp = 256, 128
b = bisip_amd.SpectraBatch('PeltonColeCole', [synthetic_columns(32, i) for i in range(E_)], nwalkers=Wp, nsteps=250, n_modes=2)
p0 = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]) + 1e-3 * np.random.RandomState(0).randn(E_, Wp, 7)
mem, host, first = [], [], None
t0 = time.perf_counter()
for rep in range(20):
    b.fit(p0, seed=5, thin_by=40, chain='device')      # 10,000 iterations, many chunks
    mean = b.get_param_mean(discard=100)
    first = mean if first is None else first
    assert np.array_equal(mean, first)                 # same seed: the same chain every time
    gc.collect()
    mem.append(torch.cuda.memory_allocated())
    host.append(rss_mb())
out['batch'] = {'fits': 20, 'iterations_each': 10000, 'seconds': round(time.perf_counter() - t0, 2), 'same_summaries_every_fit': True,
                'device_bytes_after_fit_2_and_20': [mem[1], mem[-1]], 'host_max_rss_mb_after_fit_2_and_20': [round(host[1], 1), round(host[-1], 1)]}
print(json.dumps(out))
