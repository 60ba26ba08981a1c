#!/usr/bin/env python3
"""A survey of many spectra inverted together (BASELINE config 5 in miniature): E synthetic
double-Cole-Cole spectra x 128 walkers each, all ensembles advanced by the same launches, the
chain kept in HBM and summarised there (mean / std / percentiles per spectrum)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

import numpy as np

import bisip_amd
from bisip_amd.synthetic import synthetic_columns

E, Wp = 64, 128
tables = [synthetic_columns(32, i) for i in range(E)]          # or a list of data-file paths
batch = bisip_amd.SpectraBatch('PeltonColeCole', tables, nwalkers=Wp, nsteps=200, n_modes=2)
rng = np.random.RandomState(0)
p0 = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]) + 1e-3 * rng.randn(E, Wp, 7)
batch.fit(p0, seed=1, thin_by=5, chain='device')               # 1000 iterations, 200 stored

mean = batch.get_param_mean(discard=100)                       # (E, ndim), computed on the device
std = batch.get_param_std(discard=100)
p16, p50, p84 = batch.get_param_percentile([16, 50, 84], discard=100)
print('parameters', batch.param_names)
for e in (0, 1, E - 1):
    print(f'spectrum {e:3d}  mean {np.round(mean[e], 3)}  median {np.round(p50[e], 3)}')
print('acceptance', round(float(batch.acceptance_fraction.mean()), 3))
