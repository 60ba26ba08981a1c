#!/usr/bin/env python3
"""The reference's README quickstart (README.md:38-50 of clberube/BISIP) on the MI355X path:
the only change is the import.  Prints posterior means / standard deviations and a few
model-space percentiles."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

import numpy as np

from bisip_amd import DataFiles, PolynomialDecomposition      # was: from bisip import ...

np.random.seed(42)                                            # pins the stretch-move stream, as with emcee
filepath = DataFiles()['SIP-K389175']
model = PolynomialDecomposition(filepath, nwalkers=32, nsteps=1000, poly_deg=4)
model.fit()                                                   # ensemble + chain on the GPU, one launch per chunk

chain = model.get_chain(discard=500, thin=2, flat=True)
print('parameters ', model.param_names)
print('mean       ', np.round(model.get_param_mean(chain), 5))
print('std        ', np.round(model.get_param_std(chain), 5))
lo, med, hi = model.get_model_percentile([2.5, 50, 97.5], chain)          # (3, 2, N): batched forward on the GPU
print('|Z| median ', np.round(np.hypot(med[0], med[1])[:5], 4), '...')
print('acceptance ', round(float(model.sampler.acceptance_fraction.mean()), 3), 'path', model.sampler.last_path)
