#!/usr/bin/env python3
"""get_model_percentile of ONE model over a 160,000-row chain (32 walkers x 5000 samples): the device call
(bisip_forward_percentiles: upload, forward, order statistics) against forward on the device + np.percentile on
the host.  The reference: a Python loop of 160,000 forward calls, then np.percentile (src/bisip/utils.py:17-35)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bisip_amd

m = bisip_amd.PolynomialDecomposition(bisip_amd.DataFiles()['SIP-K389175'], nwalkers=32, nsteps=10)
lo, hi = m.param_bounds
chain = np.random.RandomState(0).uniform(lo, hi, (160000, lo.size))
ctx = m._context()
for _ in range(2):
    dev = ctx.forward_percentiles(chain, [2.5, 50, 97.5])
t = time.perf_counter()
for _ in range(5):
    dev = ctx.forward_percentiles(chain, [2.5, 50, 97.5])
t_dev = (time.perf_counter() - t) / 5
t = time.perf_counter()
Z = m.forward(chain, m.data['w'])
t_fwd = time.perf_counter() - t
t = time.perf_counter()
host = np.percentile(Z, [2.5, 50, 97.5], axis=0)
t_pct = time.perf_counter() - t
print(f'device call {t_dev * 1e3:.2f} ms; forward on the device + np.percentile on the host {t_fwd * 1e3:.1f} + {t_pct * 1e3:.1f} ms; '
      f'same doubles: {bool(np.array_equal(dev, host))}, max rel diff {np.max(np.abs(dev - host) / np.maximum(1, np.abs(host))):.1e}')
