#!/usr/bin/env python3
"""Latency of bisip_logprob with host buffers for emcee-sized batches (what `vectorize=True` callers and the
host-loop sampler pay per half-step): launch + kernel + completion, ~17 us up to a few hundred rows."""
import os
import sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bisip_amd
m = bisip_amd.PolynomialDecomposition(bisip_amd.DataFiles()['SIP-K389175'], nwalkers=32, nsteps=10)
ctx = m._context()
lo, hi = m.param_bounds
for W in (16, 64, 256, 4096):
    th = np.random.RandomState(0).uniform(lo, hi, (W, 7))
    for _ in range(200): ctx.logprob(th)
    t = time.perf_counter()
    for _ in range(2000): ctx.logprob(th)
    print(W, 'rows:', round((time.perf_counter() - t) / 2000 * 1e6, 2), 'us per call (host buffers in and out)')
