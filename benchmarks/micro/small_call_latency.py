#!/usr/bin/env python3
"""Latency of one log-probability call with host buffers for emcee-sized batches -- what a
`vectorize=True` emcee run (fit(moves=...)) and the host-loop sampler pay per half-step, and what a user
calling model.log_prob() sees.  Two layers: ctx.logprob (ctypes -> bisip_logprob: staging through pinned /
mapped memory, one launch, one synchronisation, and for PolynomialDecomposition the guard of the reduced
kernel on calls 1, 2, 4, 8, ...) and model.log_prob (the reference-shaped Python method on top of it).
One JSON line per model."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bisip_amd

path = bisip_amd.DataFiles()['SIP-K389175']
for cls, kw in ((bisip_amd.PolynomialDecomposition, {}), (bisip_amd.PeltonColeCole, dict(n_modes=2)),
                (bisip_amd.Dias2000, {}), (bisip_amd.Shin2015, {})):
    m = cls(path, nwalkers=32, nsteps=10, **kw)
    ctx = m._context()
    lo, hi = m.param_bounds
    rec = {'model': cls.__name__, 'kernel': ctx.kernel_name, 'us_per_call': {}}
    for W in (16, 32, 64, 256, 4096):
        th = np.random.RandomState(0).uniform(lo, hi, (W, lo.size))
        out = {}
        for name, fn in (('ctx.logprob', ctx.logprob), ('model.log_prob', m.log_prob)):
            for _ in range(300):
                fn(th)
            best = None
            for _ in range(3):
                t = time.perf_counter()
                for _ in range(2000):
                    fn(th)
                dt = (time.perf_counter() - t) / 2000 * 1e6
                best = dt if best is None or dt < best else best
            out[name] = round(best, 2)
        rec['us_per_call'][W] = out
    print(json.dumps(rec), flush=True)
