#!/usr/bin/env python3
"""Latency of the host-buffer entry point for small batches (what an emcee-style driver
with vectorize=True pays per half-step)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'benchmarks'))
from sweep import problem
from bisip_amd.synthetic import synthetic_theta
for name, model, kw in [('PD reduced', 'pd', {}), ('CC D2', 'cc', dict(n_modes=2))]:
    ctx, bounds = problem(model, 20, **kw)
    for W in (16, 128, 2048, 32768):
        theta = synthetic_theta(bounds[0], bounds[1], W)
        for _ in range(20): ctx.logprob(theta)
        t0 = time.perf_counter()
        for _ in range(200): ctx.logprob(theta)
        dt = (time.perf_counter() - t0) / 200
        print(json.dumps({'case': name, 'W': W, 'us_per_call': round(dt * 1e6, 1)}))
