#!/usr/bin/env python3
"""After a long run (2000 half-step launches) returns from synchronize(): how long until the GPU
executes the NEXT tiny piece of work, and is the delay on the host or on the device side?"""
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
x = torch.zeros(1024, device='cuda')
mode = sys.argv[1] if len(sys.argv) > 1 else 'kernel'
for rep in range(12):
    s = S.DeviceEnsembleSampler(Wp, 7, batch.ctx, rng='philox', seed=3, n_ensembles=E, chain_on_device=True)
    s.run_mcmc(p0, 100 if rep % 2 == 0 else 2, thin_by=10)      # long run, then a short one, alternating
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == 'kernel':
        x.add_(1.0)                                   # one tiny kernel
    elif mode == 'h2d':
        y = torch.ones(1024).to('cuda')               # one tiny blocking copy
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{mode} after {"LONG" if rep % 2 == 0 else "short"} run: host call {1e3*(t1-t0):7.3f} ms, until done {1e3*(t2-t0):7.3f} ms', flush=True)
