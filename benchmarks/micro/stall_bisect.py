#!/usr/bin/env python3
"""Which ingredient of a sampler run brings the post-run stall?  ~0.1 s of device work of one kind, a
synchronisation, optional device->host copies, then a small torch kernel timed from launch to the end of
the synchronisation; 40 rounds per recipe.
    work:  logprob = bisip_logprob_dev launches;  torch = torch elementwise kernels;
           persist = the persistent sampler kernel (cfg5 slice);  half = one launch per half-step
    after: none | cpu (four .cpu() copies, as at the end of run_mcmc) | alloc (a fresh 1 GB tensor, dropped)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bisip_amd
from bisip_amd.sampler import DeviceEnsembleSampler
from bisip_amd.synthetic import synthetic_columns

E, Wp = 512, 256
batch = bisip_amd.SpectraBatch('PeltonColeCole', [synthetic_columns(32, i) for i in range(E)], nwalkers=Wp, nsteps=10, n_modes=2)
ctx = batch.ctx
p0 = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]) + 1e-3 * np.random.RandomState(0).randn(E * Wp, 7)
theta = torch.from_numpy(p0).cuda()
out = torch.empty(E * Wp, dtype=torch.float64, device='cuda')
x = torch.zeros(1 << 20, dtype=torch.float64, device='cuda')
big = torch.zeros(1 << 26, dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream


def work(kind):
    if kind == 'logprob':
        for _ in range(1500):
            ctx.logprob_dev(theta.data_ptr(), E * Wp, out.data_ptr(), st)
    elif kind == 'torch':
        for _ in range(300):
            big.add_(1.0)
    else:
        s = DeviceEnsembleSampler(Wp, 7, ctx, rng='philox', seed=3, n_ensembles=E, chain_on_device=True,
                                  persistent=(kind == 'persist'))
        s.run_mcmc(p0, int(os.environ.get('STORED', 150)), thin_by=int(os.environ.get('THIN', 40)))
        return s
    return None


for kind, after in [a.split(':') for a in sys.argv[1:]] or [('logprob', 'cpu')]:
    slow, worst, tot = 0, 0.0, 0.0
    rounds = int(os.environ.get('ROUNDS', 40))
    for rep in range(rounds):
        t0 = time.perf_counter()
        keep = work(kind)
        torch.cuda.synchronize()
        tot += time.perf_counter() - t0
        if after == 'cpu':
            _ = out[:1].cpu(); _ = out[:131072].int().cpu(); _ = theta.cpu(); _ = out.cpu()
        elif after == 'alloc':
            tmp = torch.empty(1 << 27, dtype=torch.float64, device='cuda'); del tmp
        t1 = time.perf_counter()
        x.add_(1.0)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) * 1e3
        worst = max(worst, ms)
        slow += ms > 2.0
        del keep
    print(f'work {kind:8s} ({tot / rounds * 1e3:6.1f} ms per round), then {after:5s}: small kernel launch->sync worst {worst:7.3f} ms, {slow:2d} of {rounds} above 2 ms', flush=True)
