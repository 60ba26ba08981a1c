#!/usr/bin/env python3
"""Where does DeviceEnsembleSampler._upload_state spend its time at the cfg5 shape
(512 x 256 walkers x 7 parameters = 7.3 MB of start positions)?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bisip_amd
from bisip_amd.synthetic import synthetic_columns
from bisip_amd.sampler import DeviceEnsembleSampler, HipStretchBackend, _pinned_scratch

E, Wp = 512, 256
batch = bisip_amd.SpectraBatch('PeltonColeCole', [synthetic_columns(32, i) for i in range(E)], nwalkers=Wp, nsteps=10, n_modes=2)
p0 = (np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]) + 1e-3 * np.random.RandomState(0).randn(E * Wp, 7))
be = HipStretchBackend(batch.ctx)
dev = be.device


def t(label, fn, n=5):
    torch.cuda.synchronize()
    best = []
    for _ in range(n):
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); best.append(time.perf_counter() - t0)
    print(f'{label:55s} min {min(best)*1e3:7.3f} ms  median {sorted(best)[len(best)//2]*1e3:7.3f} ms', flush=True)
    return out

t('be.tensor (pinned scratch path)', lambda: be.tensor(p0, torch.float64))
t('torch.from_numpy(p0).to(dev)  (pageable)', lambda: torch.from_numpy(p0).to(dev))
pin = _pinned_scratch('upload', p0.nbytes)[:p0.nbytes].view(torch.float64).view(p0.shape)
src = torch.from_numpy(p0)
t('  pinned.copy_(t) alone', lambda: pin.copy_(src))
t('  np.copyto into pinned alone', lambda: np.copyto(pin.numpy(), p0))
t('  pinned.to(dev, non_blocking)', lambda: pin.to(dev, non_blocking=True))
d = torch.from_numpy(p0).to(dev); lp = torch.empty(E * Wp, dtype=torch.float64, device=dev)
t('  logprob launch + sync', lambda: be.logprob(d, lp))
t('  logp .cpu().numpy() + isnan', lambda: np.any(np.isnan(lp.cpu().numpy())))
t('  torch.isnan(lp).any().item() on device', lambda: bool(torch.isnan(lp).any().item()))
s = DeviceEnsembleSampler(Wp, 7, batch.ctx, rng='philox', seed=3, n_ensembles=E, chain_on_device=True)
t('_upload_state', lambda: s._upload_state(p0))
t('_start_from (checks + upload)', lambda: s._start_from(p0))
t('_check_coords', lambda: s._check_coords(p0))
