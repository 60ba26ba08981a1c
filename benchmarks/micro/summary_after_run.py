#!/usr/bin/env python3
"""Where does the time of the first summary after a long run go?  cfg5 slice, 500 stored samples."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bisip_amd
from bisip_amd import _hip
from bisip_amd.sampler import DeviceEnsembleSampler
from bisip_amd.synthetic import synthetic_columns

E, Wp = 512, 256
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
mode = sys.argv[2] if len(sys.argv) > 2 else ''      # letters: k = keep samplers alive, p = percentiles too, t = trivial kernel first,
                                                     # l = one launch per half-step instead of the persistent kernel,
                                                     # s / S = stream synchronisation for the summary / everywhere
alive = []


def cpu_stat():
    out = {}
    for path in ('/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.stat'):
        try:
            for line in open(path):
                k, v = line.split()
                out[k] = int(v)
            break
        except OSError:
            pass
    t = os.times()
    out['proc_cpu_s'] = t.user + t.system
    out['wall'] = time.perf_counter()
    return out


def delta(a, b):
    keys = [k for k in ('nr_throttled', 'throttled_usec', 'throttled_time', 'nr_periods') if k in a]
    return {k: b[k] - a[k] for k in keys} | {'proc_cpu_s': round(b['proc_cpu_s'] - a['proc_cpu_s'], 3), 'wall_s': round(b['wall'] - a['wall'], 3)}


tiny = torch.zeros(64, device='cuda')
batch = bisip_amd.SpectraBatch('PeltonColeCole', [synthetic_columns(32, i) for i in range(E)], nwalkers=Wp, nsteps=steps, n_modes=2)
p0 = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]) + 1e-3 * np.random.RandomState(0).randn(E * Wp, 7)
if 'S' in mode:        # every synchronisation of the sampler too
    from bisip_amd.sampler import HipStretchBackend
    HipStretchBackend.synchronize = lambda self: torch.cuda.current_stream(self.device).synchronize()
for rep in range(8):
    s = DeviceEnsembleSampler(Wp, 7, batch.ctx, rng='philox', seed=3, n_ensembles=E, chain_on_device=True,
                              persistent=False if 'l' in mode else None)     # l = one launch per half-step
    c0 = cpu_stat()
    s.run_mcmc(p0, steps, thin_by=40)
    c1 = cpu_stat()
    be = s.backend
    t = s.device_chain()
    n_total, W, nd = t.shape
    first = steps // 2
    n = n_total - first
    if 't' in mode:
        t0 = time.perf_counter(); tiny.add_(1.0); torch.cuda.synchronize(); print(f'   trivial kernel + sync {1e3*(time.perf_counter()-t0):.3f} ms')
    T = [time.perf_counter()]
    mean = be.empty((E, nd), torch.float64); std = be.empty((E, nd), torch.float64)
    work = be.empty((max(1, _hip.chain_moments_workspace(n, E, nd)),), torch.float64)
    T.append(time.perf_counter())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    _hip.chain_moments_dev(t.data_ptr() + 8 * first * W * nd, n, W * nd, E, Wp, nd, mean.data_ptr(), std.data_ptr(), work.data_ptr(), be.stream())
    ev[1].record()
    T.append(time.perf_counter())
    if 's' in mode or 'S' in mode:
        torch.cuda.current_stream().synchronize()      # hipStreamSynchronize instead of hipDeviceSynchronize
    else:
        be.synchronize()
    T.append(time.perf_counter())
    on_device_ms = ev[0].elapsed_time(ev[1])      # the device's own clock between the two markers
    m = mean.cpu().numpy()
    T.append(time.perf_counter())
    _hip.chain_moments_dev(t.data_ptr() + 8 * first * W * nd, n, W * nd, E, Wp, nd, mean.data_ptr(), std.data_ptr(), work.data_ptr(), be.stream())
    be.synchronize()
    T.append(time.perf_counter())
    c2 = cpu_stat()
    print('   cgroup/process over the run:', delta(c0, c1), ' over the summary:', delta(c1, c2))
    print(f'run {rep}: alloc {1e3*(T[1]-T[0]):.3f}  launch {1e3*(T[2]-T[1]):.3f}  sync {1e3*(T[3]-T[2]):.3f} (markers on the device {on_device_ms:.3f})  copy {1e3*(T[4]-T[3]):.3f}  second call+sync {1e3*(T[5]-T[4]):.3f} ms', flush=True)
    if 'p' in mode:
        t0 = time.perf_counter(); s.param_percentiles((2.5, 50, 97.5), discard=first); print(f'   percentiles {1e3*(time.perf_counter()-t0):.3f} ms')
    if 'k' in mode:
        alive.append(s)
    del s, t
