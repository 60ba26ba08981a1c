#!/usr/bin/env python3
"""End-to-end fit() throughput (MCMC iterations/s) -- BASELINE config 1 (README quickstart)
and larger ensembles -- for the device-resident and host-loop samplers.  1 GPU."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import gc
    import torch  # noqa: F401
    import bisip_amd
    # a full collection over torch's module graph takes ~80 ms and would land in one timed run
    gc.collect()
    gc.freeze()
    path = bisip_amd.DataFiles()['SIP-K389175']
    cases = [
        ('cfg1 PolynomialDecomposition K389175 32x1000 (reference README: 558.64 it/s)',
         bisip_amd.PolynomialDecomposition, {}, 32, 1000),
        ('PeltonColeCole n_modes=2 K389175 64x1000 (reference tutorial: 356-435 it/s)',
         bisip_amd.PeltonColeCole, dict(n_modes=2), 64, 1000),
        ('Dias2000 K389175 32x1000 (reference tutorial: 511-533 it/s)', bisip_amd.Dias2000, {}, 32, 1000),
        ('PolynomialDecomposition K389175 32x5000 (nsteps default)', bisip_amd.PolynomialDecomposition, {}, 32, 5000),
        ('PeltonColeCole n_modes=2 512 walkers x 1000', bisip_amd.PeltonColeCole, dict(n_modes=2), 512, 1000),
        ('PolynomialDecomposition 4096 walkers x 200', bisip_amd.PolynomialDecomposition, {}, 4096, 200),
        ('PeltonColeCole n_modes=2 4096 walkers x 200', bisip_amd.PeltonColeCole, dict(n_modes=2), 4096, 200),
        ('PolynomialDecomposition 32768 walkers x 50', bisip_amd.PolynomialDecomposition, {}, 32768, 50),
    ]
    for name, cls, kw, W, nsteps in cases:
        for sampler in ('device-philox-persistent', 'device-philox', 'device', 'device-launches', 'host'):
            m = cls(path, nwalkers=W, nsteps=nsteps, **kw)
            lo, hi = m.param_bounds
            np.random.seed(42)
            p0 = np.random.uniform(lo, hi, (W, lo.size))
            if sampler.startswith('device-philox'):
                pers = sampler.endswith('persistent')
                from bisip_amd.sampler import DeviceEnsembleSampler
                ctx = m._context()
                ctx.set_bounds(m.param_bounds)
                prime = time.perf_counter()       # warm-up + ~0.25 s of the same work: kernels loaded, clocks up
                while time.perf_counter() - prime < 0.25:
                    DeviceEnsembleSampler(W, lo.size, ctx, rng='philox', seed=1, persistent=pers).run_mcmc(p0, 50)
                runs = []                         # fastest of three: a box now and then stalls a 3 ms run for 10 ms
                for _ in range(3):
                    smp = DeviceEnsembleSampler(W, lo.size, ctx, rng='philox', seed=1, persistent=pers)
                    t0 = time.perf_counter()
                    smp.run_mcmc(p0, nsteps)
                    runs.append(time.perf_counter() - t0)
                dt = min(runs)
                m._sampler = smp
                m._Inversion__fitted = True
            else:
                # 'device' = fit() defaults (NumPy-order stream, persistent kernel when it fits);
                # 'device-launches' = same stream, one launch per half-step
                kw_fit = dict(sampler='device', persistent=(None if sampler == 'device' else False)) if sampler != 'host' else dict(sampler='host')
                m.nsteps = 50
                prime = time.perf_counter()       # warm-up (context, kernels, allocator) + clocks up
                while time.perf_counter() - prime < 0.25:
                    m.fit(p0=p0, **kw_fit)
                m.nsteps = nsteps
                runs = []
                for _ in range(1 if sampler == 'host' else 3):
                    t0 = time.perf_counter()
                    m.fit(p0=p0, **kw_fit)
                    runs.append(time.perf_counter() - t0)
                dt = min(runs)
            print(json.dumps({'case': name, 'sampler': sampler, 'walkers': W, 'nsteps': nsteps,
                              'seconds': round(dt, 4), 'seconds_all_runs': [round(r, 4) for r in runs], 'it_per_s': round(nsteps / dt, 1),
                              'walker_steps_per_s': float('%.4g' % (nsteps * W / dt)),
                              'acceptance': round(float(m.sampler.acceptance_fraction.mean()), 3),
                              'path': getattr(m.sampler, 'last_path', None),
                              'timing_s': {k: round(v, 4) for k, v in getattr(m.sampler, 'timing', {}).items()}}),
                  flush=True)


if __name__ == '__main__':
    main()
