"""Persistent sampler kernel vs one launch per half-step, by ensemble size and model (one ensemble,
philox stream, chain kept on the device): where is the crossover?  Feeds DeviceEnsembleSampler's automatic rule."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bisip_amd
from bisip_amd.sampler import DeviceEnsembleSampler
path = bisip_amd.DataFiles()['SIP-K389175']
for cls, kw, name in ((bisip_amd.PolynomialDecomposition, {}, 'PD reduced'), (bisip_amd.PeltonColeCole, dict(n_modes=2), 'CC2'), (bisip_amd.Dias2000, {}, 'Dias'),
                      (bisip_amd.PolynomialDecomposition, dict(variant='collapsed'), 'PD collapsed'), (bisip_amd.PolynomialDecomposition, dict(variant='reduced_comp'), 'PD comp'), (bisip_amd.PeltonColeCole, dict(n_modes=1), 'CC1'), (bisip_amd.PeltonColeCole, dict(n_modes=3), 'CC3'), (bisip_amd.Shin2015, {}, 'Shin')):
    m = cls(path, nwalkers=32, nsteps=10, **kw)
    ctx = m._context(); lo, hi = m.param_bounds; ndim = lo.size
    for W in (128, 192, 256, 384, 512, 768, 1024):
        rng = np.random.RandomState(W)
        centre = (lo + hi) / 2
        p0 = centre + 1e-3 * (hi - lo) * rng.randn(W, ndim)
        res = {}
        for persistent in (True, False):
            best = 1e9
            for rep in range(3):
                s = DeviceEnsembleSampler(W, ndim, ctx, rng='philox', seed=3, persistent=persistent, chain_on_device=True)
                t0 = time.perf_counter(); s.run_mcmc(p0, 2000); best = min(best, time.perf_counter() - t0)
            res[s.last_path] = 2000 / best
        print(f'{name:10s} W={W:5d}  ' + '  '.join(f'{k}: {v/1e3:7.1f} k it/s' for k, v in res.items()), flush=True)
