"""What the stepped exponentials (kernels.h: BOUNDS_GRID) do to SMALL work, where several lanes share a walker:
fit() of one spectrum and emcee-sized calls, each against BISIP_NO_GRID=1 on the same box.  Prints iterations/s
and microseconds per call.  (profiles/r03_micro_grid_small_ensembles_before.txt holds the same script on two
designs that were taken out: one frequency per lane and round, and a series-corrected tier for rounded grids.)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import bisip_amd
from bisip_amd import _hip
from bisip_amd.batch import default_params
from bisip_amd.sampler import DeviceEnsembleSampler
from bisip_amd.synthetic import synthetic_columns
from bisip_amd.utils import columns_to_data, load_data

real = load_data(bisip_amd.DataFiles()['SIP-K389175'])
synth = columns_to_data(synthetic_columns(32, 0), 'mrad')
for label, d in (('bundled SIP-K389175 (N=20, no grid: control)', real), ('synthetic N=32 (exact grid)', synth)):
    for name, mid, kw, centre in (('PeltonColeCole', 1, dict(n_modes=2), [1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]),
                                  ('PeltonColeCole', 1, dict(n_modes=1), [1.0, 0.3, -3.0, 0.5]),
                                  ('Shin2015', 3, {}, [0.5, 0.5, -14.0, -6.0, 0.5, 0.5])):
        bounds = np.array(list(default_params(name, **kw).values()), float).T
        ndim = bounds.shape[1]
        for W in (32, 64, 256, 1024):
            res = {}
            for grid in (True, False):
                if grid: os.environ.pop('BISIP_NO_GRID', None)
                else: os.environ['BISIP_NO_GRID'] = '1'
                ctx = _hip.HipContext(mid, d['w'], d['zn'], d['zn_err'], bounds, **kw)
                rng = np.random.RandomState(1)
                p0 = np.array(centre) + 1e-4 * rng.randn(W, ndim)
                best = 0.0
                for rep in range(4):
                    s = DeviceEnsembleSampler(W, ndim, ctx, rng='philox', seed=3)
                    t = time.perf_counter(); s.run_mcmc(p0, 2000); dt = time.perf_counter() - t
                    best = max(best, 2000 / dt)
                th = rng.uniform(bounds[0], bounds[1], (W // 2, ndim))
                for _ in range(50): ctx.logprob(th)
                t = time.perf_counter()
                for _ in range(500): ctx.logprob(th)
                us = (time.perf_counter() - t) / 500 * 1e6
                res[grid] = (best, us, ctx.loop_flags, s.last_path)
                ctx.close()
            g, n = res[True], res[False]
            print(f'{label} | {name} {kw} | {W} walkers ({g[3]}): fit {g[0]:,.0f} it/s stepped (flags {g[2]}) vs {n[0]:,.0f} direct = {g[0]/n[0]:.2f}x; '
                  f'call of {W//2} rows {g[1]:.1f} us vs {n[1]:.1f} us', flush=True)
