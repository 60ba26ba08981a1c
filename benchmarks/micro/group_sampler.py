"""The multi-workgroup persistent sampler (k_stretch_group) against one launch per half-step, by ensemble size and
model: one ensemble of 1,025 ... 8,192 walkers, philox stream, chain kept on the device.  BASELINE config 2 is
the first case (single Cole-Cole, 32 synthetic frequencies, 4,096 walkers).  Feeds DeviceEnsembleSampler's
automatic rule.  us per half-step = wall time of run_mcmc / (2 x iterations)."""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bisip_amd
from bisip_amd.sampler import DeviceEnsembleSampler
from bisip_amd.synthetic import write_spectrum_file
path = write_spectrum_file(os.path.join(tempfile.mkdtemp(), 's.csv'), 32, 0)
its = int(os.environ.get('GROUP_ITS', '1000'))
for cls, kw, name in ((bisip_amd.PeltonColeCole, dict(n_modes=1), 'cfg2: ColeCole D=1, N=32'), (bisip_amd.PolynomialDecomposition, {}, 'PD reduced'),
                      (bisip_amd.PeltonColeCole, dict(n_modes=2), 'CC2'), (bisip_amd.Dias2000, {}, 'Dias'), (bisip_amd.Shin2015, {}, 'Shin'),
                      (bisip_amd.PolynomialDecomposition, dict(variant='collapsed'), 'PD collapsed')):
    m = cls(path, nwalkers=32, nsteps=10, **kw)
    ctx = m._context(); lo, hi = m.param_bounds; ndim = lo.size
    if hasattr(ctx, 'reduced_guard'):
        ctx.reduced_guard(False)                 # (the tier's guard is measured elsewhere: benchmarks/cfg4_sampler.py --no-guard)
    for W in ((4096, 2048, 8192, 1536) if name.startswith('cfg2') else (4096, 2048, 8192)):
        rng = np.random.RandomState(W)
        centre = (lo + hi) / 2
        p0 = centre + 1e-3 * (hi - lo) * rng.randn(W, ndim)
        rec = {'case': name, 'walkers': W, 'iterations': its}
        for persistent in (True, False):
            runs = []
            for rep in range(4):
                s = DeviceEnsembleSampler(W, ndim, ctx, rng='philox', seed=3, persistent=persistent, chain_on_device=True)
                t0 = time.perf_counter(); s.run_mcmc(p0, its); runs.append(time.perf_counter() - t0)
            best = min(runs[1:])
            rec[s.last_path] = {'us_per_half_step': round(best / its / 2 * 1e6, 2), 'it_per_s': round(its / best, 1),
                                'device_us_per_half_step': round((s.timing['enqueue_s'] + s.timing['drain_s']) / its / 2 * 1e6, 2)}
        print(json.dumps(rec), flush=True)
