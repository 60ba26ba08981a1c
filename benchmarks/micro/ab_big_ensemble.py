"""A/B of two library builds on one box: the launch-per-half-step sampler of a double Cole-Cole ensemble too big for
several lanes per walker (k_stretch_half<ColeCole<2>> with one lane per walker).  python ab_big_ensemble.py [library]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from bisip_amd import _hip
if len(sys.argv) > 1: _hip.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch, bisip_amd
from bisip_amd.sampler import DeviceEnsembleSampler
from bisip_amd.synthetic import write_spectrum_file
import tempfile
spec_path = write_spectrum_file(os.path.join(tempfile.mkdtemp(), 's.csv'), 32, 0)
out = {'library': os.path.relpath(_hip.LIB_PATH, ROOT)}
for W in (131072, 524288, 1048576):
    m = bisip_amd.PeltonColeCole(spec_path, nwalkers=W, nsteps=4, n_modes=2)
    ctx = m._context(); lo, hi = m.param_bounds; ctx.set_bounds(m.param_bounds)
    p0 = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]) + 1e-3 * np.random.RandomState(0).randn(W, 7)
    best = None
    for rep in range(4):
        s = DeviceEnsembleSampler(W, 7, ctx, rng='philox', seed=1, persistent=False, chain_on_device=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.run_mcmc(p0, 4, thin_by=50)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rep: best = dt if best is None or dt < best else best
        path = s.last_path; s.close()
    out[str(W)] = {'walker_steps_per_s': float('%.4g' % (W * 200 / best)), 'path': path}
print(json.dumps(out))
