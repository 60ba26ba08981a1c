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
which = os.environ.get('BIG_MODEL', 'cc2')
for W in (131072, 524288, 1048576):
    if which == 'pd':
        m = bisip_amd.PolynomialDecomposition(spec_path, nwalkers=W, nsteps=4)
        centre = np.array([1.0, 0.005, -0.003, -0.001, 0.0005, 0.0002, 0.00001])
    else:
        m = bisip_amd.PeltonColeCole(spec_path, nwalkers=W, nsteps=4, n_modes=2)
        centre = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6])
    ctx = m._context(); lo, hi = m.param_bounds; ctx.set_bounds(m.param_bounds)
    p0 = centre + 1e-3 * np.abs(centre) * np.random.RandomState(0).randn(W, 7)
    best, best_t = None, None
    for rep in range(4):
        s = DeviceEnsembleSampler(W, 7, ctx, rng='philox', seed=1, persistent=False, chain_on_device=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.run_mcmc(p0, 4, thin_by=50)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rep and (best is None or dt < best): best, best_t = dt, dict(s.timing)
        path, stream = s.last_path, s.last_stream
        if rep == 3:
            # one chunk's launches alone, HIP events around the C call (50 iterations: 100 half-steps, with the chunk's
            # repacking and, where the stream is not drawn in place, without its draw kernel -- that one is in the wall)
            be = s.backend; st = dict(s._dev); st['nh'] = (W + 1) // 2; st['thin'] = 50
            from bisip_amd.sampler import affine_splits
            st['perm'] = be.tensor(affine_splits(1, W, 1000, 50), slot='c')
            st['chain'] = be.empty((1, W, 7), torch.float64); st['logp_chain'] = be.empty((1, W), torch.float64)
            if stream == 'in place':
                st['inline'] = (2.0, 1, 1000)
            else:
                for name, dt in (('active', torch.int32), ('partner', torch.int32), ('zz', torch.float64), ('factor', torch.float64), ('logu', torch.float64)):
                    st[name] = be.empty((50, 2, (W + 1) // 2), dt)
                be.draw(st, W, 2.0, 1, 1000, 50)
            be.run(st, 50); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); be.run(st, 50); e1.record(); torch.cuda.synchronize()
            chunk_us = e0.elapsed_time(e1) * 1e3 / 100
        s.close()
    out[str(W)] = {'walker_steps_per_s': float('%.4g' % (W * 200 / best)), 'path': path, 'wall_ms': round(best * 1e3, 2),
                   'us_per_half_step_wall': round(best / 400 * 1e6, 1),
                   'stream': stream, 'us_per_half_step_launches': round(chunk_us, 1),
                   'us_per_half_step_device': round((best_t['enqueue_s'] + best_t['drain_s'] + best_t.get('guard_s', 0.0)) / 400 * 1e6, 1),
                   'timing_ms': {k: round(v * 1e3, 2) for k, v in best_t.items()}}
out['model'] = which
out['packed_state'] = os.environ.get('BISIP_NO_PACKED_STATE') is None
print(json.dumps(out))
