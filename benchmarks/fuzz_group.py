#!/usr/bin/env python3
"""Randomised soak of the multi-workgroup sampler (k_stretch_group): random model, ensemble size (1,025 ... 32,768 for
the one-lane reduced PolynomialDecomposition kernels -- up to 256 workgroups on all XCDs, the two-level barrier --
and ... 8,192 for the others), iteration count, thinning, chunking, lanes per slot and stream; the chain, the stored
log-probabilities and the acceptance counts must equal those of one launch per half-step bit for bit.  A barrier
that lets one workgroup through early shows as a different chain: every case runs hundreds to thousands of them.

    python benchmarks/fuzz_group.py --cases 300 --seed 1
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    import bisip_amd
    from bisip_amd.sampler import DeviceEnsembleSampler
    from bisip_amd.synthetic import write_spectrum_file
    rng = np.random.RandomState(args.seed)
    tmp = tempfile.mkdtemp()
    bad, barriers, t0 = 0, 0, time.time()
    sizes = {}
    for case in range(args.cases):
        N = int(rng.choice([16, 20, 32]))
        path = write_spectrum_file(os.path.join(tmp, f's{case % 8}_{N}.csv'), N, int(rng.randint(0, 100)))
        kind = rng.randint(5)
        if kind <= 1:
            m = bisip_amd.PolynomialDecomposition(path, nwalkers=32, nsteps=4, poly_deg=int(rng.randint(0, 6)),
                                                  variant=str(rng.choice(['auto', 'reduced_comp'])))
            W = int(rng.choice([rng.randint(1025, 4097), rng.randint(4097, 8193), rng.randint(8193, 32769), 32768, 16384, 8192]))
        else:
            m = [bisip_amd.PeltonColeCole(path, nwalkers=32, nsteps=4, n_modes=int(rng.randint(1, 3))),
                 bisip_amd.Dias2000(path, nwalkers=32, nsteps=4), bisip_amd.Shin2015(path, nwalkers=32, nsteps=4)][kind - 2]
            W = int(rng.choice([rng.randint(1025, 4097), rng.randint(4097, 8193), 4096, 2048]))
        ctx = m._context()
        lo, hi = m.param_bounds
        ndim = lo.size
        if ndim > 7:
            continue
        centre = (np.r_[1.0, 0.004, np.zeros(ndim - 2)] if kind <= 1 and ndim > 1 else np.r_[1.0] if kind <= 1 else (lo + hi) / 2)
        p0 = centre + 1e-3 * (hi - lo) * rng.randn(W, ndim)
        p0 = np.clip(p0, lo + 1e-9 * (hi - lo), hi - 1e-9 * (hi - lo))
        n_it = int(rng.choice([8, 40, 200, 1000]))
        stored = int(rng.choice([2, 4, 8]))
        thin = max(1, n_it // stored)
        chunk = None if rng.rand() < 0.5 else int(rng.randint(1, stored + 1)) * thin
        stream = 'philox' if rng.rand() < 0.8 or W > 8192 else 'numpy'
        lanes = None if kind <= 1 or rng.rand() < 0.6 else str(rng.choice([1, 2, 4, 8]))
        seed = int(rng.randint(1 << 30))
        out = []
        for persistent in (True, False):
            os.environ.pop('BISIP_STRETCH_LANES', None)
            if lanes is not None:
                os.environ['BISIP_STRETCH_LANES'] = lanes
            np.random.seed(seed % 100000)
            s = DeviceEnsembleSampler(W, ndim, ctx, rng=stream, seed=seed, chunk=chunk, persistent=persistent, live_dangerously=True)
            s.run_mcmc(p0, stored, thin_by=thin)
            out.append((s.last_path, s.get_chain(), s.get_log_prob(), np.array(s.acceptance_fraction)))
            s.close()
        os.environ.pop('BISIP_STRETCH_LANES', None)
        a, b = out
        same = all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))
        ran_group = a[0] == 'persistent-multi-workgroup' and b[0] == 'launch-per-half-step'
        if ran_group:
            barriers += 2 * stored * thin
            sizes[W // 4096] = sizes.get(W // 4096, 0) + 1
        if not same or not ran_group:
            bad += 1
            print(json.dumps({'case': case, 'model': type(m).__name__, 'W': W, 'iterations': stored * thin, 'thin': thin, 'chunk': chunk,
                              'stream': stream, 'lanes': lanes, 'paths': [a[0], b[0]], 'same': bool(same)}), flush=True)
        ctx.close()
    print(json.dumps({'summary': True, 'cases': args.cases, 'seed': args.seed, 'failures': bad, 'barriers_crossed': barriers,
                      'cases_by_walkers_div_4096': {str(k): v for k, v in sorted(sizes.items())}, 'seconds': round(time.time() - t0, 1)}))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
