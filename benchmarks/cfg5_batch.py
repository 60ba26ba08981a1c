#!/usr/bin/env python3
"""BASELINE config 5 (per-GPU share): a batch of independent synthetic spectra x 256 walkers
each, double Cole-Cole, all ensembles advanced together on one MI355X (512 spectra per GPU
when 4096 are sharded over 8).  Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--spectra', type=int, default=512)
    ap.add_argument('--walkers', type=int, default=256)
    ap.add_argument('--steps', type=int, default=100, help='stored samples')
    ap.add_argument('--thin-by', type=int, default=10)
    ap.add_argument('--chain', default='host', choices=['host', 'device'],
                    help="'device': stored samples stay in HBM; posterior mean/std are computed there")
    ap.add_argument('--repeat', type=int, default=3,
                    help='timed runs (fresh sampler each); the fastest is reported, all are listed: a box now and '
                         'then delays the first work after a synchronisation by tens of milliseconds')
    ap.add_argument('--persistent', action='store_true', help='force the persistent kernel (default: automatic)')
    ap.add_argument('--no-persistent', action='store_true', help='force one launch per half-step')
    args = ap.parse_args()
    import bisip_amd
    from bisip_amd.synthetic import synthetic_columns
    E, Wp = args.spectra, args.walkers
    tables = [synthetic_columns(32, i) for i in range(E)]
    batch = bisip_amd.SpectraBatch('PeltonColeCole', tables, nwalkers=Wp, nsteps=args.steps, n_modes=2)
    rng = np.random.RandomState(0)
    centre = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6])
    p0 = centre + 1e-3 * rng.randn(E, Wp, 7)
    from bisip_amd.sampler import DeviceEnsembleSampler
    batch.ctx.set_bounds(batch.param_bounds)

    def make():
        return DeviceEnsembleSampler(Wp, 7, batch.ctx, rng='philox', seed=3, n_ensembles=E,
                                     chain_on_device=(args.chain == 'device'),
                                     persistent=True if args.persistent else (False if args.no_persistent else None))
    prime = time.perf_counter()           # warm-up + clocks up (as bench.py primes before timing)
    while time.perf_counter() - prime < 0.25:
        make().run_mcmc(p0.reshape(-1, 7), 2, thin_by=args.thin_by)
    runs = []
    for _ in range(max(1, args.repeat)):
        s = make()
        t0 = time.perf_counter()
        s.run_mcmc(p0.reshape(-1, 7), args.steps, thin_by=args.thin_by)
        dt = time.perf_counter() - t0
        summary_s = percentile_s = None
        if args.chain == 'device':        # posterior mean / std of every spectrum, second half of the chain
            t1 = time.perf_counter()
            mean, std = s.param_moments(discard=args.steps // 2)
            summary_s = time.perf_counter() - t1
            dt += summary_s
            t2 = time.perf_counter()
            pct = s.param_percentiles((2.5, 50, 97.5), discard=args.steps // 2)     # not part of `seconds`
            percentile_s = time.perf_counter() - t2
            assert pct.shape == (3, E, 7) and np.all(pct[0] <= pct[1]) and np.all(pct[1] <= pct[2])
            assert mean.shape == (E, 7) and np.all(np.isfinite(std))
        runs.append((dt, summary_s, percentile_s, s))
    dt, summary_s, percentile_s, s = min(runs, key=lambda r: r[0])
    iters = args.steps * args.thin_by
    print(json.dumps({'config': 'cfg5 slice: double Cole-Cole, 32 frequencies', 'spectra': E, 'walkers_per_spectrum': Wp,
                      'chain': args.chain, 'path': s.last_path, 'summary_s': None if summary_s is None else round(summary_s, 5),
                      'percentile_s': None if percentile_s is None else round(percentile_s, 5),
                      'iterations': iters, 'stored': args.steps, 'thin_by': args.thin_by, 'seconds': round(dt, 4),
                      'seconds_all_runs': [round(r[0], 4) for r in runs],
                      'summary_s_all_runs': [None if r[1] is None else round(r[1], 5) for r in runs],
                      'it_per_s': round(iters / dt, 1),
                      'us_per_half_step': round((s.timing['enqueue_s'] + s.timing['drain_s'] + s.timing.get('guard_s', 0.0)) / iters / 2 * 1e6, 2),
                      'walker_steps_per_s': float('%.4g' % (iters * E * Wp / dt)),
                      'timing_s': {k: round(v, 4) for k, v in s.timing.items()},
                      'acceptance': round(float(s.acceptance_fraction.mean()), 3)}))


if __name__ == '__main__':
    main()
