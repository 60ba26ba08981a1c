#!/usr/bin/env python3
"""Calibrates bench.py's `cpu_baseline` (the C oracle, kind "port") against the REAL reference.

Runs ONLY in the build container: it imports the reference from /root/reference exactly as
tests/golden/make_golden.py does (cython_funcs.pyx cythonized in a scratch directory outside
the repo, np.float_ aliased, inert emcee / corner modules) and times, on ONE core and on the
SAME theta rows of the metric workload (PolynomialDecomposition P=5, c=1, N=32, S=64):

  * the reference's own per-walker call -- Inversion._log_probability(theta, forward, bounds,
    w, zn, zn_err) in a Python loop, which is what emcee runs (src/bisip/models.py:71-76,111-118);
  * the oracle (oracle/bisip_oracle.c through oracle.logprob, one thread).

The ratio (reference evals/s) / (oracle evals/s) is written to profiles/cpu_calibration.json;
bench.py multiplies its on-box oracle rate by it to report what the reference itself would do
on the GPU box's host cores.  Neither the reference nor anything derived from its code leaves
this container: the JSON holds three numbers and the machine description.

Usage:  python benchmarks/cpu_calibration.py [--rows 3000] [--repeats 5]
"""
import argparse
import json
import os
import platform
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=3000)
    ap.add_argument('--repeats', type=int, default=5)
    ap.add_argument('--scratch', default=os.path.join(tempfile.gettempdir(), 'bisip_ref_build'))
    args = ap.parse_args()

    import make_golden
    import oracle
    from bench import C_EXP, make_problem
    from bisip_amd.synthetic import synthetic_columns, synthetic_theta

    bisip = make_golden.import_reference(make_golden.build_reference(args.scratch))
    data, taus, log_taus, bounds = make_problem()
    theta = synthetic_theta(bounds[0], bounds[1], args.rows, seed=2024)

    # the reference model on the same synthetic spectrum: written to a file, loaded by ITS load_data
    cols = synthetic_columns(32, 0)
    with tempfile.NamedTemporaryFile('w', suffix='.csv', delete=False) as fh:
        fh.write('freq,amp,pha,amp_err,pha_err\n')
        for r in cols:
            fh.write(','.join(repr(float(v)) for v in r) + '\n')
        path = fh.name
    model = bisip.PolynomialDecomposition(path, nwalkers=32, nsteps=10, poly_deg=5, c_exp=C_EXP)
    os.unlink(path)
    d = model.data
    assert np.array_equal(d['w'], data['w']) and np.allclose(d['zn'], data['zn'], rtol=1e-15, atol=0)
    argt = (model.forward, model.param_bounds, d['w'], d['zn'], d['zn_err'])

    prob = oracle.OracleProblem('PolynomialDecomposition', data['w'], data['zn'], data['zn_err'], bounds,
                                taus=taus, log_taus=log_taus, c_exp=C_EXP)
    ref_rates, orc_rates = [], []
    ref_lp = np.empty(args.rows)
    for _ in range(args.repeats):
        t0 = time.perf_counter()
        for i in range(args.rows):
            ref_lp[i] = model._log_probability(theta[i], *argt)
        ref_rates.append(args.rows / (time.perf_counter() - t0))
        t0 = time.perf_counter()
        orc_lp = oracle.logprob(prob, theta, n_threads=1)
        orc_rates.append(args.rows / (time.perf_counter() - t0))
    err = float(np.max(np.abs(ref_lp - orc_lp) / np.maximum(1.0, np.abs(ref_lp))))
    ref, orc = float(np.median(ref_rates)), float(np.median(orc_rates))
    cpu = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                cpu = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    rec = {
        'workload': 'PolynomialDecomposition poly_deg=5 c_exp=1.0, 32 synthetic frequencies, theta uniform in the prior box',
        'rows': args.rows, 'repeats': args.repeats,
        'reference_evals_per_s_1core': ref, 'oracle_evals_per_s_1core': orc,
        'reference_over_oracle': ref / orc,
        'max_rel_diff_reference_vs_oracle': err,
        'machine': {'cpu': cpu, 'python': platform.python_version(), 'numpy': np.__version__},
        'note': 'reference = Inversion._log_probability per walker in a Python loop (Cython Decomp_cyth + NumPy), '
                'the call emcee makes; oracle = oracle/bisip_oracle.c, 1 thread; medians of the repeats; '
                'measured in the build container (the reference never runs on the GPU box)',
    }
    out = os.path.join(ROOT, 'profiles', 'cpu_calibration.json')
    json.dump(rec, open(out, 'w'), indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
