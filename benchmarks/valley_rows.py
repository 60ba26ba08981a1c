#!/usr/bin/env python3
"""How far from the exact value is the kernel BISIP_VARIANT_AUTO picks on rows along the flat valley of
chi^2 -- b = b_ls + R^-1 z, |z| = 1 ... 1000 posterior sigmas, the rows a sampler evaluates from burn-in
to convergence -- for PolynomialDecomposition designs of degree 6-10?  One JSON line per design:
kernel, its estimate, and per scale (rows inside the prior box, worst relative error against the
reduced form in long double = bisip_ctx_reduced_check).  The guard of bisip_logprob is switched off
so that the kernel AUTO chose from its estimate is the one measured."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bisip_amd import _hip
from bisip_amd.synthetic import synthetic_columns
from bisip_amd.utils import columns_to_data


def design(n_freq, poly_deg, c_exp, idx):
    d = columns_to_data(synthetic_columns(n_freq, idx), 'mrad')
    per = np.log10(1. / d['w'])
    lt = np.linspace(np.floor(per.min() - 1), np.floor(per.max() + 1), 2 * n_freq)
    bounds = np.array([[0.9] + [-1.0] * (poly_deg + 1), [1.1] + [1.0] * (poly_deg + 1)])
    return d, 10 ** lt, np.array([lt ** i for i in range(poly_deg + 1)]), bounds


def valley_rows(ops, bounds, scale, rng, n_rows=3000):
    n = ops['R'].shape[0]
    db = np.linalg.solve(ops['R'], (scale * rng.randn(n_rows, n)).T).T
    b = ops['bhat'][None, :] + db
    t = np.concatenate([b[:, :1], b[:, 1:] / b[:, :1]], axis=1)
    return np.ascontiguousarray(t[np.all((bounds[0] < t) & (t < bounds[1]), axis=1)])


def main():
    worst_all = 0.0
    for poly_deg in (5, 6, 7, 8, 9, 10):
        for n_freq in (20, 32, 48, 64):
            for c_exp in (1.0, 0.5, 0.3):
                for idx in (0, 1, 2):
                    d, taus, log_taus, bounds = design(n_freq, poly_deg, c_exp, idx)
                    ops = _hip.polydecomp_operands(d['w'], d['zn'], d['zn_err'], taus, log_taus, c_exp)
                    ctx = _hip.HipContext(0, d['w'], d['zn'], d['zn_err'], bounds, poly_deg=poly_deg, c_exp=c_exp,
                                          taus=taus, log_taus=log_taus)
                    ctx.reduced_guard(False)
                    rng = np.random.RandomState(poly_deg * 100 + n_freq)
                    rec = {'poly_deg': poly_deg, 'n_freq': n_freq, 'c_exp': c_exp, 'spectrum': idx,
                           'kernel': ctx.kernel_name, 'estimate': ctx.reduced_error, 'scales': {}}
                    for scale in (1, 3, 10, 30, 100, 300, 1000):
                        t = valley_rows(ops, bounds, scale, rng)
                        if len(t) and ctx.variant in ('reduced', 'reduced_comp'):
                            err = ctx.reduced_check(t, ctx.logprob(t))
                            rec['scales'][scale] = [len(t), err]
                            worst_all = max(worst_all, err)
                    rec['worst'] = max((v[1] for v in rec['scales'].values()), default=0.0)
                    if '--all' in sys.argv or rec['worst'] > 1e-11:
                        print(json.dumps(rec), flush=True)
                    ctx.close()
    print(json.dumps({'worst_over_all_designs': worst_all}))


if __name__ == '__main__':
    main()
