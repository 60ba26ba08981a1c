#!/usr/bin/env python3
"""Randomised parity campaign: HIP kernels (through the C ABI) against the CPU oracle on random
models, shapes, error scales, prior boxes and batch sizes -- a wider net than the pytest suite
(which has to finish in seconds).  Uses oracle/ as the checker, like tests/ do.

    python benchmarks/fuzz_parity.py --cases 300 --seed 1 > gpurun_out/fuzz.jsonl

One JSON line per case (worst relative log-prob / Z errors against the stated tolerances,
-inf agreement), then a summary line; exits 1 on any violation.

--valley moves half of the checked rows of every PolynomialDecomposition case into the flat valley of
chi^2 -- most of them onto the shell log-probability = 0 -- where the REFERENCE'S OWN double arithmetic is
up to 1e-7 (relative to max(1, |logp|)) from the exact value of its formula on nearly collinear designs.
There "agrees with the reference to 1e-10" cannot be asked of any arithmetic that is not the reference's
bit for bit, so each row is judged against the exact value too (bisip_polydecomp_reduced_reference, pinned
by 50-digit arithmetic in tests/test_host_logic.py): a row counts as a violation when the kernel is more
than 1e-10 from the reference AND more than 1e-10 from the exact value.  The summary reports both
distances, and the reference's own.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LOGP_TOL, Z_TOL = 1e-10, 1e-12
NAMES = ['PolynomialDecomposition', 'PeltonColeCole', 'Dias2000', 'Shin2015']


def valley_rows(ops, bounds, rng, n_rows):
    """Rows along the flat valley of chi^2 of a PolynomialDecomposition design, b = b_ls + s R^-1 z:
    a third at s = 1 ... 30 posterior sigmas (a sampler from convergence back to burn-in), two thirds ON
    the shell log-probability = 0 (s such that rest + s^2 |z|^2 = 2 lconst), where the tolerance's
    denominator max(1, |logp|) is 1 and the absolute error of a chi^2 of several hundred counts."""
    n = ops['R'].shape[0]
    if not np.all(np.isfinite(ops['bhat'])) or np.any(np.diag(ops['R']) == 0):
        return np.empty((0, n))
    z = rng.randn(n_rows, n)
    sc = np.where(np.arange(n_rows) % 3 == 0, rng.choice([1.0, 3.0, 10.0, 30.0], n_rows),
                  np.sqrt(np.maximum(2.0 * ops['lconst'] - ops['rest'], 0.0) / (z * z).sum(axis=1)))
    with np.errstate(all='ignore'):
        db = np.linalg.solve(ops['R'], (z * sc[:, None]).T).T
        b = ops['bhat'][None, :] + db
        t = np.concatenate([b[:, :1], b[:, 1:] / b[:, :1]], axis=1)
    t = t[np.all(np.isfinite(t), axis=1)]
    return t[np.all((bounds[0] < t) & (t < bounds[1]), axis=1)]


def frequencies_as_in_a_file(cols, key):
    """ColeCole / Shin choose their loop by the frequencies (kernels.h: BOUNDS_GRID): a third of their problems
    keep the generator's exact geometric grid (exponentials stepped by multiplication), a third hold it
    rounded to 4-6 digits as a data file would, a third have frequencies moved by up to 3 % (both: one
    exponential per frequency).  Decided from `key`, not from the campaign's random stream: every problem of
    every earlier campaign keeps its number and, for the other models, its content."""
    sub = np.random.RandomState(key % (1 << 31))
    mode = sub.randint(3)
    if mode == 1:
        digits = int(sub.randint(4, 7))
        cols[:, 0] = np.array([float('%.*g' % (digits, f)) for f in cols[:, 0]])
    elif mode == 2:
        cols[:, 0] *= 1.0 + 0.03 * sub.uniform(-1, 1, len(cols))
    return cols


def draw_case(rng, widen=1.0, valley=False):
    """One random problem: model, shape, spectrum, prior box, theta batch, rows to check."""
    from bisip_amd.batch import default_params
    from bisip_amd.synthetic import synthetic_columns
    from bisip_amd.utils import columns_to_data
    names = NAMES
    model = names[rng.randint(4)]
    N = int(rng.choice([1, 2, 3, 5, 16, 20, 21, 24, 30, 32, 33, 48, 64, 80]))
    idx = int(rng.randint(0, 4096))
    cols = synthetic_columns(N, idx)
    if model in ('PeltonColeCole', 'Shin2015'):
        cols = frequencies_as_in_a_file(cols, idx * 131 + N)
    cols[:, 3] *= 10.0 ** rng.uniform(-1, 1)          # amplitude errors x0.1 .. x10
    cols[:, 4] *= 10.0 ** rng.uniform(-1, 1)          # phase errors
    d = columns_to_data(cols, 'mrad')
    kw, okw, variants = {}, {}, ['auto']
    if model == 'PolynomialDecomposition':
        P = int(rng.randint(0, 11))
        c_exp = float(rng.choice([1.0, 0.5, rng.uniform(0.2, 1.0)]))
        S = int(rng.choice([2 * N, 40, max(2, N)]))
        per = np.log10(1. / d['w'])
        lt = np.linspace(np.floor(per.min() - 1), np.floor(per.max() + 1), S)
        taus, log_taus = 10 ** lt, np.array([lt ** i for i in range(P + 1)])
        kw = dict(poly_deg=P, c_exp=c_exp, taus=taus, log_taus=log_taus)
        okw = dict(taus=taus, log_taus=log_taus, c_exp=c_exp)
        # 'auto' = what the product runs: the QR-reduced kernel where its host-side error estimate
        # allows, else the per-frequency form
        variants = ['auto', 'collapsed', 'reduced_comp'] + (['faithful'] if P <= 7 else []) + (['wave'] if N <= 64 else [])
        params = default_params(model, poly_deg=P)
    elif model == 'PeltonColeCole':
        D = int(rng.randint(1, 6))
        kw = okw = dict(n_modes=D)
        params = default_params(model, n_modes=D)
    else:
        params = default_params(model)
    bounds = np.array(list(params.values()), float).T
    if widen != 1.0:
        mid0, half0 = bounds.mean(0), 0.5 * (bounds[1] - bounds[0])
        bounds = np.array([mid0 - widen * half0, mid0 + widen * half0])
        params = {k2: [bounds[0, i], bounds[1, i]] for i, k2 in enumerate(params)}
    # sometimes a narrower prior box, so that a good share of the rows is outside
    if rng.rand() < 0.3:
        mid, half = bounds.mean(0), 0.5 * (bounds[1] - bounds[0])
        bounds = np.array([mid - half * rng.uniform(0.3, 1.0, half.size), mid + half * rng.uniform(0.3, 1.0, half.size)])
    ndim = bounds.shape[1]
    W = int(rng.choice([1, 7, 63, 64, 65, 300, 4097, 9000, 20000, 40000]))
    full = np.array(list(params.values()), float).T
    theta = rng.uniform(full[0], full[1], (W, ndim))
    if model == 'PolynomialDecomposition':
        theta[: W // 2, 1:] *= 10.0 ** rng.uniform(-4, 0)   # clouds where the fit is decent
    for _ in range(min(W, 12)):                      # edge rows: non-finite and on-bound components
        r, q = rng.randint(W), rng.randint(ndim)
        theta[r, q] = rng.choice([np.nan, np.inf, -np.inf, bounds[0, q], bounds[1, q], full[0, q], full[1, q]])
    n_ref = min(W, 1500)
    rows = np.sort(rng.choice(W, n_ref, replace=False))
    if valley and model == 'PolynomialDecomposition' and 2 * N >= kw['poly_deg'] + 2 and W >= 64:
        # a share of the checked rows moves into the valley / onto the shell (drawn AFTER everything
        # else, so the rest of the case is the one the plain campaign of the same seed draws)
        from bisip_amd import _hip
        ops = _hip.polydecomp_operands(d['w'], d['zn'], d['zn_err'], kw['taus'], kw['log_taus'], kw['c_exp'])
        t = valley_rows(ops, bounds, np.random.RandomState(rng.randint(1 << 30)), 3 * min(len(rows), 600))
        k = min(len(t), len(rows) // 2)
        theta[rows[:k]] = t[:k]
    return dict(model=model, N=N, d=d, kw=kw, okw=okw, variants=variants, bounds=bounds, W=W, theta=theta,
                rows=rows, ndim=ndim)




def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--widen', type=float, default=1.0,
                    help='scale every default prior box about its centre (a user may widen params bounds)')
    ap.add_argument('--valley', action='store_true',
                    help='PolynomialDecomposition: half of the checked rows lie along the flat valley of chi^2, '
                         'most of them on the shell log-probability = 0')
    ap.add_argument('--only', type=int, default=None, help='evaluate just this case (same random stream) and print the worst rows')
    args = ap.parse_args()
    import oracle
    from bisip_amd import _hip
    rng = np.random.RandomState(args.seed)
    MODEL_IDS = {name: i for i, name in enumerate(NAMES)}
    worst = dict(logp=0.0, Z=0.0)
    valley = dict(cases=0, worst_auto_vs_exact=0.0, worst_comp_vs_exact=0.0, worst_reference_vs_exact=0.0,
                  worst_auto_vs_reference=0.0, cases_where_reference_is_off=0)
    bad = 0
    # PolynomialDecomposition problems whose design has a triangle (2N >= P+2): which kernel AUTO ran
    auto = dict(problems=0, reduced=0, reduced_comp=0, collapsed=0, worst_err_reduced=0.0, by_degree={})
    t_start = time.time()
    for case in range(args.cases):
        c = draw_case(rng, args.widen, args.valley)
        model, N, d, kw, okw, variants = c['model'], c['N'], c['d'], c['kw'], c['okw'], c['variants']
        bounds, W, theta, rows, ndim = c['bounds'], c['W'], c['theta'], c['rows'], c['ndim']
        if args.only is not None and case != args.only:
            continue
        prob = oracle.OracleProblem(model, d['w'], d['zn'], d['zn_err'], bounds, **okw)
        want = oracle.logprob(prob, theta[rows], n_threads=8)
        rec = dict(case=case, model=model, N=N, W=W, ndim=ndim, **{k2: (v if np.isscalar(v) else None) for k2, v in kw.items() if k2 in ('poly_deg', 'c_exp', 'n_modes')})
        errs = {}
        exact = None
        if args.valley and model == 'PolynomialDecomposition' and 2 * N >= kw['poly_deg'] + 2:
            # the exact log-probability of the rows inside the box (the prior decides the others, exactly)
            exact = _hip.polydecomp_reduced_reference(d['w'], d['zn'], d['zn_err'], kw['taus'], kw['log_taus'],
                                                      kw['c_exp'], np.nan_to_num(theta[rows], nan=0.0, posinf=0.0, neginf=0.0))
            inbox = np.isfinite(want)
            rec['reference_vs_exact'] = float('%.3g' % np.max(np.abs(want[inbox] - exact[inbox]) / np.maximum(1.0, np.abs(exact[inbox])))) if inbox.any() else 0.0
            valley['cases'] += 1
            valley['worst_reference_vs_exact'] = max(valley['worst_reference_vs_exact'], rec['reference_vs_exact'])
            valley['cases_where_reference_is_off'] += rec['reference_vs_exact'] > LOGP_TOL
        for v in variants:
            try:
                ctx = _hip.HipContext(MODEL_IDS[model], d['w'], d['zn'], d['zn_err'], bounds, variant=v, **kw)
                got = ctx.logprob(theta)[rows]
            except RuntimeError as exc:      # a variant that does not support this shape says so
                if 'status -4' not in str(exc):
                    raise
                rec.setdefault('unsupported', []).append(v)
                continue
            fin = np.isfinite(want)
            same = np.array_equal(np.isneginf(got), np.isneginf(want)) and not np.any(np.isnan(got))
            rel = np.abs(got[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))
            e_ref = float(np.max(rel)) if fin.any() else 0.0
            e = e_ref
            if exact is not None and fin.any():
                rel_x = np.abs(got[fin] - exact[fin]) / np.maximum(1.0, np.abs(exact[fin]))
                e = float(np.max(np.minimum(rel, rel_x)))        # the nearer of the two yardsticks, row by row
                errs[('auto' if v == 'auto' else ctx.kernel_name) + ' vs exact'] = float(np.max(rel_x))
                if v == 'auto':
                    valley['worst_auto_vs_exact'] = max(valley['worst_auto_vs_exact'], float(np.max(rel_x)))
                    valley['worst_auto_vs_reference'] = max(valley['worst_auto_vs_reference'], e_ref)
                if v == 'reduced_comp':
                    valley['worst_comp_vs_exact'] = max(valley['worst_comp_vs_exact'], float(np.max(rel_x)))
            errs[ctx.kernel_name if v != 'auto' else 'auto:' + ctx.kernel_name] = e
            if v == 'auto' and model == 'PolynomialDecomposition':
                rec['reduced_error_estimate'] = float('%.3g' % ctx.reduced_error)
                rec['auto_variant'] = ctx.variant
                if 2 * N >= kw['poly_deg'] + 2:
                    auto['problems'] += 1
                    auto[ctx.variant] += 1
                    deg = auto['by_degree'].setdefault(str(kw['poly_deg']), dict(problems=0, reduced=0, reduced_comp=0, collapsed=0))
                    deg['problems'] += 1
                    deg[ctx.variant] += 1
                    if ctx.variant != 'collapsed':
                        auto['worst_err_reduced'] = max(auto['worst_err_reduced'], e)
            if args.only is not None and fin.any():
                rel = np.abs(got - want) / np.maximum(1.0, np.abs(want))
                rel[~fin] = 0
                top = np.argsort(rel)[-3:]
                print(v, 'worst rows', [(float(want[i]), float(got[i]), float(rel[i])) for i in top], 'bounds', bounds.tolist() if v == variants[0] else '', file=sys.stderr)
            # valley campaign: the per-frequency formulations (collapsed / faithful / wave) add up, in double,
            # terms a million times their sum exactly as the reference does, only not bit for bit: on those
            # rows they are as far from the exact value as the reference is, and are reported, not judged
            judged = exact is None or v in ('auto', 'reduced_comp')
            if not judged:
                valley['worst_per_frequency_vs_exact'] = max(valley.get('worst_per_frequency_vs_exact', 0.0),
                                                             float(np.max(rel_x)) if fin.any() else 0.0)
            if not same or (judged and e > LOGP_TOL):
                bad += 1
                rec.setdefault('violations', []).append(dict(variant=v, err=e, neg_inf_match=bool(same)))
            worst['logp'] = max(worst['logp'], e)
            if v == ('collapsed' if model == 'PolynomialDecomposition' else 'auto'):
                ok_rows = rows[np.all(np.isfinite(theta[rows]), axis=1)][:200]
                if ok_rows.size:
                    Zw = oracle.forward(prob, theta[ok_rows])
                    Zg = ctx.forward(theta)[ok_rows]
                    finite = np.isfinite(Zw)
                    ez = float(np.max(np.abs(Zg[finite] - Zw[finite])) / max(1.0, float(np.max(np.abs(Zw[finite]))))) if finite.any() else 0.0
                    # (valley rows: forward() of the reference is itself 1e-12 ... 2e-11 off on degree 8-10
                    # designs -- sums of terms 1e3-1e5 times the response -- reported, not judged)
                    if (ez > Z_TOL and exact is None) or not np.array_equal(np.isfinite(Zg), finite):
                        bad += 1
                        rec.setdefault('violations', []).append(dict(variant=v, forward_err=ez))
                    worst['Z'] = max(worst['Z'], ez)
                    errs['forward'] = ez
            ctx.close()
        rec['err'] = {k2: float('%.3g' % v2) for k2, v2 in errs.items()}
        print(json.dumps(rec), flush=True)
    print(json.dumps(dict(summary=True, cases=args.cases, seed=args.seed, valley=bool(args.valley), violations=bad,
                          worst_logp_rel_err=worst['logp'], worst_Z_rel_err=worst['Z'],
                          tolerances=dict(logp=LOGP_TOL, Z=Z_TOL), seconds=round(time.time() - t_start, 1),
                          **({'valley_rows': valley} if args.valley else {}),
                          auto_on_polydecomp=dict(auto, frac_reduced=round((auto['reduced'] + auto['reduced_comp']) /
                                                                            max(1, auto['problems']), 4)))))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
