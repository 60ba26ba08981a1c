#!/usr/bin/env python3
"""Generate the golden parity vectors under tests/golden/ from the REAL reference.

Runs ONLY in the build container (it needs /root/reference and cython); the
resulting ``*.npz`` files are data (inputs + expected outputs) and are what
travels to the GPU box.  Recipe = SURVEY.md Appendix C:

* the reference's ``cython_funcs.pyx`` is cythonized in a scratch directory
  outside the repo (nothing generated is ever committed),
* ``np.float_`` is aliased (removed in NumPy 2) and inert ``emcee``/``corner``
  modules are registered so that ``import bisip`` works without the sampler,
* every value below comes from the reference's own ``load_data``,
  ``PolynomialDecomposition.__init__``, ``forward`` and ``_log_probability``.

Usage:  python tests/golden/make_golden.py [--scratch /tmp/bisip_ref_build]
"""

import argparse
import os
import shutil
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'


def build_reference(scratch):
    os.makedirs(scratch, exist_ok=True)
    pkg = os.path.join(scratch, 'pkg')
    if os.path.isdir(pkg):
        shutil.rmtree(pkg)
    shutil.copy(os.path.join(REF, 'src/bisip/cython_funcs.pyx'), scratch)
    with open(os.path.join(scratch, 'setup.py'), 'w') as fh:
        fh.write(
            "import numpy\n"
            "from setuptools import setup, Extension\n"
            "from Cython.Build import cythonize\n"
            "setup(ext_modules=cythonize([Extension('cython_funcs', ['cython_funcs.pyx'],\n"
            "      include_dirs=[numpy.get_include()])], language_level=3))\n")
    subprocess.check_call([sys.executable, 'setup.py', 'build_ext', '--inplace'],
                          cwd=scratch, stdout=subprocess.DEVNULL)
    os.makedirs(pkg)
    shutil.copytree(os.path.join(REF, 'src/bisip'), os.path.join(pkg, 'bisip'))
    for fn in os.listdir(scratch):
        if fn.startswith('cython_funcs.') and fn.endswith('.so'):
            shutil.copy(os.path.join(scratch, fn), os.path.join(pkg, 'bisip'))
    return pkg


def import_reference(pkg):
    np.float_ = np.float64
    sys.modules['emcee'] = types.ModuleType('emcee')
    m = types.ModuleType('corner')
    m.corner = None
    sys.modules['corner'] = m
    os.environ['MPLBACKEND'] = 'Agg'
    sys.path.insert(0, pkg)
    import bisip  # noqa
    return bisip


def edge_rows(lo, hi, inside):
    """Rows that exercise the open-box prior: below/above/on-bound/NaN/inf."""
    rows = []
    nd = lo.size
    for q, kind in [(0, 'below'), (nd - 1, 'above'), (0, 'on_lo'), (nd - 1, 'on_hi'),
                    (1 % nd, 'nan'), (0, 'pinf'), (nd - 1, 'ninf'), (0, 'inside')]:
        r = inside.copy()
        if kind == 'below':
            r[q] = lo[q] - 0.25 * (hi[q] - lo[q])
        elif kind == 'above':
            r[q] = hi[q] + 0.25 * (hi[q] - lo[q])
        elif kind == 'on_lo':
            r[q] = lo[q]
        elif kind == 'on_hi':
            r[q] = hi[q]
        elif kind == 'nan':
            r[q] = np.nan
        elif kind == 'pinf':
            r[q] = np.inf
        elif kind == 'ninf':
            r[q] = -np.inf
        rows.append(r)
    return np.array(rows)


def evaluate(model, theta):
    """Reference forward + log-probability for every row of theta."""
    d = model.data
    w, zn, zn_err = d['w'], d['zn'], d['zn_err']
    bounds = model.param_bounds
    n = theta.shape[0]
    Z = np.full((n, 2, d['N']), np.nan)
    logp = np.empty(n)
    with np.errstate(all='ignore'):
        for i in range(n):
            logp[i] = model._log_probability(theta[i], model.forward, bounds, w, zn, zn_err)
            if np.all(np.isfinite(theta[i])):
                Z[i] = model.forward(theta[i], w)
    return Z, logp


def polydecomp_lsq(model):
    """Near-posterior centre for PolynomialDecomposition: weighted least squares on
    the (linear) forward model, evaluated through the reference's own forward()."""
    d = model.data
    w, zn, zn_err = d['w'], d['zn'], d['zn_err']
    P = model.poly_deg
    base = model.forward(np.r_[1.0, np.zeros(P + 1)], w).ravel()  # == 1 + 0i
    cols = [base]
    for p in range(P + 1):
        e = np.zeros(P + 1)
        e[p] = 1.0
        cols.append(model.forward(np.r_[1.0, e], w).ravel() - base)  # == -G_p
    A = np.array(cols).T / zn_err.ravel()[:, None]
    y = zn.ravel() / zn_err.ravel()
    b, *_ = np.linalg.lstsq(A, y, rcond=None)
    return np.r_[b[0], b[1:] / b[0]]


POST_CENTRES = {
    # tutorial posterior means (reference docs/tutorials/*.ipynb)
    ('PeltonColeCole', 1): [1.02, 0.36, -2.1, 0.50],
    ('PeltonColeCole', 2): [1.01, 0.14, 0.93, -1.57, -12.8, 0.45, 0.61],
    ('PeltonColeCole', 3): [1.01, 0.14, 0.50, 0.40, -1.57, -12.8, -6.0, 0.45, 0.61, 0.5],
    ('Dias2000', 0): [1.02, 0.69, -10.3, 7.8, 0.71],
    ('Shin2015', 0): [0.5, 0.5, -14.0, -6.0, 0.3, 0.45],
}


def theta_sets(model, key, seed, n_prior=40, n_post=24):
    lo, hi = model.param_bounds
    lo = np.asarray(lo, float)
    hi = np.asarray(hi, float)
    rng = np.random.RandomState(seed)
    prior = rng.uniform(lo, hi, (n_prior, lo.size))
    if type(model).__name__ == 'PolynomialDecomposition':
        centre = polydecomp_lsq(model)
        scale = 2e-4 * np.maximum(np.abs(centre), 1e-6)
    else:
        centre = np.array(POST_CENTRES[key], float)
        scale = 0.01 * (hi - lo)
    post = centre + scale * rng.randn(n_post, lo.size)
    post[0] = centre
    post = np.clip(post, lo + 1e-9 * (hi - lo), hi - 1e-9 * (hi - lo))
    edges = edge_rows(lo, hi, post[0])
    return np.ascontiguousarray(np.vstack([prior, post, edges])), n_prior, n_post


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scratch', default='/tmp/bisip_ref_build')
    ap.add_argument('--only-extended', action='store_true',
                    help='regenerate only the ext*.npz cases (leaves the case*.npz files untouched)')
    ap.add_argument('--only-valley', action='store_true',
                    help='regenerate only the valley*.npz cases (leaves the others untouched)')
    args = ap.parse_args()

    pkg = build_reference(args.scratch)
    bisip = import_reference(pkg)
    sys.path.insert(0, REPO)
    from bisip_amd.synthetic import write_spectrum_file  # our generator (inputs only)

    datadir = os.path.join(pkg, 'bisip', 'data')
    bundled = {k: os.path.join(datadir, k + '.dat') for k in
               ['SIP-K389170', 'SIP-K389172', 'SIP-K389173', 'SIP-K389174',
                'SIP-K389175', 'SIP-K389176']}
    synth = {}
    for n_freq, idx in [(32, 0), (64, 0), (32, 7), (20, 3), (5, 11), (3, 12)]:
        name = f'synthetic-N{n_freq}-i{idx}'
        synth[name] = write_spectrum_file(os.path.join(args.scratch, name + '.dat'), n_freq, idx)
    base_synth = [k for k in synth if k not in ('synthetic-N5-i11', 'synthetic-N3-i12')]

    if args.only_extended:
        extended_cases(bisip, bundled, synth)
        return
    if args.only_valley:
        valley_cases(bisip, bundled, synth)
        return

    # ---- (1) load_data fixtures (pins SURVEY §8 a12) -------------------------------
    ld = {}
    probe = bisip.Dias2000.__new__(bisip.Dias2000)  # load_data is a plain mixin method
    for name, path in list(bundled.items()) + [(k, synth[k]) for k in base_synth]:
        for headers, units in [(1, 'mrad'), (9, 'mrad'), (1, 'rad'), (1, 'deg')]:
            if name.startswith('synthetic') and (headers != 1 or units != 'mrad'):
                continue
            d = probe.load_data(path, headers, units)
            tag = f'{name}|{headers}|{units}'
            ld[tag + '|raw'] = np.loadtxt(path, skiprows=1, delimiter=',')
            for k in ['w', 'zn', 'zn_err']:
                ld[tag + '|' + k] = np.asarray(d[k])
            ld[tag + '|norm_factor'] = np.float64(d['norm_factor'])
    np.savez_compressed(os.path.join(HERE, 'load_data.npz'), **ld)

    # ---- (2) model cases ------------------------------------------------------------
    cases = []
    k75 = bundled['SIP-K389175']
    for name, path in bundled.items():
        cases.append((name, path, 'PolynomialDecomposition', dict(poly_deg=5, c_exp=1.0)))
    cases += [
        ('SIP-K389175', k75, 'PolynomialDecomposition', dict(poly_deg=4, c_exp=1.0)),
        ('SIP-K389175', k75, 'PolynomialDecomposition', dict(poly_deg=4, c_exp=0.5)),
        ('SIP-K389175', k75, 'PolynomialDecomposition', dict(poly_deg=5, c_exp=0.73)),
        ('SIP-K389175', k75, 'PolynomialDecomposition', dict(poly_deg=3, c_exp=1.0)),
        ('synthetic-N32-i0', synth['synthetic-N32-i0'], 'PolynomialDecomposition', dict(poly_deg=5, c_exp=1.0)),
        ('synthetic-N32-i0', synth['synthetic-N32-i0'], 'PolynomialDecomposition', dict(poly_deg=5, c_exp=0.5)),
        ('synthetic-N32-i7', synth['synthetic-N32-i7'], 'PolynomialDecomposition', dict(poly_deg=5, c_exp=1.0)),
        ('synthetic-N64-i0', synth['synthetic-N64-i0'], 'PolynomialDecomposition', dict(poly_deg=5, c_exp=1.0)),
        ('synthetic-N20-i3', synth['synthetic-N20-i3'], 'PolynomialDecomposition', dict(poly_deg=5, c_exp=1.0)),
    ]
    for name, path in [('SIP-K389175', k75), ('SIP-K389172', bundled['SIP-K389172']),
                       ('synthetic-N32-i0', synth['synthetic-N32-i0']),
                       ('synthetic-N64-i0', synth['synthetic-N64-i0'])]:
        for nm in (1, 2, 3):
            cases.append((name, path, 'PeltonColeCole', dict(n_modes=nm)))
        cases.append((name, path, 'Dias2000', {}))
        cases.append((name, path, 'Shin2015', {}))

    manifest = []
    for ci, (dname, path, cls, kw) in enumerate(cases):
        model = getattr(bisip, cls)(path, nwalkers=32, nsteps=10, **kw)
        key = (cls, kw.get('n_modes', 0))
        theta, n_prior, n_post = theta_sets(model, key, seed=9000 + ci)
        Z, logp = evaluate(model, theta)
        out = dict(theta=theta, Z=Z, logp=logp,
                   w=model.data['w'], zn=model.data['zn'], zn_err=model.data['zn_err'],
                   bounds=np.asarray(model.param_bounds, float),
                   n_prior=np.int64(n_prior), n_post=np.int64(n_post),
                   raw=np.loadtxt(path, skiprows=1, delimiter=','))
        if cls == 'PolynomialDecomposition':
            out.update(log_tau=model.log_tau, log_taus=model.log_taus, taus=model.taus,
                       poly_deg=np.int64(model.poly_deg), c_exp=np.float64(model.c_exp))
        if cls == 'PeltonColeCole':
            out.update(n_modes=np.int64(model.n_modes))
        out['param_names'] = np.array(model.param_names)
        fname = f'case{ci:02d}_{cls}_{dname}.npz'
        np.savez_compressed(os.path.join(HERE, fname), **out)
        manifest.append(f'{fname}\t{cls}\t{dname}\t{kw}')
        finite = np.isfinite(logp)
        print(f'{fname}: rows={theta.shape[0]} finite={finite.sum()} '
              f'logp[min,max]=({logp[finite].min():.6g},{logp[finite].max():.6g})')
    with open(os.path.join(HERE, 'MANIFEST.tsv'), 'w') as fh:
        fh.write('\n'.join(manifest) + '\n')

    # ---- (3) anchors of SURVEY.md Appendix C, re-derived (not copied) -----------------
    anchors = [
        ('A1', 'PolynomialDecomposition', dict(poly_deg=4, c_exp=1.0),
         [0.997613, 0.006870, -0.003937, -0.001338, 0.000741, 0.000219]),
        ('A2', 'PolynomialDecomposition', dict(poly_deg=4, c_exp=0.5),
         [0.997613, 0.006870, -0.003937, -0.001338, 0.000741, 0.000219]),
        ('A3', 'PolynomialDecomposition', dict(), [1.0, 0.005, -0.003, -0.001, 0.0005, 0.0002, 0.00001]),
        ('A4', 'PeltonColeCole', dict(n_modes=1), [1.0, 0.3, -2.0, 0.5]),
        ('A5', 'PeltonColeCole', dict(n_modes=2), [1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]),
        ('A6', 'Dias2000', dict(), [1.0, 0.25, -10.0, 5.0, 0.5]),
        ('A7', 'Shin2015', dict(), [0.5, 0.5, -14.0, -6.0, 0.3, 0.45]),
    ]
    with open(os.path.join(HERE, 'anchors.tsv'), 'w') as fh:
        for aid, cls, kw, th in anchors:
            model = getattr(bisip, cls)(k75, nwalkers=32, nsteps=10, **kw)
            d = model.data
            lp = model._log_probability(np.array(th), model.forward, model.param_bounds,
                                        d['w'], d['zn'], d['zn_err'])
            fh.write(f'{aid}\t{cls}\t{kw}\t{th}\t{float(lp)!r}\n')
            print(aid, repr(lp))
    extended_cases(bisip, bundled, synth)
    valley_cases(bisip, bundled, synth)


def exact_logp(model, rows, digits=50):
    """The reference's formula -- Decomp_cyth (src/bisip/cython_funcs.pyx:75-94) + _log_likelihood
    (src/bisip/models.py:59-62) -- evaluated from the model's own double operands in `digits`-digit
    arithmetic (mpmath), rounded once at the end.  What the reference would return without rounding."""
    import mpmath as mp
    mp.mp.dps = digits
    d = model.data
    w = [mp.mpf(float(x)) for x in d['w']]
    taus = [mp.mpf(float(x)) for x in model.taus]
    L = model.log_taus
    c = mp.mpf(float(model.c_exp))
    K = [[1 - 1 / (1 + (mp.mpc(0, 1) * wj * tk) ** c) for tk in taus] for wj in w]       # C_Debye per unit m
    s2 = [[mp.mpf(float(d['zn_err'][p, j])) ** 2 for j in range(len(w))] for p in (0, 1)]
    const = mp.fsum(2 * mp.log(v) for row in s2 for v in row)
    out = []
    for th in rows:
        r0, a = mp.mpf(float(th[0])), [mp.mpf(float(x)) for x in th[1:]]
        M = [mp.fsum(a[i] * mp.mpf(float(L[i, k])) for i in range(len(a))) for k in range(len(taus))]
        tot = mp.mpf(0)
        for j in range(len(w)):
            Z = r0 * (1 - mp.fsum(M[k] * K[j][k] for k in range(len(taus))))
            tot += (mp.mpf(float(d['zn'][0, j])) - Z.real) ** 2 / s2[0][j] + (mp.mpf(float(d['zn'][1, j])) - Z.imag) ** 2 / s2[1][j]
        out.append(float(-(tot + const) / 2))
    return np.array(out)


def valley_cases(bisip, bundled, synth):
    """Rows along the flat valley of chi^2 of PolynomialDecomposition designs -- b = b_ls + s R^-1 z, a third
    at s = 1 ... 30 posterior sigmas, two thirds ON the shell log-probability = 0 -- with the REAL reference's
    log-probability AND the exact value of its formula (50 digits).  On nearly collinear designs the two differ
    by far more than the parity tolerance: these files record by how much, independently of this repository's
    kernels and oracle (only WHERE the rows lie comes from its host code).  Files valley*.npz."""
    sys.path.insert(0, os.path.join(REPO, 'benchmarks'))
    from bisip_amd import _hip
    from fuzz_parity import valley_rows
    cases = [('SIP-K389175', bundled['SIP-K389175'], dict(poly_deg=5, c_exp=1.0)),
             ('SIP-K389175', bundled['SIP-K389175'], dict(poly_deg=4, c_exp=0.5)),
             ('SIP-K389175', bundled['SIP-K389175'], dict(poly_deg=8, c_exp=1.0)),
             ('SIP-K389172', bundled['SIP-K389172'], dict(poly_deg=7, c_exp=0.5)),
             ('synthetic-N32-i0', synth['synthetic-N32-i0'], dict(poly_deg=5, c_exp=1.0)),
             ('synthetic-N32-i7', synth['synthetic-N32-i7'], dict(poly_deg=6, c_exp=1.0)),
             ('synthetic-N64-i0', synth['synthetic-N64-i0'], dict(poly_deg=9, c_exp=0.5)),
             ('synthetic-N20-i3', synth['synthetic-N20-i3'], dict(poly_deg=10, c_exp=1.0))]
    manifest = []
    for ci, (dname, path, kw) in enumerate(cases):
        model = bisip.PolynomialDecomposition(path, nwalkers=32, nsteps=10, **kw)
        d = model.data
        bounds = np.asarray(model.param_bounds, float)
        ops = _hip.polydecomp_operands(d['w'], d['zn'], d['zn_err'], model.taus, model.log_taus, model.c_exp)
        theta = np.ascontiguousarray(valley_rows(ops, bounds, np.random.RandomState(6000 + ci), 400)[:48])
        assert len(theta) >= 24, (dname, kw, len(theta))
        _, logp = evaluate(model, theta)
        exact = exact_logp(model, theta)
        out = dict(theta=theta, logp=logp, logp_exact=exact, w=d['w'], zn=d['zn'], zn_err=d['zn_err'], bounds=bounds,
                   log_tau=model.log_tau, log_taus=model.log_taus, taus=model.taus,
                   poly_deg=np.int64(model.poly_deg), c_exp=np.float64(model.c_exp),
                   raw=np.loadtxt(path, skiprows=1, delimiter=','), param_names=np.array(model.param_names))
        fname = f'valley{ci:02d}_PolynomialDecomposition_{dname}.npz'
        np.savez_compressed(os.path.join(HERE, fname), **out)
        off = np.abs(logp - exact) / np.maximum(1.0, np.abs(exact))
        manifest.append(f'{fname}\tPolynomialDecomposition\t{dname}\t{kw}\treference vs exact: max {off.max():.2e}')
        print(f'{fname}: rows={len(theta)} logp in ({exact.min():.4g}, {exact.max():.4g}); the reference is up to '
              f'{off.max():.2e} (relative to max(1, |logp|)) from the exact value of its formula')
    with open(os.path.join(HERE, 'MANIFEST_VALLEY.tsv'), 'w') as fh:
        fh.write('\n'.join(manifest) + '\n')


def extended_cases(bisip, bundled, synth):
    """Shapes beyond the reference's tutorials -- high and zero polynomial degree, small
    exponents, fewer data rows than unknowns, 4 and 5 Cole-Cole modes -- plus, for every
    case, forward() at on-bound values of EVERY parameter (where its complex arithmetic goes
    through 1/0 and inf^-1 and still returns finite numbers).  Files ext*.npz."""
    k75 = bundled['SIP-K389175']
    cases = [('SIP-K389175', k75, 'PolynomialDecomposition', dict(poly_deg=p, c_exp=c))
             for p, c in [(0, 1.0), (1, 0.5), (8, 1.0), (8, 0.3805172729737878), (9, 0.21152054037418078), (10, 1.0), (10, 0.5)]]
    cases += [('synthetic-N5-i11', synth['synthetic-N5-i11'], 'PolynomialDecomposition', dict(poly_deg=8, c_exp=0.3805172729737878)),
              ('synthetic-N3-i12', synth['synthetic-N3-i12'], 'PolynomialDecomposition', dict(poly_deg=10, c_exp=1.0)),
              ('synthetic-N64-i0', synth['synthetic-N64-i0'], 'PolynomialDecomposition', dict(poly_deg=9, c_exp=0.5)),
              ('SIP-K389175', k75, 'PeltonColeCole', dict(n_modes=4)),
              ('SIP-K389175', k75, 'PeltonColeCole', dict(n_modes=5)),
              ('synthetic-N5-i11', synth['synthetic-N5-i11'], 'PeltonColeCole', dict(n_modes=2)),
              ('SIP-K389173', bundled['SIP-K389173'], 'Dias2000', {}),
              ('SIP-K389173', bundled['SIP-K389173'], 'Shin2015', {}),
              ('synthetic-N3-i12', synth['synthetic-N3-i12'], 'Dias2000', {}),
              ('synthetic-N3-i12', synth['synthetic-N3-i12'], 'Shin2015', {})]
    manifest = []
    for ci, (dname, path, cls, kw) in enumerate(cases):
        model = getattr(bisip, cls)(path, nwalkers=32, nsteps=10, **kw)
        lo, hi = (np.asarray(b, float) for b in model.param_bounds)
        rng = np.random.RandomState(7000 + ci)
        n_prior, n_post = 40, 24
        prior = rng.uniform(lo, hi, (n_prior, lo.size))
        if cls == 'PolynomialDecomposition':
            centre = np.r_[1.0, np.zeros(lo.size - 1)]
            post = centre + np.r_[1e-3, np.full(lo.size - 1, 1e-4)] * rng.randn(n_post, lo.size)   # clouds of small coefficients
        else:
            centre = 0.5 * (lo + hi)
            post = centre + 0.05 * (hi - lo) * rng.randn(n_post, lo.size)
        post = np.clip(post, lo + 1e-9 * (hi - lo), hi - 1e-9 * (hi - lo))
        theta = np.ascontiguousarray(np.vstack([prior, post, edge_rows(lo, hi, post[0])]))
        Z, logp = evaluate(model, theta)
        # forward() at every parameter's bounds (one component at a time)
        base = 0.5 * (lo + hi) if cls != 'PolynomialDecomposition' else np.r_[1.0, np.full(lo.size - 1, 0.01)]
        fwd = []
        for q in range(lo.size):
            for v in (lo[q], hi[q]):
                r = base.copy()
                r[q] = v
                fwd.append(r)
        fwd = np.array(fwd)
        w = model.data['w']
        with np.errstate(all='ignore'):
            Zf = np.array([model.forward(r, w) for r in fwd])
        out = dict(theta=theta, Z=Z, logp=logp, theta_fwd_edges=fwd, Z_fwd_edges=Zf,
                   w=w, zn=model.data['zn'], zn_err=model.data['zn_err'],
                   bounds=np.asarray(model.param_bounds, float),
                   n_prior=np.int64(n_prior), n_post=np.int64(n_post),
                   raw=np.loadtxt(path, skiprows=1, delimiter=','))
        if cls == 'PolynomialDecomposition':
            out.update(log_tau=model.log_tau, log_taus=model.log_taus, taus=model.taus,
                       poly_deg=np.int64(model.poly_deg), c_exp=np.float64(model.c_exp))
        if cls == 'PeltonColeCole':
            out.update(n_modes=np.int64(model.n_modes))
        out['param_names'] = np.array(model.param_names)
        fname = f'ext{ci:02d}_{cls}_{dname}.npz'
        np.savez_compressed(os.path.join(HERE, fname), **out)
        manifest.append(f'{fname}\t{cls}\t{dname}\t{kw}')
        finite = np.isfinite(logp)
        print(f'{fname}: rows={theta.shape[0]} finite logp={finite.sum()} finite fwd-edge values={np.isfinite(Zf).mean():.2f} '
              f'logp[min,max]=({logp[finite].min():.6g},{logp[finite].max():.6g})')
    with open(os.path.join(HERE, 'MANIFEST_EXT.tsv'), 'w') as fh:
        fh.write('\n'.join(manifest) + '\n')


if __name__ == '__main__':
    main()
