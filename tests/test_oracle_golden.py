"""The oracle pinned against the REAL reference.

tests/golden/*.npz were written by tests/golden/make_golden.py, which imports the
reference (its .pyx cythonized out of tree) and records inputs + outputs.  These
tests need neither the reference nor a GPU.

The C restatement reproduces the reference bit for bit on the machine that made the
fixtures; glibc selects libm variants per CPU (ifunc), so the asserted bound is
1e-13 relative rather than equality -- three orders below the parity tolerance.
"""

import os

import numpy as np
import pytest

import oracle
from conftest import (GOLDEN, assert_logp_close, assert_Z_close, case_id, case_model, extended_cases,
                      golden_cases, valley_cases)

ORACLE_RTOL = 1e-13


@pytest.mark.parametrize('path', golden_cases(), ids=case_id)
def test_c_oracle_logprob_and_forward(path):
    g = np.load(path)
    model = case_model(path)
    prob = oracle.OracleProblem.from_golden(g, model)
    got = oracle.logprob(prob, g['theta'])
    assert_logp_close(got, g['logp'], ORACLE_RTOL)
    rows = np.all(np.isfinite(g['theta']), axis=1)
    Z = oracle.forward(prob, g['theta'][rows])
    assert_Z_close(Z, g['Z'][rows], 1e-14)
    # threaded evaluation (the cpu_baseline path) returns the same numbers, same order
    assert np.array_equal(oracle.logprob(prob, g['theta'], n_threads=4), got)


@pytest.mark.parametrize('path', extended_cases(), ids=case_id)
def test_c_oracle_on_extended_shapes(path):
    """High / zero polynomial degree, tiny exponents, 2N < P+2, 4-5 modes; and forward() at
    on-bound values of every parameter, where the reference's complex arithmetic passes through
    1/0 and inf^-1 and still returns finite numbers (Dias delta = 0, m = 1; Shin R = 0)."""
    g = np.load(path)
    model = case_model(path)
    prob = oracle.OracleProblem.from_golden(g, model)
    assert_logp_close(oracle.logprob(prob, g['theta']), g['logp'], ORACLE_RTOL)
    rows = np.all(np.isfinite(g['theta']), axis=1)
    assert_Z_close(oracle.forward(prob, g['theta'][rows]), g['Z'][rows], 1e-14)
    assert np.all(np.isfinite(g['Z_fwd_edges']))
    assert_Z_close(oracle.forward(prob, g['theta_fwd_edges']), g['Z_fwd_edges'], 1e-14)


@pytest.mark.parametrize('path', valley_cases(), ids=case_id)
def test_valley_rows_reference_oracle_and_exact_value(path):
    """Where the reference's own arithmetic stops being accurate.  For rows along the valley of chi^2 /
    on the shell logp = 0 the fixtures hold the REAL reference's log-probability and the exact value of its
    formula (50 digits).  (1) The oracle is still the reference, bit for bit.  (2) The host yardstick
    (bisip_polydecomp_reduced_reference: what the reduced kernels are estimated, checked and guarded
    against; operands and evaluation in binary128) is the exact value to 1e-12 on EVERY design here --
    the tolerance tests/test_gpu_parity.py holds it to -- including the worst-conditioned one (degree 9,
    64 frequencies, c = 0.5: terms 6e7 times the row sums).  (3) The reference is up to 8e-9 from the
    exact value on the degree 7-10 designs -- eighty times the parity tolerance -- and within 4e-12 on
    degree <= 6."""
    from bisip_amd import _hip
    g = np.load(path)
    prob = oracle.OracleProblem.from_golden(g, 'PolynomialDecomposition')
    assert_logp_close(oracle.logprob(prob, g['theta']), g['logp'], ORACLE_RTOL)
    exact = g['logp_exact']
    scale = np.maximum(1.0, np.abs(exact))
    mine = _hip.polydecomp_reduced_reference(g['w'], g['zn'], g['zn_err'], g['taus'], g['log_taus'], float(g['c_exp']), g['theta'])
    P = int(g['poly_deg'])
    mine_off = float(np.max(np.abs(mine - exact) / scale))
    assert mine_off <= 1e-12
    ref_off = float(np.max(np.abs(g['logp'] - exact) / scale))
    assert ref_off <= 4e-12 or mine_off <= ref_off / 20
    print(f'{case_id(path)}: degree {P}: reference {ref_off:.1e}, yardstick {mine_off:.1e} from the exact value')
    assert ref_off <= 4e-12 if P <= 6 else ref_off > 4e-11


@pytest.mark.parametrize('path', golden_cases()[::3], ids=case_id)
def test_numpy_restatement(path):
    g = np.load(path)
    model = case_model(path)
    prob = oracle.OracleProblem.from_golden(g, model)
    with np.errstate(all='ignore'):
        got = np.array([oracle.numpy_logprob(prob, t) for t in g['theta']])
    assert_logp_close(got, g['logp'], ORACLE_RTOL)


def test_prior_edges_are_minus_inf():
    """Open box: on-bound, outside, NaN and +-inf rows give -inf (reference
    src/bisip/models.py:64-69); the last 8 rows of every case are such edge rows."""
    for path in golden_cases():
        g = np.load(path)
        logp = g['logp']
        assert np.isneginf(logp[-8:-1]).all(), path
        assert np.isfinite(logp[-1]), path
        b = g['bounds']
        for row in g['theta'][-8:]:
            assert oracle.log_prior(row, b) == (0.0 if (np.all(b[0] < row) and np.all(row < b[1])) else -np.inf)


def test_likelihood_pieces():
    g = np.load(golden_cases()[4])
    prob = oracle.OracleProblem.from_golden(g, 'PolynomialDecomposition')
    i = int(g['n_prior'])  # first near-posterior row
    ll = oracle.log_likelihood(g['Z'][i], g['zn'], g['zn_err'])
    assert abs(ll - g['logp'][i]) <= 1e-13 * max(1, abs(g['logp'][i]))
    # the constant is 2*ln(sigma^2), not ln(2*pi*sigma^2)  (src/bisip/models.py:62)
    const = -0.5 * np.sum(2 * np.log(g['zn_err'] ** 2))
    chi2 = np.sum((g['zn'] - g['Z'][i]) ** 2 / g['zn_err'] ** 2)
    assert abs((const - 0.5 * chi2) - ll) <= 1e-12 * max(1, abs(ll))


def test_anchor_values():
    """SURVEY.md Appendix C anchors, regenerated by make_golden.py into anchors.tsv."""
    import ast
    from bisip_amd.utils import load_data
    from bisip_amd import DataFiles
    d = load_data(DataFiles()['SIP-K389175'])
    w = d['w']
    for line in open(os.path.join(GOLDEN, 'anchors.tsv')):
        aid, cls, kw, th, lp = line.rstrip('\n').split('\t')
        kw, th, lp = ast.literal_eval(kw), np.array(ast.literal_eval(th)), float(lp)
        okw = {}
        if cls == 'PolynomialDecomposition':
            P = kw.get('poly_deg', 5)
            per = np.log10(1. / w)
            log_tau = np.linspace(np.floor(per.min() - 1), np.floor(per.max() + 1), 2 * w.size)
            okw = dict(taus=10 ** log_tau, log_taus=np.array([log_tau ** i for i in range(P + 1)]),
                       c_exp=kw.get('c_exp', 1.0))
            bounds = np.array([[0.9] + [-1] * (P + 1), [1.1] + [1] * (P + 1)], float)
        elif cls == 'PeltonColeCole':
            D = kw['n_modes']
            okw = dict(n_modes=D)
            bounds = np.array([[0.9] + [0] * D + [-15] * D + [0] * D, [1.1] + [1] * D + [5] * D + [1] * D], float)
        elif cls == 'Dias2000':
            bounds = np.array([[0.9, 0, -20, 0, 0], [1.1, 1, 0, 150, 1]], float)
        else:
            bounds = np.array([[0, 0, -15, -7, 0, 0], [1, 1, -13, -5, 1, 1]], float)
        prob = oracle.OracleProblem(cls, w, d['zn'], d['zn_err'], bounds, **okw)
        got = oracle.logprob(prob, th[None, :])[0]
        assert abs(got - lp) <= ORACLE_RTOL * max(1, abs(lp)), (aid, got, lp)


def test_load_data_matches_reference():
    """bisip_amd.utils.load_data against the reference's load_data outputs (a12)."""
    from bisip_amd.utils import columns_to_data
    ld = np.load(os.path.join(GOLDEN, 'load_data.npz'))
    tags = sorted({k.rsplit('|', 1)[0] for k in ld.files})
    assert len(tags) >= 20
    for tag in tags:
        name, headers, units = tag.split('|')
        raw = ld[tag + '|raw'][int(headers) - 1:]
        d = columns_to_data(raw, units)
        for k in ('w', 'zn', 'zn_err'):
            np.testing.assert_allclose(d[k], ld[tag + '|' + k], rtol=2e-16, atol=0)
        assert abs(d['norm_factor'] - float(ld[tag + '|norm_factor'])) <= 1e-16 * d['norm_factor']


def test_load_data_from_file_and_spot_values():
    from bisip_amd import DataFiles
    from bisip_amd.utils import load_data
    files = DataFiles()
    assert sorted(files) == ['SIP-K389170', 'SIP-K389172', 'SIP-K389173', 'SIP-K389174',
                             'SIP-K389175', 'SIP-K389176']
    d = load_data(files['SIP-K389175'])
    assert d['N'] == 20
    assert abs(d['norm_factor'] - 41229.19) < 1e-6
    assert d['w'][0] == 37699.11184307752
    assert abs(d['zn'][0, 0] - 0.7837583862295252) < 1e-15
    assert abs(d['zn_err'][1, 0] - 0.007431786404572684) < 1e-17
    d9 = load_data(files['SIP-K389172'], headers=9)
    assert d9['N'] == 12


def test_load_data_batch_stacks_the_references_operands(tmp_path):
    """load_data_batch (the ingest of a batch context, used by SpectraBatch): every row is the
    reference's per-file load_data output; paths and raw tables mix; ragged batches raise."""
    import bisip_amd
    from bisip_amd.utils import load_data_batch
    ld = np.load(os.path.join(GOLDEN, 'load_data.npz'))
    files = bisip_amd.DataFiles()
    names = sorted(files)
    mixed = [files[n] if i % 2 else np.loadtxt(files[n], skiprows=1, delimiter=',')
             for i, n in enumerate(names)]
    b = load_data_batch(mixed)
    assert b['w'].shape == (6, 20) and b['zn'].shape == (6, 2, 20) and b['N'] == 20
    for i, n in enumerate(names):
        for key in ('w', 'zn', 'zn_err'):
            assert np.allclose(b[key][i], ld[f'{n}|1|mrad|{key}'], rtol=2e-16, atol=0)
        assert b['norm_factor'][i] == float(ld[f'{n}|1|mrad|norm_factor'])
    b9 = load_data_batch([files[n] for n in names], headers=9, ph_units='deg')
    assert b9['N'] == 12 and b9['w'].shape == (6, 12)
    with pytest.raises(ValueError, match='different frequency counts'):
        load_data_batch([files[names[0]], np.loadtxt(files[names[1]], skiprows=9, delimiter=',')])
    with pytest.raises(ValueError):
        load_data_batch([])



def test_batch_ingest_is_bitwise_the_per_file_path(tmp_path):
    """load_data_batch = the C parser (bisip_read_tables) + arithmetic batched over the spectra; it must
    give, bit for bit, what load_data (np.loadtxt + the reference's per-file arithmetic, pinned above)
    gives for every file -- for each phase unit, for files with comments, blank lines, CR LF, signs,
    exponents and extra columns, and for files only np.loadtxt understands (which the parser hands back)."""
    import bisip_amd
    from bisip_amd import _hip
    from bisip_amd.synthetic import synthetic_columns
    from bisip_amd.utils import load_data, load_data_batch
    rng = np.random.RandomState(3)
    paths = []
    for i in range(40):
        cols = synthetic_columns(24, i)
        cols[:, 2] *= rng.choice([-1.0, 1.0])
        p = tmp_path / f's{i:02d}.csv'
        style = i % 5
        if style == 0:                    # np.savetxt's %.18e
            np.savetxt(p, cols, delimiter=',', header='freq,amp,pha,amp_err,pha_err')
        elif style == 1:                  # shortest round-trip decimals, CR LF, blanks around numbers
            p.write_bytes(('f,a,p,ae,pe\r\n' + '\r\n'.join(' , '.join(repr(float(v)) for v in r) for r in cols) + '\r\n').encode())
        elif style == 2:                  # comments, blank lines, explicit plus signs, no final newline
            lines = ['header']
            for k, r in enumerate(cols):
                lines.append(','.join(('+' if v >= 0 else '') + f'{v:.17g}' for v in r) + ('  # row %d' % k if k % 3 == 0 else ''))
                if k % 7 == 0:
                    lines += ['', '# a comment line']
            p.write_text('\n'.join(lines))
        elif style == 3:                  # extra columns are ignored
            np.savetxt(p, np.hstack([cols, rng.rand(24, 2)]), delimiter=',', header='x')
        else:                             # few significant digits
            np.savetxt(p, cols, delimiter=',', header='x', fmt='%.6g')
        paths.append(str(p))
    # the parser alone against np.loadtxt
    tabs, status = _hip.read_tables(paths, 1, 24, threads=3)
    assert not status.any()
    for p, t in zip(paths, tabs):
        assert np.array_equal(t, np.loadtxt(p, skiprows=1, delimiter=',')[:, :5])
    # whole ingest against the per-file path, every unit
    for units in ('mrad', 'rad', 'deg'):
        b = load_data_batch(paths, headers=1, ph_units=units, threads=4)
        for i, p in enumerate(paths):
            d = load_data(p, 1, units)
            for key in ('w', 'zn', 'zn_err'):
                assert np.array_equal(b[key][i], d[key]), (units, i, key)
            assert b['norm_factor'][i] == d['norm_factor']
    files = bisip_amd.DataFiles()
    for headers, units in ((1, 'mrad'), (9, 'deg')):
        b = load_data_batch([files[n] for n in sorted(files)], headers, units)
        for i, n in enumerate(sorted(files)):
            d = load_data(files[n], headers, units)
            assert all(np.array_equal(b[k][i], d[k]) for k in ('w', 'zn', 'zn_err'))
    # what only np.loadtxt understands is handed back (status 1) and still loads ...
    odd = tmp_path / 'odd.csv'
    cols = synthetic_columns(24, 99)
    odd.write_text('h\n' + '\n'.join(','.join('nan' if (k == 5 and c == 4) else f'{v:.17g}' for c, v in enumerate(r))
                                      for k, r in enumerate(cols)))
    _, st = _hip.read_tables([paths[0], str(odd), str(tmp_path / 'missing.csv')], 1, 24)
    assert list(st) == [0, 1, 1]
    b = load_data_batch([paths[0], str(odd)])
    assert np.isnan(b['zn_err'][1]).any() and np.array_equal(b['w'][1], load_data(str(odd))['w'])
    # ... and what np.loadtxt rejects fails the same way (a row of blanks is a row to np.loadtxt)
    for name, text in (('ragged.csv', 'h\n1,2,3,4,5\n1,2,3,4\n' + '1,2,3,4,5\n' * 22),
                       ('blanks.csv', 'h\n1,2,3,4,5\n   \n' + '1,2,3,4,5\n' * 23),
                       ('words.csv', 'h\n1,2,3,4,five\n' + '1,2,3,4,5\n' * 23)):
        bad = tmp_path / name
        bad.write_text(text)
        with pytest.raises(ValueError):
            np.loadtxt(str(bad), skiprows=1, delimiter=',')
        assert _hip.read_tables([str(bad)], 1, 24)[1][0] == 1
        with pytest.raises(ValueError):
            load_data_batch([paths[0], str(bad)])
    with pytest.raises(ValueError, match='different frequency counts'):
        short = tmp_path / 'short.csv'
        np.savetxt(short, synthetic_columns(20, 1), delimiter=',', header='x')
        load_data_batch([paths[0], str(short)])
    with pytest.raises((OSError, ValueError)):
        load_data_batch([paths[0], str(tmp_path / 'missing.csv')])
