"""GPU parity: the HIP kernels (through the C ABI) against the golden vectors the real
reference produced, and against the CPU oracle on fresh seeded inputs.

Tolerance: |dlogp| <= 1e-10*max(1,|logp|), |dZ| <= 1e-12*max(1,max|Z|);
-inf rows and row order must match exactly.
"""

import numpy as np
import pytest

from conftest import (assert_logp_close, assert_Z_close, case_id, case_model, golden_cases)

pytestmark = pytest.mark.gpu

MODEL_IDS = {'PolynomialDecomposition': 0, 'PeltonColeCole': 1, 'Dias2000': 2, 'Shin2015': 3}


def make_ctx(g, model, variant='auto'):
    from bisip_amd import _hip
    kw = {}
    if model == 'PolynomialDecomposition':
        kw = dict(poly_deg=int(g['poly_deg']), c_exp=float(g['c_exp']), taus=g['taus'],
                  log_taus=g['log_taus'])
    if model == 'PeltonColeCole':
        kw = dict(n_modes=int(g['n_modes']))
    return _hip.HipContext(MODEL_IDS[model], g['w'], g['zn'], g['zn_err'], g['bounds'],
                           variant=variant, **kw)


def variants_for(g, model):
    if model != 'PolynomialDecomposition':
        return ['auto']
    v = ['reduced', 'collapsed']
    if 3 <= int(g['poly_deg']) <= 5 and g['taus'].size <= 128:
        v.append('faithful')
    return v


@pytest.mark.parametrize('path', golden_cases(), ids=case_id)
def test_logprob_matches_reference_golden(path):
    g = np.load(path)
    model = case_model(path)
    for variant in variants_for(g, model):
        ctx = make_ctx(g, model, variant)
        got = ctx.logprob(g['theta'])
        err = assert_logp_close(got, g['logp'])
        print(f'{case_id(path)} [{ctx.kernel_name}] max rel err {err:.2e}')
        ctx.close()


@pytest.mark.parametrize('path', golden_cases(), ids=case_id)
def test_forward_matches_reference_golden(path):
    g = np.load(path)
    model = case_model(path)
    ctx = make_ctx(g, model)
    rows = np.all(np.isfinite(g['theta']), axis=1)
    got = ctx.forward(g['theta'][rows])
    assert_Z_close(got, g['Z'][rows])
    ctx.close()
