import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# Parity tolerance stated by BASELINE.md §3 / SURVEY.md §8c (fp64):
LOGP_RTOL = 1e-10   # |dlogp| <= 1e-10 * max(1, |logp_ref|)
Z_RTOL = 1e-12      # |dZ|    <= 1e-12 * max(1, max|Z_ref|)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def golden_cases():
    return sorted(glob.glob(os.path.join(GOLDEN, 'case*.npz')))


def extended_cases():
    """Shapes beyond the reference's tutorials (high / zero polynomial degree, small exponents,
    fewer data rows than unknowns, 4-5 Cole-Cole modes) with forward() at every parameter's
    bounds; written by the same script from the real reference."""
    return sorted(glob.glob(os.path.join(GOLDEN, 'ext*.npz')))


def valley_cases():
    """Rows along the flat valley of chi^2 / on the shell logp = 0 of PolynomialDecomposition designs, with the
    REAL reference's log-probability and the exact value of its formula (50-digit mpmath); same script."""
    return sorted(glob.glob(os.path.join(GOLDEN, 'valley*.npz')))


def case_model(path):
    return os.path.basename(path).split('_')[1]


def case_id(path):
    return os.path.basename(path)[:-4]


def assert_logp_close(got, want, rtol=LOGP_RTOL):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape
    ninf = np.isneginf(want)
    assert np.array_equal(np.isneginf(got), ninf), '-inf rows differ'
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), 'finite rows differ'
    if fin.any():
        err = np.abs(got[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))
        assert err.max() <= rtol, f'max rel err {err.max():.3e} > {rtol:g}'
        return float(err.max())
    return 0.0


def assert_Z_close(got, want, rtol=Z_RTOL):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape
    scale = np.maximum(1.0, np.nanmax(np.abs(want), axis=(-2, -1), keepdims=True))
    err = np.abs(got - want) / scale
    assert np.nanmax(err) <= rtol, f'max rel Z err {np.nanmax(err):.3e} > {rtol:g}'
    return float(np.nanmax(err))


@pytest.fixture(scope='session')
def hip_lib():
    from bisip_amd import _hip
    return _hip.load_library()
