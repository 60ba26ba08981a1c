"""Synthetic SIP spectra for benchmarks and parity tests.

The generator is the one SURVEY.md §8(d) pins for every config marked
"synthetic": a noisy double Cole-Cole truth on a log-spaced frequency grid that
spans the bundled instrument range (first/last rows of
``src/bisip/data/SIP-K389175.dat`` in the reference), written in the 5-column
layout the reference's ``load_data`` reads (``src/bisip/utils.py:120-122``).

Nothing here runs on the GPU; it only produces host inputs.
"""

import numpy as np

# truth near the tutorial posterior (reference docs/tutorials/pelton.ipynb:346-418)
TRUTH_R0 = 1.0
TRUTH_M = (0.15, 0.5)
TRUTH_LN_TAU = (-1.5, -12.0)
TRUTH_C = (0.45, 0.6)
TRUTH_SCALE = 4.0e4  # ohm-m

F_MAX = 6e3
F_MIN = 1.1444e-2


def synthetic_columns(n_freq, spectrum_index=0):
    """Return the raw (n_freq, 5) table: freq, amp, pha[mrad], amp_err, pha_err[mrad].

    Draw order from ``RandomState(1234 + spectrum_index)``: amplitude noise
    (randn), phase noise (randn), phase error (rand).
    """
    f = np.logspace(np.log10(F_MAX), np.log10(F_MIN), n_freq)
    w = 2 * np.pi * f
    z = np.zeros(n_freq, dtype=np.complex128)
    for m, lt, c in zip(TRUTH_M, TRUTH_LN_TAU, TRUTH_C):
        z += m * (1.0 - 1.0 / (1.0 + (1j * w * np.exp(lt)) ** c))
    z = TRUTH_SCALE * TRUTH_R0 * (1.0 - z)

    rng = np.random.RandomState(1234 + int(spectrum_index))
    amp = np.abs(z) * (1.0 + 0.002 * rng.randn(n_freq))
    pha = 1e3 * np.angle(z) + 0.2 * rng.randn(n_freq)
    amp_err = 0.02 * amp
    pha_err = 0.5 + 2.0 * rng.rand(n_freq)
    return np.column_stack([f, amp, pha, amp_err, pha_err])


def write_spectrum_file(path, n_freq, spectrum_index=0):
    """Write one synthetic spectrum as a comma-separated file with one header line."""
    cols = synthetic_columns(n_freq, spectrum_index)
    np.savetxt(path, cols, delimiter=',', fmt='%.18e',
               header='freq, amp, pha, amp_err, pha_err', comments='')
    return path


def synthetic_theta(lo, hi, n_walkers, seed=2024):
    """Walker positions of SURVEY.md §8(d): uniform in the open prior box."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    rng = np.random.RandomState(seed)
    return np.ascontiguousarray(rng.uniform(lo, hi, (int(n_walkers), lo.size)))
