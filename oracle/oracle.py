"""ctypes front end of ``libbisip_oracle.so`` plus a NumPy restatement.

TEST INFRASTRUCTURE ONLY (see ``bisip_oracle.c``).  Parity pin: the C library and
the NumPy restatement below are both checked against ``tests/golden/*.npz``
(outputs of the real reference) by ``tests/test_oracle_golden.py``.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libbisip_oracle.so')
_lib = None

MODEL_IDS = {'PolynomialDecomposition': 0, 'PeltonColeCole': 1, 'Dias2000': 2, 'Shin2015': 3}

_dp = ctypes.POINTER(ctypes.c_double)


class _CProblem(ctypes.Structure):
    _fields_ = [('model_id', ctypes.c_int), ('N', ctypes.c_int), ('ndim', ctypes.c_int),
                ('w', _dp), ('zn', _dp), ('zn_err', _dp), ('lo', _dp), ('hi', _dp),
                ('S', ctypes.c_int), ('D', ctypes.c_int), ('c_exp', ctypes.c_double),
                ('taus', _dp), ('log_taus', _dp), ('n_modes', ctypes.c_int)]


def build_oracle(force=False):
    """Compile the C restatement with the recipe in oracle/Makefile."""
    src = os.path.join(_HERE, 'bisip_oracle.c')
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'libbisip_oracle.so'])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build_oracle()
        lib = ctypes.CDLL(_LIB_PATH)
        lib.oracle_logprob_batch.argtypes = [ctypes.POINTER(_CProblem), _dp, ctypes.c_int64,
                                             _dp, ctypes.c_int]
        lib.oracle_logprob_batch.restype = ctypes.c_int
        lib.oracle_forward_batch.argtypes = [ctypes.POINTER(_CProblem), _dp, ctypes.c_int64, _dp]
        lib.oracle_forward_batch.restype = ctypes.c_int
        lib.oracle_log_prior.argtypes = [ctypes.c_int, _dp, _dp, _dp]
        lib.oracle_log_prior.restype = ctypes.c_double
        lib.oracle_log_likelihood.argtypes = [ctypes.c_int, _dp, _dp, _dp]
        lib.oracle_log_likelihood.restype = ctypes.c_double
        lib.oracle_max_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


class OracleProblem:
    """Walker-independent operands of one inversion (the ``args`` tuple the reference
    hands emcee at src/bisip/models.py:108-109, plus the model's precompute)."""

    def __init__(self, model, w, zn, zn_err, bounds, taus=None, log_taus=None, c_exp=1.0,
                 n_modes=0):
        self.model = model
        self.w = _c(w)
        self.zn = _c(zn).reshape(2, -1)
        self.zn_err = _c(zn_err).reshape(2, -1)
        b = _c(bounds).reshape(2, -1)
        self.lo, self.hi = _c(b[0]), _c(b[1])
        self.N = self.w.size
        self.ndim = self.lo.size
        self.taus = _c(taus) if taus is not None else np.zeros(1)
        self.log_taus = _c(log_taus) if log_taus is not None else np.zeros((1, 1))
        self.c_exp = float(c_exp)
        self.n_modes = int(n_modes)
        self._c = _CProblem(MODEL_IDS[model], self.N, self.ndim, _p(self.w), _p(self.zn),
                            _p(self.zn_err), _p(self.lo), _p(self.hi),
                            self.taus.size, self.log_taus.shape[0], self.c_exp,
                            _p(self.taus), _p(self.log_taus), self.n_modes)

    @classmethod
    def from_golden(cls, g, model):
        kw = {}
        if model == 'PolynomialDecomposition':
            kw = dict(taus=g['taus'], log_taus=g['log_taus'], c_exp=float(g['c_exp']))
        if model == 'PeltonColeCole':
            kw = dict(n_modes=int(g['n_modes']))
        return cls(model, g['w'], g['zn'], g['zn_err'], g['bounds'], **kw)


def logprob(problem, theta, n_threads=1):
    """theta (W, ndim) -> logp (W,) through the C restatement."""
    lib = _load()
    theta = _c(theta).reshape(-1, problem.ndim)
    out = np.empty(theta.shape[0])
    rc = lib.oracle_logprob_batch(ctypes.byref(problem._c), _p(theta), theta.shape[0], _p(out),
                                  int(n_threads))
    if rc != 0:
        raise RuntimeError('oracle_logprob_batch failed')
    return out


def forward(problem, theta):
    """theta (W, ndim) -> Z (W, 2, N) through the C restatement."""
    lib = _load()
    theta = _c(theta).reshape(-1, problem.ndim)
    out = np.empty((theta.shape[0], 2, problem.N))
    rc = lib.oracle_forward_batch(ctypes.byref(problem._c), _p(theta), theta.shape[0], _p(out))
    if rc != 0:
        raise RuntimeError('oracle_forward_batch failed')
    return out


def log_prior(theta, bounds):
    lib = _load()
    theta = _c(theta)
    b = _c(bounds).reshape(2, -1)
    lo, hi = _c(b[0]), _c(b[1])
    return lib.oracle_log_prior(theta.size, _p(theta), _p(lo), _p(hi))


def log_likelihood(Z, zn, zn_err):
    lib = _load()
    Z, zn, zn_err = _c(Z), _c(zn), _c(zn_err)
    return lib.oracle_log_likelihood(Z.size // 2, _p(Z), _p(zn), _p(zn_err))


def max_threads():
    return int(_load().oracle_max_threads())


# ----------------------------------------------------------------------------------
# NumPy restatement (small cases; mirrors the reference's Python layer literally)
# ----------------------------------------------------------------------------------

def numpy_forward(problem, th):
    """One walker.  src/bisip/cython_funcs.pyx:33-108 in NumPy complex128."""
    w = problem.w
    th = np.asarray(th, dtype=np.float64)
    jw = 1j * w
    if problem.model == 'PolynomialDecomposition':
        # Decomp_cyth, src/bisip/cython_funcs.pyx:75-94
        M = np.zeros(problem.taus.size)
        for i in range(problem.log_taus.shape[0]):
            M = M + th[1 + i] * problem.log_taus[i]
        z = np.zeros(w.size, dtype=np.complex128)
        for k in range(problem.taus.size):
            z += M[k] * (1 - 1.0 / (1 + (jw * problem.taus[k]) ** problem.c_exp))
        z = th[0] * (1 - z)
    elif problem.model == 'PeltonColeCole':
        # ColeCole_cyth, src/bisip/cython_funcs.pyx:49-62
        D = problem.n_modes
        m, lt, c = th[1:1 + D], th[1 + D:1 + 2 * D], th[1 + 2 * D:]
        z = np.zeros(w.size, dtype=np.complex128)
        for i in range(D):
            z += m[i] * (1.0 - 1.0 / (1.0 + (jw * np.exp(lt[i])) ** c[i]))
        z = th[0] * (1 - z)
    elif problem.model == 'Dias2000':
        # C_Dias, src/bisip/cython_funcs.pyx:36-40
        R0, m, log_tau, eta, delta = th
        tau_p = np.exp(log_tau) * (1 / delta - 1) / (1 - m)
        tau_pp = np.exp(log_tau) ** 2 * eta ** 2
        mu = jw * np.exp(log_tau) + (jw * tau_pp) ** 0.5
        z = R0 * (1 - m * (1 - 1.0 / (1 + jw * tau_p * (1 + 1 / mu))))
    else:
        # C_Shin, src/bisip/cython_funcs.pyx:42-44, 96-108
        R, log_Q, n = th[:2], th[2:4], th[4:]
        z = np.zeros(w.size, dtype=np.complex128)
        for i in range(2):
            z_cpe = 1 / (np.exp(log_Q[i]) * jw ** n[i])
            z += (1 / z_cpe + 1 / R[i]) ** -1
    return np.array([z.real, z.imag])


def numpy_logprob(problem, th):
    """One walker.  src/bisip/models.py:59-76 verbatim in structure."""
    th = np.asarray(th, dtype=np.float64)
    if not ((problem.lo < th).all() and (th < problem.hi).all()):
        return -np.inf
    sigma2 = problem.zn_err ** 2
    return -0.5 * np.sum((problem.zn - numpy_forward(problem, th)) ** 2 / sigma2
                         + 2 * np.log(sigma2))
