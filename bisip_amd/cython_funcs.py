"""Call-compatible stand-ins for the reference's compiled forward functions.

The reference's models call four functions of its Cython module with one parameter vector
at a time (src/bisip/cython_funcs.pyx:49, :64, :75, :96; call sites src/bisip/models.py:228,
:269-271, :303-305, :347-349).  Code written against those names keeps working on top of the
HIP forward kernels: same names, positional/keyword arguments and ``(2, N)`` float64 result,
same exceptions for a wrong dtype or rank.  Each call is one ``bisip_forward`` launch on one
row, so this is a convenience for existing scripts, not a fast path -- batch work goes through
``model.forward(theta_rows)`` or ``bisip_forward`` directly.
"""
from collections import OrderedDict

import numpy as np

from . import _hip

_MODEL_PD, _MODEL_CC, _MODEL_DIAS, _MODEL_SHIN = 0, 1, 2, 3
_contexts = OrderedDict()   # (model_id, w bytes, desc bytes) -> HipContext
_MAX_CONTEXTS = 8


def _vec(name, a, ndim=1):
    """The typed-buffer checks Cython performs at the ``def`` boundary
    (``np.ndarray[DTYPE_t, ndim=1]``): not an array -> TypeError; wrong dtype or rank ->
    ValueError."""
    if not isinstance(a, np.ndarray):
        raise TypeError(f"Argument '{name}' has incorrect type (expected numpy.ndarray, "
                        f"got {type(a).__name__})")
    if a.dtype != np.float64:
        raise ValueError(f"Buffer dtype mismatch, expected 'DTYPE_t' but got '{a.dtype}'")
    if a.ndim != ndim:
        raise ValueError(f'Buffer has wrong number of dimensions (expected {ndim}, got {a.ndim})')
    return np.ascontiguousarray(a)


def _forward(model_id, w, theta, **desc):
    key = (model_id, w.tobytes(),
           tuple(sorted((k, np.asarray(v).tobytes()) for k, v in desc.items())))
    ctx = _contexts.get(key)
    if ctx is None:
        ones = np.ones((2, w.size))
        open_box = np.array([np.full(theta.size, -np.inf), np.full(theta.size, np.inf)])
        ctx = _hip.HipContext(model_id, w, ones, ones, open_box, **desc)
        if len(_contexts) >= _MAX_CONTEXTS:
            _contexts.popitem(last=False)[1].close()
        _contexts[key] = ctx
    else:
        _contexts.move_to_end(key)
    return ctx.forward(theta[None, :])[0]


def ColeCole_cyth(w, R0, m, lt, c):
    """Pelton Cole-Cole impedance, ``len(m)`` modes (src/bisip/cython_funcs.pyx:49-62)."""
    w, m, lt, c = _vec('w', w), _vec('m', m), _vec('lt', lt), _vec('c', c)
    if not (m.size == lt.size == c.size):
        raise ValueError('m, lt and c must have one entry per mode')
    theta = np.concatenate(([float(R0)], m, lt, c))
    return _forward(_MODEL_CC, w, theta, n_modes=m.size)


def Dias2000_cyth(w, R0, m, log_tau, eta, delta):
    """Dias (2000) impedance (src/bisip/cython_funcs.pyx:64-73)."""
    w = _vec('w', w)
    theta = np.array([R0, m, log_tau, eta, delta], dtype=np.float64)
    return _forward(_MODEL_DIAS, w, theta)


def Decomp_cyth(w, taus, log_taus, c_exp, R0, a):
    """Debye / Warburg polynomial decomposition (src/bisip/cython_funcs.pyx:75-94).
    ``log_taus`` is ``(len(a), len(taus))``, coefficients ``a`` ascending."""
    w, taus, a = _vec('w', w), _vec('taus', taus), _vec('a', a)
    log_taus = _vec('log_taus', log_taus, ndim=2)
    if log_taus.shape != (a.size, taus.size):
        raise ValueError(f'log_taus must have shape ({a.size}, {taus.size}), got {log_taus.shape}')
    theta = np.concatenate(([float(R0)], a))
    return _forward(_MODEL_PD, w, theta, poly_deg=a.size - 1, c_exp=float(c_exp), taus=taus,
                    log_taus=log_taus)


def Shin2015_cyth(w, R, log_Q, n):
    """Shin (2015) two-CPE impedance (src/bisip/cython_funcs.pyx:96-108)."""
    w, R, log_Q, n = _vec('w', w), _vec('R', R), _vec('log_Q', log_Q), _vec('n', n)
    if not (R.size == log_Q.size == n.size == 2):
        raise ValueError('R, log_Q and n must each hold two values')
    theta = np.concatenate((R, log_Q, n))
    return _forward(_MODEL_SHIN, w, theta)
