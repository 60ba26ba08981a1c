"""ctypes binding of ``libbisip_hip.so`` (C ABI in ``include/bisip_hip.h``).

This is the only compute backend of the package.  There is no CPU fallback: if the
shared library is missing or no MI355X is visible, calls raise ``RuntimeError``.
"""

import ctypes
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libbisip_hip.so')

MODEL_POLYDECOMP, MODEL_COLECOLE, MODEL_DIAS2000, MODEL_SHIN2015 = 0, 1, 2, 3
VARIANT_AUTO, VARIANT_FAITHFUL, VARIANT_COLLAPSED, VARIANT_REDUCED, VARIANT_WAVE, VARIANT_REDUCED_COMP = 0, 1, 2, 3, 4, 5
VARIANTS = {'auto': VARIANT_AUTO, 'faithful': VARIANT_FAITHFUL, 'collapsed': VARIANT_COLLAPSED,
            'reduced': VARIANT_REDUCED, 'wave': VARIANT_WAVE, 'reduced_comp': VARIANT_REDUCED_COMP}

_dp = ctypes.POINTER(ctypes.c_double)


class ModelDesc(ctypes.Structure):
    """``bisip_model_desc`` of include/bisip_hip.h."""
    _fields_ = [('n_modes', ctypes.c_int), ('poly_deg', ctypes.c_int), ('n_taus', ctypes.c_int),
                ('c_exp', ctypes.c_double), ('taus', _dp), ('log_taus', _dp)]


class StretchArgs(ctypes.Structure):
    """``bisip_stretch_args`` of include/bisip_hip.h (device pointers as integers)."""
    _fields_ = [('coords', ctypes.c_void_p), ('logp', ctypes.c_void_p),
                ('active', ctypes.c_void_p), ('partner', ctypes.c_void_p),
                ('zz', ctypes.c_void_p), ('factor', ctypes.c_void_p), ('logu', ctypes.c_void_p),
                ('n_slots', ctypes.c_int64), ('slot_lo', ctypes.c_int64), ('slot_hi', ctypes.c_int64),
                ('block', ctypes.c_void_p), ('chain_row', ctypes.c_void_p),
                ('logp_row', ctypes.c_void_p), ('naccept', ctypes.c_void_p),
                ('status', ctypes.c_void_p), ('pad', ctypes.c_int64), ('world', ctypes.c_int32),
                ('walkers_per_spectrum', ctypes.c_int64)]


class PersistArgs(ctypes.Structure):
    """``bisip_persist_args`` of include/bisip_hip.h."""
    _fields_ = [('coords', ctypes.c_void_p), ('logp', ctypes.c_void_p), ('n_walkers', ctypes.c_int64),
                ('walkers_per_ensemble', ctypes.c_int64), ('n_steps', ctypes.c_int64),
                ('thin_by', ctypes.c_int64), ('active', ctypes.c_void_p), ('partner', ctypes.c_void_p),
                ('zz', ctypes.c_void_p), ('factor', ctypes.c_void_p), ('logu', ctypes.c_void_p),
                ('chain', ctypes.c_void_p), ('logp_chain', ctypes.c_void_p),
                ('naccept', ctypes.c_void_p), ('status', ctypes.c_void_p)]


# name -> (restype, argtypes); every symbol include/bisip_hip.h declares
SYMBOLS = {
    'bisip_ctx_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, _dp, _dp,
                                        ctypes.POINTER(ModelDesc)]),
    'bisip_batch_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, _dp, _dp,
                                          ctypes.POINTER(ModelDesc)]),
    'bisip_ctx_nspectra': (ctypes.c_int, [ctypes.c_void_p]),
    'bisip_ctx_destroy': (None, [ctypes.c_void_p]),
    'bisip_ctx_set_bounds': (ctypes.c_int, [ctypes.c_void_p, _dp, _dp]),
    'bisip_ctx_set_spectrum_offset': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64]),
    'bisip_ctx_set_variant': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'bisip_ctx_get_variant': (ctypes.c_int, [ctypes.c_void_p]),
    'bisip_logprob': (ctypes.c_int, [ctypes.c_void_p, _dp, ctypes.c_int64, _dp]),
    'bisip_logprob_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.c_void_p, ctypes.c_void_p]),
    'bisip_forward': (ctypes.c_int, [ctypes.c_void_p, _dp, ctypes.c_int64, _dp]),
    'bisip_forward_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.c_void_p, ctypes.c_void_p]),
    'bisip_loglike_z': (ctypes.c_int, [ctypes.c_void_p, _dp, ctypes.c_int64, _dp]),
    'bisip_loglike_z_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                           ctypes.c_void_p, ctypes.c_void_p]),
    'bisip_stretch_half_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StretchArgs), ctypes.c_void_p]),
    'bisip_stretch_eval_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StretchArgs), ctypes.c_void_p]),
    'bisip_stretch_apply_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StretchArgs), ctypes.c_void_p]),
    'bisip_stretch_run_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StretchArgs), ctypes.c_int64,
                                             ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    'bisip_stretch_run_sharded_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(StretchArgs),
                                                     ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    'bisip_stretch_run_sharded_sim_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(StretchArgs),
                                                         ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    'bisip_rccl_unique_id': (ctypes.c_int, [ctypes.c_void_p]),
    'bisip_rccl_comm_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_int]),
    'bisip_rccl_comm_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'bisip_stretch_draw_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_double,
                                              ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64] +
                               [ctypes.c_void_p] * 7),
    'bisip_stretch_philox_inline': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64]),
    'bisip_stretch_run_philox_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StretchArgs), ctypes.c_int64, ctypes.c_int64,
                                                    ctypes.c_int64, ctypes.c_double, ctypes.c_uint64, ctypes.c_int64,
                                                    ctypes.c_void_p, ctypes.c_void_p]),
    'bisip_stretch_persistent_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(PersistArgs), ctypes.c_void_p]),
    'bisip_chain_moments_workspace': (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int]),
    'bisip_chain_moments_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                               ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 4),
    'bisip_chain_percentiles_workspace': (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                                           ctypes.c_int]),
    'bisip_chain_percentiles_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                   ctypes.c_int64, ctypes.c_int, _dp, ctypes.c_int, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    'bisip_column_percentiles_workspace': (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    'bisip_column_percentiles_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, _dp, ctypes.c_int,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    'bisip_grouped_percentiles_workspace': (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    'bisip_grouped_percentiles_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _dp, ctypes.c_int,
                                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    'bisip_forward_percentiles': (ctypes.c_int, [ctypes.c_void_p, _dp, ctypes.c_int64, _dp, ctypes.c_int, _dp]),
    'bisip_numpy_stretch_stream': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.c_int64,
                                                  ctypes.c_double, ctypes.c_int64] + [ctypes.c_void_p] * 4),
    'bisip_forward_spectrum_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                                  ctypes.c_void_p, ctypes.c_void_p]),
    'bisip_forward_spectra_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                                 ctypes.c_void_p, ctypes.c_void_p]),
    'bisip_forward_columns_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                                 ctypes.c_void_p, ctypes.c_void_p]),
    'bisip_columns_percentiles_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, _dp, ctypes.c_int,
                                                     ctypes.c_void_p, ctypes.c_void_p]),
    'bisip_ctx_reduced_check': (ctypes.c_int, [ctypes.c_void_p, _dp, ctypes.c_int64, _dp, _dp]),
    'bisip_read_tables': (ctypes.c_int, [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    'bisip_philox4x32': (None, [ctypes.POINTER(ctypes.c_uint32)] * 3),
    'bisip_ctx_ndim': (ctypes.c_int, [ctypes.c_void_p]),
    'bisip_ctx_nfreq': (ctypes.c_int, [ctypes.c_void_p]),
    'bisip_ctx_device': (ctypes.c_int, [ctypes.c_void_p]),
    'bisip_ctx_loop_flags': (ctypes.c_int, [ctypes.c_void_p]),
    'bisip_ctx_reduced_tiers': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    'bisip_frequency_grid_step': (ctypes.c_int, [ctypes.c_int, _dp, _dp]),
    'bisip_ctx_loglike_const': (ctypes.c_double, [ctypes.c_void_p]),
    'bisip_ctx_kernel_name': (ctypes.c_char_p, [ctypes.c_void_p]),
    'bisip_ctx_reduced_error': (ctypes.c_double, [ctypes.c_void_p]),
    'bisip_polydecomp_operands': (ctypes.c_int, [ctypes.c_int, _dp, _dp, _dp, ctypes.POINTER(ModelDesc),
                                                 _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    'bisip_polydecomp_reduced_estimates': (ctypes.c_int, [ctypes.c_int, _dp, _dp, _dp, ctypes.POINTER(ModelDesc),
                                                          _dp, _dp, _dp]),
    'bisip_ctx_reduced_guard': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64),
                                               _dp, ctypes.POINTER(ctypes.c_int)]),
    'bisip_clock_probe_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]),
    'bisip_fp64_stream_probe_lanes': (ctypes.c_int64, []),
    'bisip_fp64_stream_probe_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p]),
    'bisip_ctx_reduced_guard_rows': (ctypes.c_int, [ctypes.c_void_p, _dp, ctypes.c_int64, _dp, _dp, ctypes.POINTER(ctypes.c_int)]),
    'bisip_chain_shell_rows_workspace': (ctypes.c_int64, [ctypes.c_int64]),
    'bisip_ensemble_gram_workspace': (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int]),
    'bisip_ensemble_gram_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p]),
    'bisip_chain_shell_rows_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_void_p]),
    'bisip_polydecomp_reduced_reference': (ctypes.c_int, [ctypes.c_int, _dp, _dp, _dp, ctypes.POINTER(ModelDesc),
                                                          _dp, ctypes.c_int64, _dp]),
    'bisip_abi_version': (ctypes.c_int, []),
    'bisip_device_count': (ctypes.c_int, []),
    'bisip_last_error': (ctypes.c_char_p, []),
}

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own ``libamdhip64.so.7``; the system ROCm has one
    with the same soname.  Only one HIP runtime may own the GPU in a process, so when
    torch is installed (it supplies device memory, streams and RCCL around this library)
    its runtime is mapped first and ``libbisip_hip.so`` binds to it by soname."""
    if 'torch' in sys.modules:
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def load_library():
    """Load the HIP library; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        _share_hip_runtime_with_torch()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `make -C bisip_amd/csrc` '
                '(or __graft_entry__.build()).  bisip_amd has no CPU fallback.')
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and the header drift apart
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        msg = load_library().bisip_last_error().decode('utf-8', 'replace')
        if rc == -1:
            raise ValueError(f'bisip_hip: {msg}')
        raise RuntimeError(f'bisip_hip (status {rc}): {msg}')


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def polydecomp_operands(w, zn, zn_err, taus, log_taus, c_exp):
    """Host-side precompute of the PolynomialDecomposition operands (no GPU needed)."""
    lib = load_library()
    w, zn, zn_err = _c(w).ravel(), _c(zn).reshape(2, -1), _c(zn_err).reshape(2, -1)
    taus = _c(taus).ravel()
    log_taus = _c(log_taus).reshape(-1, taus.size)
    N, D = w.size, log_taus.shape[0]
    n = D + 1
    desc = ModelDesc(0, D - 1, taus.size, float(c_exp), _p(taus), _p(log_taus))
    out = dict(G_re=np.empty((N, D)), G_im=np.empty((N, D)), R=np.empty((n, n)),
               bhat=np.empty(n), e=np.empty(n), rest=np.empty(1), lconst=np.empty(1))
    _check(lib.bisip_polydecomp_operands(N, _p(w), _p(zn), _p(zn_err), ctypes.byref(desc),
                                         *[_p(out[k]) for k in
                                           ('G_re', 'G_im', 'R', 'bhat', 'e', 'rest', 'lconst')]))
    out['rest'] = float(out['rest'][0])
    out['lconst'] = float(out['lconst'][0])
    return out


def polydecomp_reduced_estimates(w, zn, zn_err, taus, log_taus, c_exp, bounds):
    """Host-side error estimates (plain, compensated) of the QR-reduced kernels for one spectrum
    and prior box -- what BISIP_VARIANT_AUTO decides on.  No GPU needed."""
    lib = load_library()
    w, zn, zn_err = _c(w).ravel(), _c(zn).reshape(2, -1), _c(zn_err).reshape(2, -1)
    taus = _c(taus).ravel()
    log_taus = _c(log_taus).reshape(-1, taus.size)
    b = _c(bounds).reshape(2, -1)
    lo, hi = _c(b[0]), _c(b[1])
    if lo.size != log_taus.shape[0] + 1:
        raise ValueError('bounds must have shape (2, poly_deg + 2)')
    desc = ModelDesc(0, log_taus.shape[0] - 1, taus.size, float(c_exp), _p(taus), _p(log_taus))
    est = np.empty(2)
    _check(lib.bisip_polydecomp_reduced_estimates(w.size, _p(w), _p(zn), _p(zn_err), ctypes.byref(desc), _p(lo), _p(hi), _p(est)))
    return float(est[0]), float(est[1])


def frequency_grid_step(w):
    """Host only: the common step of ln w if ``w`` is a geometric grid in the sense of
    bisip_frequency_grid_step (include/bisip_hip.h), else None."""
    lib = load_library()
    w = _c(w).ravel()
    step = ctypes.c_double(0.0)
    rc = lib.bisip_frequency_grid_step(int(w.size), w.ctypes.data_as(_dp), ctypes.cast(ctypes.byref(step), _dp))
    if rc < 0:
        _check(rc)
    return float(step.value) if rc == 1 else None


def polydecomp_reduced_reference(w, zn, zn_err, taus, log_taus, c_exp, theta):
    """Host-only, no GPU: the PolynomialDecomposition log-likelihood of the rows of ``theta`` with nothing
    rounded to double on the way (bisip_polydecomp_reduced_reference): the yardstick of the reduced
    kernels' estimates, checks and guard."""
    w, zn, zn_err, taus = _c(w), _c(zn), _c(zn_err), _c(taus).ravel()
    log_taus = _c(log_taus).reshape(-1, taus.size)
    theta = _c(theta).reshape(-1, log_taus.shape[0] + 1)
    desc = ModelDesc()
    desc.poly_deg = log_taus.shape[0] - 1
    desc.c_exp = float(c_exp)
    desc.n_taus = taus.size
    desc.taus = _p(taus)
    desc.log_taus = _p(log_taus)
    out = np.empty(theta.shape[0])
    _check(load_library().bisip_polydecomp_reduced_reference(w.size, _p(w), _p(zn), _p(zn_err), ctypes.byref(desc),
                                                            _p(theta), theta.shape[0], _p(out)))
    return out


def device_count():
    return int(load_library().bisip_device_count())


class HipContext:
    """One inversion problem resident on one GPU (``bisip_ctx``)."""

    def __init__(self, model_id, w, zn, zn_err, bounds, device=0, n_modes=0, poly_deg=0,
                 c_exp=1.0, taus=None, log_taus=None, variant='auto'):
        lib = load_library()
        if lib.bisip_device_count() < 1:
            raise RuntimeError('bisip_amd needs a visible AMD GPU (hipGetDeviceCount() == 0); '
                               'there is no CPU fallback')
        w = _c(w)
        batch = w.ndim == 2          # (E, N): batch of spectra
        E = w.shape[0] if batch else 1
        N = w.shape[-1]
        w = w.reshape(E, N)
        zn = _c(zn).reshape(E, 2, -1)
        zn_err = _c(zn_err).reshape(E, 2, -1)
        b = _c(bounds).reshape(2, -1)
        lo, hi = _c(b[0]), _c(b[1])
        if zn.shape[2] != N or zn_err.shape[2] != N:
            raise ValueError('zn and zn_err must have shape (2, N) [or (E, 2, N)] with N = len(w)')
        desc = ModelDesc()
        desc.n_modes = int(n_modes)
        desc.poly_deg = int(poly_deg)
        desc.c_exp = float(c_exp)
        keep = []
        if taus is not None:
            taus = _c(taus).ravel()
            log_taus = _c(log_taus).reshape(int(poly_deg) + 1, taus.size)
            desc.n_taus = taus.size
            desc.taus = _p(taus)
            desc.log_taus = _p(log_taus)
            keep = [taus, log_taus]
        handle = ctypes.c_void_p()
        if batch:
            _check(lib.bisip_batch_create(ctypes.byref(handle), int(device), int(model_id), E, N,
                                          _p(w), _p(zn), _p(zn_err), lo.size, _p(lo), _p(hi),
                                          ctypes.byref(desc)))
        else:
            _check(lib.bisip_ctx_create(ctypes.byref(handle), int(device), int(model_id), N,
                                        _p(w), _p(zn), _p(zn_err), lo.size, _p(lo), _p(hi),
                                        ctypes.byref(desc)))
        del keep
        self._lib = lib
        self._h = handle
        self.ndim = lo.size
        self.N = N
        self.n_spectra = E
        self.device = int(device)
        self.model_id = int(model_id)
        self.n_modes = int(n_modes)
        self.poly_deg = int(poly_deg)
        # bisip_logprob's guard of the QR-reduced kernels (see logprob)
        self._guarded = self.model_id == MODEL_POLYDECOMP
        self._calls, self._escalations_seen, self._guard_warned, self._forced = 0, 0, False, False
        self._guard_enabled = True
        if variant != 'auto':
            self.set_variant(variant)

    # -- lifetime ---------------------------------------------------------------------
    def close(self):
        if getattr(self, '_h', None):
            self._lib.bisip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration ----------------------------------------------------------------
    def set_spectrum_offset(self, first_spectrum):
        """This batch context holds spectra [first_spectrum, first_spectrum + E) of a survey: the
        Philox stream of each is keyed by its survey index."""
        _check(self._lib.bisip_ctx_set_spectrum_offset(self._h, int(first_spectrum)))

    def set_bounds(self, bounds):
        b = _c(bounds).reshape(2, -1)
        if b.shape[1] != self.ndim:
            raise ValueError(f'bounds must have shape (2, {self.ndim})')
        lo, hi = _c(b[0]), _c(b[1])
        _check(self._lib.bisip_ctx_set_bounds(self._h, _p(lo), _p(hi)))

    def set_variant(self, variant):
        v = VARIANTS[variant] if isinstance(variant, str) else int(variant)
        _check(self._lib.bisip_ctx_set_variant(self._h, v))
        self._forced = v != VARIANTS['auto']

    @property
    def variant(self):
        v = self._lib.bisip_ctx_get_variant(self._h)
        return {num: name for name, num in VARIANTS.items()}[v]

    @property
    def kernel_name(self):
        return self._lib.bisip_ctx_kernel_name(self._h).decode()

    @property
    def reduced_tiers(self):
        """PolynomialDecomposition: (spectra on the plain, spectra on the compensated QR-reduced kernel); a
        batch on 'auto' decides per spectrum, inside one launch."""
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(self._lib.bisip_ctx_reduced_tiers(self._h, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    @property
    def loop_flags(self):
        """ColeCole / Shin: 1 = shared reciprocals (the prior box keeps the products normal), 3 = that and
        exponentials stepped along a geometric frequency grid; 0 = safe loop or another model."""
        return int(self._lib.bisip_ctx_loop_flags(self._h))

    @property
    def persistent_walkers(self):
        """Largest single ensemble for which the persistent sampler kernel (one workgroup holds the
        ensemble) beats one launch per half-step with this context's log-probability kernel --
        measured crossovers, benchmarks/micro/persistent_crossover.py: the cheaper the kernel per
        walker, the longer one CU keeps up with launches that spread over the chip."""
        name = self.kernel_name
        if 'reduced_comp' in name:
            # the compensated tier with its triangle in scalar registers and its low words in LDS wins in the
            # persistent kernel at every degree up to 512 walkers (1.6-2.3x the launch path at 32 walkers,
            # degrees 5-10; profiles/r03_micro_persistent_comp_by_degree.txt), up to 1024 at degree <= 5
            return 1024 if self.poly_deg <= 5 else 512
        if 'reduced' in name:
            return 1024
        if 'Dias' in name:
            return 768
        if 'ColeCole' in name:
            return {1: 768, 2: 512}.get(self.n_modes, 256)
        return 512          # Shin, per-frequency PolynomialDecomposition

    @property
    def group_walkers(self):
        """Largest single ensemble, beyond one workgroup, for which the multi-workgroup persistent kernel
        (k_stretch_group: a barrier among the ensemble's workgroups per half-step instead of a launch) beats one
        launch per half-step -- benchmarks/micro/group_sampler.py: 2.5-5.2 us against 4.3-6.3 per half-step at
        2,048 and 4,096 walkers for the reduced PolynomialDecomposition kernels (one lane per slot: one WAVE per
        workgroup, 1.45-1.75x), Cole-Cole, Dias and Shin (1.15-1.4x);
        at 8,192 the two tie or the launches win (fewer lanes per walker), and the per-frequency
        PolynomialDecomposition form, one lane per walker, loses everywhere.  0: never.  A walker's row must fit 64
        bytes (ndim <= 7)."""
        if self.ndim > 7 or self.n_spectra != 1:
            return 0
        name = self.kernel_name
        if 'PDCollapsed' in name or 'faithful' in name or 'wave' in name:
            return 0
        # the reduced PolynomialDecomposition kernels (one lane per slot, one wave per workgroup) up to 32,768 walkers:
        # up to 256 workgroups on all XCDs, meeting at a two-level barrier
        return 32768 if 'k_logprob_pd_reduced' in name else 4096

    @property
    def persistent_in_big_batches(self):
        """A batch that fills the chip (>= 65,536 walkers in all) runs its launches in the bulk regime; the
        persistent kernel -- one workgroup per ensemble, waiting on dependency chains -- still wins with every
        kernel except the compensated triangle from degree 9, where the two tie (512 x 256, whole fits: 5.9
        against 10.7 us per half-step at degree 5, 11.3 against 15.2 at degree 6, 10.4 against 13.5 at degree
        7, 15.0 against 16.4 at degree 8, 24.2 against 23.8 at degree 9; benchmarks/micro/batch_comp_by_degree.py)."""
        return not ('reduced_comp' in self.kernel_name and self.poly_deg > 8)

    @property
    def reduced_error(self):
        """Estimated worst relative log-prob error of the QR-reduced kernel for the current box."""
        return float(self._lib.bisip_ctx_reduced_error(self._h))

    def reduced_check(self, theta, logp):
        """Worst |logp - reference| / max(1, |reference|) of log-probabilities a QR-reduced kernel
        gave for the rows of ``theta``, against the reduced form in long double from the unrounded
        operands (host; bisip_ctx_reduced_check).  PolynomialDecomposition contexts only."""
        theta = self._theta2d(theta)
        logp = _c(logp).ravel()
        if logp.size != theta.shape[0]:
            raise ValueError('one log-probability per row of theta')
        out = np.empty(1)
        _check(self._lib.bisip_ctx_reduced_check(self._h, _p(theta), theta.shape[0], _p(logp), _p(out)))
        return float(out[0])

    def reduced_guard_rows(self, theta, logp):
        """The guard for rows that came from the DEVICE (bisip_ctx_reduced_guard_rows): measure the
        log-probabilities the context's QR-reduced kernel gave ``theta`` as ``reduced_check`` does and, past
        2e-11 on a context that chose its formulation itself ('auto'), move the context to the next
        formulation.  Returns ``(worst relative error, escalated)``; after an escalation every
        log-probability the caller holds is stale (the device sampler re-runs its chunk).  Rows with a NaN
        (unfilled slots of chain_shell_rows_dev) and rows outside the prior are skipped."""
        theta = self._theta2d(theta)
        logp = _c(logp).ravel()
        if logp.size != theta.shape[0]:
            raise ValueError('one log-probability per row of theta')
        worst, esc = np.empty(1), ctypes.c_int(0)
        _check(self._lib.bisip_ctx_reduced_guard_rows(self._h, _p(theta), theta.shape[0], _p(logp), _p(worst),
                                                      ctypes.byref(esc)))
        return float(worst[0]), bool(esc.value)

    @property
    def guards_itself(self):
        """True for a PolynomialDecomposition context that CHOSE a QR-reduced kernel from its estimate
        ('auto'): the formulation the device sampler measures on its own rows and may move."""
        return (self.model_id == MODEL_POLYDECOMP and not self._forced and self.variant in ('reduced', 'reduced_comp')
                and self._guard_enabled)

    @property
    def loglike_const(self):
        return float(self._lib.bisip_ctx_loglike_const(self._h))

    # -- compute ----------------------------------------------------------------------
    def _theta2d(self, theta):
        theta = np.asarray(theta)
        if theta.ndim != 2 or theta.shape[1] != self.ndim:
            raise ValueError(f'theta must have shape (W, {self.ndim}), got {theta.shape}')
        return _c(theta)

    def logprob(self, theta):
        """theta (W, ndim) host -> logp (W,) host.

        PolynomialDecomposition: the library measures the QR-reduced kernel on rows of this very
        batch on the first call and every 2^n-th after it (bisip_ctx_reduced_guard); a context on
        'auto' that fails moves to the next formulation and re-evaluates the batch -- reported here
        as a RuntimeWarning -- and a caller-forced reduced variant past the tolerance is named."""
        theta = self._theta2d(theta)
        out = np.empty(theta.shape[0], dtype=np.float64)
        _check(self._lib.bisip_logprob(self._h, _p(theta), theta.shape[0], _p(out)))
        if self._guarded and theta.shape[0]:
            self._calls += 1
            if self._calls & (self._calls - 1) == 0:
                self._report_guard()
        return out

    def reduced_guard(self, enable=None):
        """(checks made, worst relative error seen, formulation changes) of the guard bisip_logprob runs
        on the QR-reduced kernels; ``enable`` True / False switches it on / off."""
        n, w, e = ctypes.c_int64(0), ctypes.c_double(0.0), ctypes.c_int(0)
        if enable is not None:
            self._guard_enabled = bool(enable)
        _check(self._lib.bisip_ctx_reduced_guard(self._h, -1 if enable is None else int(bool(enable)),
                                                 ctypes.byref(n), ctypes.cast(ctypes.byref(w), _dp), ctypes.byref(e)))
        return int(n.value), float(w.value), int(e.value)

    def _report_guard(self):
        import warnings
        _, worst, esc = self.reduced_guard()
        if esc > self._escalations_seen:
            self._escalations_seen = esc
            warnings.warn(f'the QR-reduced kernel was {worst:.1e} (relative) away from the exact log-probability on '
                          f'rows of this batch; the context now runs {self.kernel_name} and the batch was '
                          're-evaluated with it', RuntimeWarning, stacklevel=3)
        elif worst > 1e-10 and self._forced and not self._guard_warned and self.variant in ('reduced', 'reduced_comp'):
            self._guard_warned = True
            warnings.warn(f'the {self.variant!r} kernel is {worst:.1e} (relative) away from the exact log-probability '
                          "on rows of this batch (tolerance 1e-10); use variant='auto' or 'reduced_comp'",
                          RuntimeWarning, stacklevel=3)

    def forward(self, theta):
        """theta (W, ndim) host -> Z (W, 2, N) host."""
        theta = self._theta2d(theta)
        out = np.empty((theta.shape[0], 2, self.N), dtype=np.float64)
        _check(self._lib.bisip_forward(self._h, _p(theta), theta.shape[0], _p(out)))
        return out

    def forward_percentiles(self, theta, p):
        """``np.percentile(forward(theta), p, axis=0)`` computed on the device: theta (W, ndim)
        host -> (len(p), 2, N) host (single-spectrum contexts; a batch context raises NotImplementedError)."""
        theta = self._theta2d(theta)
        p = _c(np.atleast_1d(p)).ravel()
        out = np.empty((p.size, 2, self.N), dtype=np.float64)
        rc = self._lib.bisip_forward_percentiles(self._h, _p(theta), theta.shape[0], _p(p), p.size, _p(out))
        if rc == -4:
            raise NotImplementedError(self._lib.bisip_last_error().decode('utf-8', 'replace'))
        _check(rc)
        return out

    def loglike_z(self, Z):
        """Gaussian log-likelihood of caller-computed responses: Z (W, 2, N) host -> (W,) host."""
        Z = _c(Z)
        if Z.ndim != 3 or Z.shape[1:] != (2, self.N):
            raise ValueError(f'Z must have shape (W, 2, {self.N}), got {Z.shape}')
        out = np.empty(Z.shape[0], dtype=np.float64)
        _check(self._lib.bisip_loglike_z(self._h, _p(Z), Z.shape[0], _p(out)))
        return out

    def logprob_dev(self, d_theta_ptr, W, d_out_ptr, stream=0):
        """Device pointers (ints), asynchronous on ``stream`` (a hipStream_t as int)."""
        _check(self._lib.bisip_logprob_dev(self._h, ctypes.c_void_p(d_theta_ptr), int(W),
                                           ctypes.c_void_p(d_out_ptr), ctypes.c_void_p(stream)))

    def forward_dev(self, d_theta_ptr, W, d_Z_ptr, stream=0):
        _check(self._lib.bisip_forward_dev(self._h, ctypes.c_void_p(d_theta_ptr), int(W),
                                           ctypes.c_void_p(d_Z_ptr), ctypes.c_void_p(stream)))

    def forward_spectrum_dev(self, spectrum, d_theta_ptr, W, d_Z_ptr, stream=0):
        """Batch context: forward of W rows that all belong to one spectrum (device pointers)."""
        _check(self._lib.bisip_forward_spectrum_dev(self._h, int(spectrum), ctypes.c_void_p(d_theta_ptr), int(W),
                                                    ctypes.c_void_p(d_Z_ptr), ctypes.c_void_p(stream)))

    def forward_columns_dev(self, first_spectrum, n_spectra, d_theta_ptr, W, d_cols_ptr, stream=0):
        """Batch (or single) context: the responses of W rows over n_spectra consecutive spectra, written
        column-major (n_spectra, 2N, W / n_spectra) -- what columns_percentiles_dev reads (device pointers)."""
        _check(self._lib.bisip_forward_columns_dev(self._h, int(first_spectrum), int(n_spectra), ctypes.c_void_p(d_theta_ptr),
                                                   int(W), ctypes.c_void_p(d_cols_ptr), ctypes.c_void_p(stream)))

    def forward_spectra_dev(self, first_spectrum, n_spectra, d_theta_ptr, W, d_Z_ptr, stream=0):
        """Batch context: forward of W rows over n_spectra consecutive spectra, W / n_spectra rows each
        (a multiple of 64 when n_spectra > 1), one launch (device pointers)."""
        _check(self._lib.bisip_forward_spectra_dev(self._h, int(first_spectrum), int(n_spectra), ctypes.c_void_p(d_theta_ptr),
                                                   int(W), ctypes.c_void_p(d_Z_ptr), ctypes.c_void_p(stream)))

    # -- device-resident stretch move ---------------------------------------------------
    def stretch_half_dev(self, args, stream=0):
        _check(self._lib.bisip_stretch_half_dev(self._h, ctypes.byref(args), ctypes.c_void_p(stream)))

    def stretch_eval_dev(self, args, stream=0):
        _check(self._lib.bisip_stretch_eval_dev(self._h, ctypes.byref(args), ctypes.c_void_p(stream)))

    def stretch_apply_dev(self, args, stream=0):
        _check(self._lib.bisip_stretch_apply_dev(self._h, ctypes.byref(args), ctypes.c_void_p(stream)))

    def stretch_run_dev(self, first_args, W, n_steps, thin_by=1, stream=0):
        _check(self._lib.bisip_stretch_run_dev(self._h, ctypes.byref(first_args), int(W),
                                               int(n_steps), int(thin_by), ctypes.c_void_p(stream)))

    def stretch_run_sharded_dev(self, comm, first_args, W, n_steps, thin_by=1, stream=0):
        """``comm``: an ncclComm_t as an integer (rccl_comm_create() or torch's _comm_ptr())."""
        _check(self._lib.bisip_stretch_run_sharded_dev(self._h, ctypes.c_void_p(comm), ctypes.byref(first_args),
                                                       int(W), int(n_steps), int(thin_by), ctypes.c_void_p(stream)))

    def stretch_run_sharded_sim_dev(self, world, first_args, W, n_steps, thin_by=1, stream=0):
        """The sharded half-step loop with every rank of a ``world``-rank group evaluated on this device in
        turn and no collective (test aid: bisip_stretch_run_sharded_sim_dev)."""
        _check(self._lib.bisip_stretch_run_sharded_sim_dev(self._h, int(world), ctypes.byref(first_args),
                                                           int(W), int(n_steps), int(thin_by), ctypes.c_void_p(stream)))

    def stretch_persistent_dev(self, args, stream=0):
        """Returns False when the ensemble does not fit one workgroup (status -4)."""
        rc = self._lib.bisip_stretch_persistent_dev(self._h, ctypes.byref(args), ctypes.c_void_p(stream))
        if rc == -4:
            return False
        _check(rc)
        return True

    def stretch_philox_inline(self, W):
        """True where :meth:`stretch_run_philox_dev` serves this context's ensemble of ``W`` walkers (a single
        ensemble big enough for the packed-state half-step)."""
        return bool(self._lib.bisip_stretch_philox_inline(self._h, int(W)))

    def stretch_run_philox_dev(self, first_args, W, n_steps, thin_by, a, seed, step0, perm, stream=0):
        """:meth:`stretch_run_dev` with the Philox stream drawn in place by every half-step launch: no stream arrays
        (``first_args``' five stream pointers are ignored); ``perm``: device pointer of this chunk's (n_steps, 3)
        affine splits.  Same chain as :meth:`stretch_draw_dev` + :meth:`stretch_run_dev`."""
        _check(self._lib.bisip_stretch_run_philox_dev(self._h, ctypes.byref(first_args), int(W), int(n_steps), int(thin_by),
                                                      float(a), int(seed), int(step0), ctypes.c_void_p(perm),
                                                      ctypes.c_void_p(stream)))

    def stretch_draw_dev(self, W, a, seed, step0, n_steps, perm, active, partner, zz, factor, logu,
                         stream=0):
        _check(self._lib.bisip_stretch_draw_dev(self._h, int(W), float(a), int(seed), int(step0),
                                                int(n_steps), *[ctypes.c_void_p(p) for p in
                                                                (perm, active, partner, zz, factor, logu)],
                                                ctypes.c_void_p(stream)))


RCCL_ID_BYTES = 128


def clock_probe_dev(d_out_ptr, window_us, stream=0):
    """One wavefront that brackets ``window_us`` with (shader clock, 100 MHz clock) reads into the
    four int64 at ``d_out_ptr``: enqueue it on a side stream next to a kernel under measurement;
    engine clock in GHz = (out[1] - out[0]) / (out[3] - out[2]) * 0.1."""
    _check(load_library().bisip_clock_probe_dev(ctypes.c_void_p(d_out_ptr), float(window_us), ctypes.c_void_p(stream)))


def fp64_stream_probe_lanes():
    """Doubles the output buffer of :func:`fp64_stream_probe_dev` must hold."""
    return int(load_library().bisip_fp64_stream_probe_lanes())


def fp64_stream_probe_dev(d_out_ptr, rounds, stream=0):
    """One launch of a stream of independent fp64 FMAs (8 waves per SIMD on every compute unit, 32 * ``rounds``
    per wave, operands with full mantissas): the ceiling a compute-bound kernel is held to.  Returns the launch's
    number of wave-instructions; the caller times it."""
    n = ctypes.c_int64(0)
    _check(load_library().bisip_fp64_stream_probe_dev(ctypes.c_void_p(d_out_ptr), int(rounds), ctypes.byref(n), ctypes.c_void_p(stream)))
    return int(n.value)


def rccl_unique_id():
    """ncclGetUniqueId as bytes (rank 0 makes it, every rank passes it to rccl_comm_create)."""
    buf = ctypes.create_string_buffer(RCCL_ID_BYTES)
    _check(load_library().bisip_rccl_unique_id(buf))
    return buf.raw


def rccl_comm_create(world, rank, unique_id, device):
    """ncclCommInitRank on ``device`` (collective over the ranks); returns the ncclComm_t as int."""
    if len(unique_id) != RCCL_ID_BYTES:
        raise ValueError(f'unique_id must be {RCCL_ID_BYTES} bytes')
    comm = ctypes.c_void_p()
    buf = ctypes.create_string_buffer(bytes(unique_id), RCCL_ID_BYTES)
    _check(load_library().bisip_rccl_comm_create(ctypes.byref(comm), int(world), int(rank), buf, int(device)))
    return comm.value


def rccl_comm_destroy(comm):
    _check(load_library().bisip_rccl_comm_destroy(ctypes.c_void_p(comm)))


def chain_moments_workspace(n_samples, n_ensembles, ndim):
    return int(load_library().bisip_chain_moments_workspace(n_samples, n_ensembles, ndim))


def chain_moments_dev(d_chain_ptr, n_samples, sample_stride, n_ensembles, walkers_per_ensemble, ndim,
                      d_mean_ptr, d_std_ptr, d_work_ptr, stream=0):
    """Device pointers (ints); asynchronous on ``stream``."""
    _check(load_library().bisip_chain_moments_dev(d_chain_ptr, n_samples, sample_stride, n_ensembles,
                                                  walkers_per_ensemble, ndim, d_mean_ptr, d_std_ptr,
                                                  d_work_ptr, stream))


def ensemble_gram_workspace(W, ndim):
    """Doubles of device scratch ensemble_gram_dev needs (0: ndim beyond 8)."""
    return int(load_library().bisip_ensemble_gram_workspace(int(W), int(ndim)))


def ensemble_gram_dev(d_coords_ptr, W, ndim, d_out_ptr, d_work_ptr, stream=0):
    """Shifted sums and second moments of a device-resident (W, ndim) ensemble into d_out
    (ndim + ndim (ndim + 1) / 2 doubles); device pointers (ints), asynchronous on ``stream``."""
    _check(load_library().bisip_ensemble_gram_dev(d_coords_ptr, int(W), int(ndim), d_out_ptr, d_work_ptr, stream))


def chain_shell_rows_workspace(n_ensembles):
    """Bytes of device scratch chain_shell_rows_dev needs."""
    return int(load_library().bisip_chain_shell_rows_workspace(int(n_ensembles)))


def chain_shell_rows_dev(d_chain_ptr, d_logp_ptr, n_samples, n_ensembles, walkers_per_ensemble, ndim, k, n_stride,
                         d_out_ptr, d_work_ptr, stream=0, ties=True):
    """Per ensemble the k stored samples of smallest |logp| (+ n_stride walkers of the first sample) into
    d_out (n_ensembles, k + n_stride, ndim + 1); device pointers (ints), asynchronous on ``stream``.
    ``ties=False`` leaves out the samples that tie with the k-th (a reproducible set)."""
    _check(load_library().bisip_chain_shell_rows_dev(d_chain_ptr, d_logp_ptr, int(n_samples), int(n_ensembles),
                                                     int(walkers_per_ensemble), int(ndim), int(k), int(n_stride),
                                                     int(bool(ties)), d_out_ptr, d_work_ptr, stream))


def chain_percentiles_workspace(n_samples, n_ensembles, walkers_per_ensemble, ndim, n_percentiles):
    """Bytes of device scratch bisip_chain_percentiles_dev needs (0: shape not supported)."""
    return int(load_library().bisip_chain_percentiles_workspace(n_samples, n_ensembles, walkers_per_ensemble,
                                                                ndim, n_percentiles))


def chain_percentiles_dev(d_chain_ptr, n_samples, sample_stride, n_ensembles, walkers_per_ensemble, ndim,
                          percentiles, d_out_ptr, d_work_ptr, work_bytes, stream=0):
    """Device pointers (ints); ``percentiles`` is a host array in [0, 100]."""
    p = _c(percentiles).ravel()
    _check(load_library().bisip_chain_percentiles_dev(d_chain_ptr, n_samples, sample_stride, n_ensembles,
                                                      walkers_per_ensemble, ndim, _p(p), p.size, d_out_ptr,
                                                      d_work_ptr, work_bytes, stream))


def column_percentiles_workspace(n_rows, n_cols, n_percentiles):
    """BYTES of device workspace for column_percentiles_dev (0: more than 2^31 values)."""
    return int(load_library().bisip_column_percentiles_workspace(int(n_rows), int(n_cols), int(n_percentiles)))


def column_percentiles_dev(d_rows_ptr, n_rows, n_cols, percentiles, d_out_ptr, d_work_ptr, work_bytes, stream=0):
    """np.percentile(rows, p, axis=0) of a device-resident (n_rows, n_cols) array; device pointers (ints)."""
    p = _c(percentiles).ravel()
    _check(load_library().bisip_column_percentiles_dev(d_rows_ptr, int(n_rows), int(n_cols), _p(p), p.size,
                                                       d_out_ptr, d_work_ptr, int(work_bytes), stream))


def columns_percentiles_dev(d_cols_ptr, n_columns, n, percentiles, d_out_ptr, stream=0):
    """np.percentile of n_columns contiguous device columns of n values: d_out (len(p), n_columns).  Device
    pointers (ints); no workspace."""
    p = _c(percentiles).ravel()
    _check(load_library().bisip_columns_percentiles_dev(d_cols_ptr, int(n_columns), int(n), _p(p), p.size, d_out_ptr, stream))


def grouped_percentiles_workspace(n_groups, n_rows, n_cols, n_percentiles):
    """BYTES of device workspace for grouped_percentiles_dev (0: more than 2^31 values in one sort)."""
    return int(load_library().bisip_grouped_percentiles_workspace(int(n_groups), int(n_rows), int(n_cols), int(n_percentiles)))


def grouped_percentiles_dev(d_rows_ptr, n_groups, n_rows, n_cols, percentiles, d_out_ptr, d_work_ptr, work_bytes, stream=0):
    """np.percentile(rows[g], p, axis=0) for every g of a device-resident (n_groups, n_rows, n_cols) array in
    one segmented sort; d_out (len(p), n_groups, n_cols).  Device pointers (ints)."""
    p = _c(percentiles).ravel()
    _check(load_library().bisip_grouped_percentiles_dev(d_rows_ptr, int(n_groups), int(n_rows), int(n_cols), _p(p), p.size,
                                                        d_out_ptr, d_work_ptr, int(work_bytes), stream))


def numpy_stretch_stream(rng, W, a, n_steps, out=None):
    """n_steps iterations of the stretch move's RandomState stream, generated in C; ``rng``
    (a numpy.random.RandomState) is advanced exactly as draw_step would advance it.
    Returns active, partner (int32) and zz, u (float64), each (n_steps, 2, (W+1)//2);
    ``out`` = four preallocated C-contiguous arrays of those shapes to fill instead."""
    lib = load_library()
    name, key, pos, has_gauss, cached = rng.get_state()
    if name != 'MT19937':
        raise ValueError('expected an MT19937 RandomState')
    key = np.ascontiguousarray(key, dtype=np.uint32).copy()
    cpos = ctypes.c_int32(int(pos))
    nh = (int(W) + 1) // 2
    shape = (int(n_steps), 2, nh)
    if out is None:
        out = (np.empty(shape, np.int32), np.empty(shape, np.int32), np.empty(shape), np.empty(shape))
    active, partner, zz, u = out
    for arr, dt in ((active, np.int32), (partner, np.int32), (zz, np.float64), (u, np.float64)):
        if arr.shape != shape or arr.dtype != dt or not arr.flags.c_contiguous:
            raise ValueError(f'out arrays must be C-contiguous {shape}: int32, int32, float64, float64')
    _check(lib.bisip_numpy_stretch_stream(key.ctypes.data, ctypes.byref(cpos), int(W), float(a),
                                          int(n_steps), active.ctypes.data, partner.ctypes.data,
                                          zz.ctypes.data, u.ctypes.data))
    rng.set_state((name, key, int(cpos.value), has_gauss, cached))
    return active, partner, zz, u


def read_tables(paths, headers, n_rows, threads=1):
    """Parse many 5-column spectrum files on `threads` host threads (bisip_read_tables).
    Returns ``tables (n_files, n_rows, 5)`` and ``status (n_files,)``: 0 = filled with what
    np.loadtxt yields, 1 = read this file with np.loadtxt instead (see include/bisip_hip.h)."""
    import os
    lib = load_library()
    enc = [os.fsencode(p) for p in paths]
    arr = (ctypes.c_char_p * len(enc))(*enc)
    tables = np.empty((len(enc), int(n_rows), 5), dtype=np.float64)
    status = np.ones(len(enc), dtype=np.int32)
    _check(lib.bisip_read_tables(arr, len(enc), int(headers), int(n_rows), tables.ctypes.data,
                                 status.ctypes.data, int(threads)))
    return tables, status


def philox4x32(counter, key):
    """One Philox4x32-10 block on the host (same inline code as the device stream)."""
    lib = load_library()
    c = (ctypes.c_uint32 * 4)(*[int(x) & 0xffffffff for x in counter])
    k = (ctypes.c_uint32 * 2)(*[int(x) & 0xffffffff for x in key])
    out = (ctypes.c_uint32 * 4)()
    lib.bisip_philox4x32(c, k, out)
    return [int(x) for x in out]
