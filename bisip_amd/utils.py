"""Data ingest and chain statistics (host side of the drop-in surface).

``load_data`` defines the constant operands of the hot path -- ``w (N,)``,
``zn (2,N)``, ``zn_err (2,N)`` -- and follows the arithmetic of the reference's
``utils.load_data`` (src/bisip/utils.py:108-146) step for step so that the
operands are bit-identical (pinned by tests/golden/load_data.npz).  The chain
statistics mirror src/bisip/utils.py:17-106; the model-space percentiles run the
batched forward kernel instead of a Python loop over the chain.
"""

import warnings

import numpy as np

_COLUMNS = ('freq', 'amp', 'pha', 'amp_err', 'pha_err')


def columns_to_data(table, ph_units='mrad'):
    """(N,5) table [freq, amp, pha, amp_err, pha_err] -> the reference's data dict.

    reference: src/bisip/utils.py:121-144
    """
    table = np.asarray(table, dtype=np.float64)
    if table.ndim != 2 or table.shape[1] < 5:
        raise ValueError('expected 5 comma-separated columns: freq, amp, pha, amp_err, pha_err')
    data = {name: table[:, i] for i, name in enumerate(_COLUMNS)}
    if ph_units == 'mrad':
        data['pha'] = data['pha'] / 1000
        data['pha_err'] = data['pha_err'] / 1000
    if ph_units == 'deg':
        data['pha'] = np.radians(data['pha'])
        data['pha_err'] = np.radians(data['pha_err'])
    amp, pha = data['amp'], data['pha']
    cos_p, sin_p = np.cos(pha), np.sin(pha)
    data['Z'] = amp * (cos_p + 1j * sin_p)
    # first-order propagation of (amp_err, pha_err) to the real/imaginary parts
    err_im = np.sqrt((amp * cos_p * data['pha_err']) ** 2 + (sin_p * data['amp_err']) ** 2)
    err_re = np.sqrt((amp * sin_p * data['pha_err']) ** 2 + (cos_p * data['amp_err']) ** 2)
    data['Z_err'] = err_re + 1j * err_im
    data['norm_factor'] = max(abs(data['Z']))
    zn = data['Z'] / data['norm_factor']
    zn_err = data['Z_err'] / data['norm_factor']
    data['zn'] = np.array([zn.real, zn.imag])
    data['zn_err'] = np.array([zn_err.real, zn_err.imag])
    data['N'] = len(data['freq'])
    data['w'] = 2 * np.pi * data['freq']
    return data


def load_data(filename, headers=1, ph_units='mrad'):
    """Read one 5-column spectrum file (reference: src/bisip/utils.py:108-146,
    format: docs/user/data_format.rst:7-28)."""
    table = np.loadtxt(f'{filename}', skiprows=headers, delimiter=',')
    return columns_to_data(np.atleast_2d(table), ph_units)


def _is_path(sp):
    return isinstance(sp, (str, bytes)) or hasattr(sp, '__fspath__')


def tables_to_operands(tables, ph_units='mrad'):
    """(E,N,5) stacked tables -> the stacked operands of a batch context, each spectrum exactly what
    ``columns_to_data`` (the reference's per-file arithmetic, src/bisip/utils.py:124-144) gives.
    Everything that rounds the same for any array shape -- products, quotients, square roots --
    is done once over the whole batch; cos / sin / |Z| stay one call per spectrum on arrays laid out
    as in the per-file path (NumPy picks SIMD or scalar code for those by layout and length, and
    the two need not agree in the last bit)."""
    T = np.ascontiguousarray(tables, dtype=np.float64)
    if T.ndim != 3 or T.shape[2] != 5:
        raise ValueError('expected stacked (E, N, 5) tables')
    E, N = T.shape[:2]
    freq, amp, pha, amp_err, pha_err = (T[:, :, i] for i in range(5))
    if ph_units == 'mrad':
        pha, pha_err = pha / 1000, pha_err / 1000
    if ph_units == 'deg':
        pha, pha_err = np.radians(pha), np.radians(pha_err)
    cos_p, sin_p = np.empty((E, N)), np.empty((E, N))
    for e in range(E):
        np.cos(pha[e], out=cos_p[e])
        np.sin(pha[e], out=sin_p[e])
    Z = amp * (cos_p + 1j * sin_p)
    err_im = np.sqrt((amp * cos_p * pha_err) ** 2 + (sin_p * amp_err) ** 2)
    err_re = np.sqrt((amp * sin_p * pha_err) ** 2 + (cos_p * amp_err) ** 2)
    Z_err = err_re + 1j * err_im
    mod = np.empty((E, N))
    for e in range(E):
        np.abs(Z[e], out=mod[e])
    norm = mod.max(axis=1)
    for e in np.flatnonzero(np.isnan(norm)):     # the reference's Python max() skips NaNs its own way
        norm[e] = max(mod[e])
    zn = Z / norm[:, None]
    zn_err = Z_err / norm[:, None]
    return {'w': 2 * np.pi * freq,
            'zn': np.ascontiguousarray(np.stack([zn.real, zn.imag], axis=1)),
            'zn_err': np.ascontiguousarray(np.stack([zn_err.real, zn_err.imag], axis=1)),
            'norm_factor': norm, 'N': N}


def load_data_batch(spectra, headers=1, ph_units='mrad', threads=None):
    """Ingest many spectra that share one frequency count (BASELINE config 5):
    every item is a file path (read as ``load_data`` reads it) or a raw (N,5) table
    [freq, amp, pha, amp_err, pha_err].  Returns the stacked operands of a batch context --
    ``w (E,N)``, ``zn (E,2,N)``, ``zn_err (E,2,N)``, ``norm_factor (E,)``, ``N`` -- each row
    exactly what the reference's per-file ``load_data`` yields (src/bisip/utils.py:108-146).

    Files are parsed by the library on ``threads`` host threads (default: the CPUs this process may
    use, at most 16) -- ``bisip_read_tables``, ~40x np.loadtxt on a survey of thousands of small files
    (benchmarks/ingest.py); a file it does not recognise as plain 5-column text is read with
    np.loadtxt itself, so it behaves, or fails, as in the reference."""
    import os
    spectra = list(spectra)
    if not spectra:
        raise ValueError('no spectra')
    files = [i for i, sp in enumerate(spectra) if _is_path(sp)]

    def checked(table):
        table = np.asarray(table, dtype=np.float64)
        if table.ndim != 2 or table.shape[1] < 5:
            raise ValueError('expected 5 comma-separated columns: freq, amp, pha, amp_err, pha_err')
        return table

    def loadtxt(i):
        return checked(np.atleast_2d(np.loadtxt(f'{spectra[i]}', skiprows=headers, delimiter=',')))

    first = loadtxt(files[0]) if files else checked(spectra[0])      # fixes the row count of the batch
    N = first.shape[0]
    T = np.empty((len(spectra), N, 5))
    counts = {N}

    def place(i, table):
        counts.add(table.shape[0])
        if table.shape[0] == N:
            T[i] = table[:, :5]

    if files:
        place(files[0], first)
        rest = [i for i in files[1:] if not isinstance(spectra[i], bytes)]
        status = np.zeros(0, dtype=np.int32)
        if rest:
            from . import _hip
            if threads is None:
                threads = min(cpu_quota(), 16)
            got, status = _hip.read_tables([os.fspath(spectra[i]) for i in rest], headers, N, threads)
            T[rest] = got
        todo = set(files[1:]) - {i for i, st in zip(rest, status) if st == 0}
        for i in sorted(todo):
            place(i, loadtxt(i))
    for i, sp in enumerate(spectra):
        if not _is_path(sp):
            place(i, checked(sp))
    if len(counts) != 1:
        raise ValueError(f'spectra have different frequency counts: {sorted(counts)}')
    return tables_to_operands(T, ph_units)


class utils(object):
    """Mixin with the reference's utility methods (src/bisip/utils.py:15)."""

    def load_data(self, filename, headers=1, ph_units='mrad'):
        return load_data(filename, headers, ph_units)

    def parse_chain(self, chain, **kwargs):
        """reference: src/bisip/utils.py:87-106"""
        if chain is None:
            kwargs['flat'] = True
            chain = self.get_chain(**kwargs)
            if 'discard' not in kwargs and 'thin' not in kwargs:
                warnings.warn('No samples were discarded from the chain.\n'
                              'Pass discard and thin keywords to remove '
                              'burn-in samples and reduce autocorrelation.', UserWarning)
            return chain
        if chain.ndim > 2:
            raise ValueError('Flatten chain by passing flat=True.')
        if 'discard' in kwargs or 'thin' in kwargs:
            raise ValueError('Please pass either a chain obtained with the get_chain() '
                             'method or pass discard and thin keywords to parse the full '
                             'chain. Do not pass both.')
        return chain

    def get_model_percentile(self, p=[2.5, 50, 97.5], chain=None, **kwargs):
        """Percentiles of the model response over a chain (src/bisip/utils.py:17-35);
        the forward pass over the whole chain is one batched kernel launch."""
        s = self._device_chain_sampler(chain, kwargs)
        if s is not None:       # fit(chain='device'): forward and percentiles where the chain lies
            try:
                out = s.model_percentiles(p, discard=kwargs.get('discard', 0), thin=kwargs.get('thin', 1))
                return out if np.ndim(p) else out[0]
            except NotImplementedError:
                pass
        chain = np.ascontiguousarray(self.parse_chain(chain, **kwargs), dtype=np.float64)
        # forward over the chain and the percentiles over axis 0 both on the device: only the chain goes
        # up and (len(p), 2, N) comes back (bisip_forward_percentiles) -- np.percentile's own doubles
        out = self._context().forward_percentiles(chain, p)
        return out if np.ndim(p) else out[0]

    def _device_chain_sampler(self, chain, kwargs):
        """The sampler, when its chain lives in HBM (fit(chain='device')) and the caller asked for a
        summary of the fitted chain rather than of an array of its own; else None."""
        s = getattr(self, '_sampler', None)
        if chain is not None or not getattr(s, 'chain_on_device', False):
            return None
        self._check_if_fitted()
        extra = set(kwargs) - {'discard', 'thin', 'flat'}
        if extra:
            raise TypeError(f'unexpected keyword(s) {sorted(extra)}')
        if 'discard' not in kwargs and 'thin' not in kwargs:      # same advice as parse_chain
            warnings.warn('No samples were discarded from the chain.\n'
                          'Pass discard and thin keywords to remove '
                          'burn-in samples and reduce autocorrelation.', UserWarning)
        return s

    def get_param_percentile(self, p=[2.5, 50, 97.5], chain=None, **kwargs):
        """reference: src/bisip/utils.py:37-53"""
        s = self._device_chain_sampler(chain, kwargs)
        if s is not None:      # sorted and interpolated where the chain lies
            out = s.param_percentiles(p, discard=kwargs.get('discard', 0), thin=kwargs.get('thin', 1))[:, 0, :]
            return out if np.ndim(p) else out[0]
        return np.percentile(self.parse_chain(chain, **kwargs), p, axis=0)

    def get_param_mean(self, chain=None, **kwargs):
        """reference: src/bisip/utils.py:55-69"""
        s = self._device_chain_sampler(chain, kwargs)
        if s is not None:
            return s.param_moments(discard=kwargs.get('discard', 0), thin=kwargs.get('thin', 1))[0][0]
        return np.mean(self.parse_chain(chain, **kwargs), axis=0)

    def get_param_std(self, chain=None, **kwargs):
        """reference: src/bisip/utils.py:71-85"""
        s = self._device_chain_sampler(chain, kwargs)
        if s is not None:
            return s.param_moments(discard=kwargs.get('discard', 0), thin=kwargs.get('thin', 1))[1][0]
        return np.std(self.parse_chain(chain, **kwargs), axis=0)


_QUOTA_APPLIED = False


def cpu_quota():
    """CPUs this process may use: the cgroup CFS quota (cpu.max / cfs_quota_us) when there is
    one, else the affinity mask."""
    import math
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max',):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != 'max':
                n = min(n, max(1, math.floor(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0 and p > 0:
            n = min(n, max(1, q // p))
    except (OSError, ValueError):
        pass
    return n


def respect_cpu_quota():
    """Cap the host thread pools (torch intra-op, BLAS/OpenMP behind NumPy) at the container's
    CPU quota, once per process.  On a 256-core host with a 16-CPU quota the default pools
    (128 torch + 64 OpenBLAS threads) spend the quota of a 100 ms scheduler period in a few
    milliseconds of a parallel memcpy or SVD and the whole process is then throttled for the
    rest of the period: 20-95 ms stalls at random host-side places (measured: cpu.stat
    nr_throttled).  Set BISIP_KEEP_THREADS=1 to leave the pools alone."""
    global _QUOTA_APPLIED
    import os
    if _QUOTA_APPLIED or os.environ.get('BISIP_KEEP_THREADS'):
        return
    _QUOTA_APPLIED = True
    n = cpu_quota()
    try:   # one process per GPU under torchrun: the ranks of this node share the quota
        n = max(1, n // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1'))))
    except ValueError:
        pass
    if n >= (os.cpu_count() or 1):
        return
    try:
        import torch
        if torch.get_num_threads() > n:
            torch.set_num_threads(n)
    except Exception:
        pass
    try:
        from threadpoolctl import threadpool_limits
        respect_cpu_quota._limits = threadpool_limits(limits=n)   # kept alive: the cap stays
    except Exception:
        pass
