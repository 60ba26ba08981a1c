"""CPU stand-in for bisip_amd.sampler.HipStretchBackend, used ONLY by tests to run the
multi-rank driver logic of DeviceEnsembleSampler over gloo.  Implements the semantics
of bisip_stretch_{half,eval,apply}_dev (include/bisip_hip.h) in NumPy on CPU tensors,
with the oracle as the log-probability."""

import numpy as np
import torch


class NumpyStretchBackend:
    def __init__(self, logprob_fn):
        self.logprob_fn = logprob_fn

    def tensor(self, array, dtype=None):
        return torch.as_tensor(np.ascontiguousarray(array), dtype=dtype).clone()

    def empty(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    zeros = empty

    def logprob(self, coords_t, out_t):
        out_t[:] = torch.from_numpy(self.logprob_fn(coords_t.numpy()))

    def synchronize(self):
        pass

    def _slot(self, st, k, h, n_slots, lo, hi):
        ndim = st['coords'].shape[1]
        coords, logp = st['coords'].numpy(), st['logp'].numpy()
        idx = st['active'][k, h, lo:hi].numpy()
        par = st['partner'][k, h, lo:hi].numpy()
        z = st['zz'][k, h, lo:hi].numpy()
        s, c = coords[idx], coords[par]
        q = c - (c - s) * z[:, None]
        new = self.logprob_fn(q)
        if np.any(np.isnan(new)):
            st['status'][0] |= 1
        old = logp[idx]
        acc = st['factor'][k, h, lo:hi].numpy() + new - old > st['logu'][k, h, lo:hi].numpy()
        rows = np.where(acc[:, None], q, s)
        return idx, rows, np.where(acc, new, old), acc

    def _commit(self, st, k, idx, rows, lps, acc):
        coords, logp = st['coords'].numpy(), st['logp'].numpy()
        coords[idx[acc]] = rows[acc]
        logp[idx[acc]] = lps[acc]
        st['chain'].numpy()[k][idx] = rows
        st['logp_chain'].numpy()[k][idx] = lps
        st['naccept'].numpy()[idx[acc]] += 1

    def half(self, st, k, h, n_slots):
        self._commit(st, k, *self._slot(st, k, h, n_slots, 0, n_slots))

    def eval(self, st, k, h, n_slots, lo, hi, block_t):
        idx, rows, lps, acc = self._slot(st, k, h, n_slots, lo, hi)
        blk = block_t.numpy()
        ndim = rows.shape[1]
        blk[:hi - lo, :ndim] = rows
        blk[:hi - lo, ndim] = lps
        blk[:hi - lo, ndim + 1] = acc

    def apply(self, st, k, h, n_slots, gathered_t, pad, world):
        g = gathered_t.numpy().reshape(world, pad, -1)
        base, extra = divmod(n_slots, world)
        parts = [g[r, :base + (1 if r < extra else 0)] for r in range(world)]
        full = np.concatenate(parts, axis=0)
        ndim = full.shape[1] - 2
        idx = st['active'][k, h, :n_slots].numpy()
        self._commit(st, k, idx, full[:, :ndim], full[:, ndim], full[:, ndim + 1] > 0)
