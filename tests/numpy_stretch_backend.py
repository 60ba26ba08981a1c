"""CPU stand-in for bisip_amd.sampler.HipStretchBackend, used ONLY by tests to run the
multi-rank driver logic of DeviceEnsembleSampler over gloo.  Implements the semantics
of bisip_stretch_{half,eval,apply}_dev (include/bisip_hip.h) in NumPy on CPU tensors,
with the oracle as the log-probability."""

import numpy as np
import torch


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 on uint64 arrays holding 32-bit values (independent of
    the C implementation in bisip_amd/csrc/philox.h)."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    mask = np.uint64(0xffffffff)
    c0, c1, c2, c3 = [np.asarray(x, dtype=np.uint64) & mask for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask, p1 & mask, \
                         ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask, p0 & mask
        k0 = (k0 + np.uint64(W0)) & mask
        k1 = (k1 + np.uint64(W1)) & mask
    return c0, c1, c2, c3


def u53(a, b):
    return (((a >> np.uint64(5)) << np.uint64(26)) | (b >> np.uint64(6))).astype(np.float64) / 9007199254740992.0


def philox_stream(W, ndim, a, seed, step0, perm, E=1, e0=0):
    """NumPy statement of the rng='philox' contract (bisip_amd/csrc/sampler_kernels.h):
    returns active, partner, zz, factor, logu of shape (n, 2, E*nh); W = walkers per
    ensemble, walker ids are global (e*W + i); e0 = survey index of ensemble 0 (keys the counter)."""
    n = perm.shape[0]
    nh = (W + 1) // 2
    out = dict(active=np.zeros((n, 2, E, nh), np.int32), partner=np.zeros((n, 2, E, nh), np.int32),
               zz=np.ones((n, 2, E, nh)), factor=np.zeros((n, 2, E, nh)), logu=np.zeros((n, 2, E, nh)))
    k0, k1 = seed & 0xffffffff, seed >> 32
    for k in range(n):
        A, Ainv, B = [int(x) for x in perm[k]]
        for h in (0, 1):
            Ns = nh if h == 0 else W // 2
            Nc = W // 2 if h == 0 else nh
            t = np.arange(Ns, dtype=np.uint64)
            ti = t.astype(np.int64)
            for e in range(E):
                c2 = h | ((e0 + e) << 1)
                x0, x1, x2, _ = philox4x32_10(t, step0 + k, c2, 0, k0, k1)
                y0, y1, _, _ = philox4x32_10(t, step0 + k, c2, 1, k0, k1)
                r = ((x2 * np.uint64(Nc)) >> np.uint64(32)).astype(np.int64)
                v = (a - 1.0) * u53(x0, x1) + 1.0
                z = (v * v) / a
                out['active'][k, h, e, :Ns] = e * W + (Ainv * ((2 * ti + h - B) % W)) % W
                out['partner'][k, h, e, :Ns] = e * W + (Ainv * ((2 * r + (1 - h) - B) % W)) % W
                out['zz'][k, h, e, :Ns] = z
                out['factor'][k, h, e, :Ns] = (ndim - 1.0) * np.log(z)
                with np.errstate(divide='ignore'):
                    out['logu'][k, h, e, :Ns] = np.log(u53(y0, y1))
    return {name: arr.reshape(n, 2, E * nh) for name, arr in out.items()}


class NumpyStretchBackend:
    def __init__(self, logprob_fn, n_ensembles=1):
        self.logprob_fn = logprob_fn
        self.n_ensembles = n_ensembles

    def tensor(self, array, dtype=None, slot='a'):
        return torch.as_tensor(np.ascontiguousarray(array), dtype=dtype).clone()

    def empty(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    zeros = empty

    def logprob(self, coords_t, out_t):
        out_t[:] = torch.from_numpy(self.logprob_fn(coords_t.numpy()))

    def synchronize(self):
        pass

    def flag_nan(self, logp_t, status_t):
        if torch.isnan(logp_t).any():
            status_t[0] |= 2

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
        thin = st.get('thin', 1)
        if (k + 1) % thin == 0:
            st['chain'].numpy()[k // thin][idx] = rows
            st['logp_chain'].numpy()[k // thin][idx] = lps
        st['naccept'].numpy()[idx[acc]] += 1

    def run_persistent(self, st, wp, n_steps):
        if wp * (st['coords'].shape[1] + 1) * 8 > 65536 or (wp + 1) // 2 > 512:
            return False
        self.run(st, n_steps)
        return True

    def stream_staging(self, n, nh, slot=0):
        kinds = (('active', torch.int32), ('partner', torch.int32), ('zz', torch.float64),
                 ('factor', torch.float64), ('logu', torch.float64))
        return {name: torch.zeros((n, 2, nh), dtype=dt) for name, dt in kinds}

    def upload_staged(self, stage, slot=0):
        return dict(stage)

    def host_buffer(self, shape):
        return torch.zeros(shape, dtype=torch.float64)

    def copy_out(self, dst_host, src_dev):
        dst_host.copy_(src_dev)

    def half(self, st, k, h, n_slots):
        self._commit(st, k, *self._slot(st, k, h, n_slots, 0, n_slots))

    def run(self, st, n_steps):
        W = st['coords'].shape[0]
        for k in range(n_steps):
            self.half(st, k, 0, (W + 1) // 2)
            self.half(st, k, 1, W // 2)

    def draw(self, st, W, a, seed, step0, n_steps):
        ndim = st['coords'].shape[1]
        arrs = philox_stream(W, ndim, a, seed, step0, st['perm'].numpy(), self.n_ensembles)
        for name, arr in arrs.items():
            st[name][:] = torch.from_numpy(arr)

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
