"""Walker sharding across the GPUs of one node (one process per GPU, RCCL over xGMI).

The log-probability has no cross-walker term (reference src/bisip/models.py:59-76),
so walkers shard naturally:

* log-prob / bench mode -- contiguous blocks of theta rows per rank, operands
  replicated in each rank's context, NO collective on the data path; outputs
  concatenate in rank order, so walker index i keeps its place.
* sampler mode -- every rank keeps the whole ensemble (W*ndim doubles: at most a few
  MB) and the same RNG stream, but evaluates the log-probability only for its block
  of the active half.  After the accept step each rank owns the new positions and
  log-probs of its block; ONE all-gather per half-step (payload (ndim+1) doubles per
  active walker) rebuilds the full ensemble everywhere.  ``backend='nccl'`` is RCCL on
  ROCm; the CPU tests run the same code over ``gloo``.
"""

import numpy as np


def shard_range(n_rows, world, rank):
    """Contiguous block [start, stop) of rank `rank`; the first n_rows % world ranks
    get one extra row."""
    n_rows, world, rank = int(n_rows), int(world), int(rank)
    base, extra = divmod(n_rows, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(n_rows, world):
    return [shard_range(n_rows, world, r)[1] - shard_range(n_rows, world, r)[0]
            for r in range(world)]


class ShardedLogProb:
    """Vectorised log-probability whose rows are split across the ranks of a group.

    ``local_fn(theta_rows) -> logp_rows`` runs on this rank's device.  Calling the
    object with the full (replicated) theta returns the full logp on every rank:
    one all-gather of 8 B per walker.  ``local(theta)`` evaluates only this rank's
    block (the bench / pure log-prob mode: no collective).
    """

    def __init__(self, local_fn, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.local_fn = local_fn
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def local(self, theta):
        a, b = shard_range(len(theta), self.world, self.rank)
        return self.local_fn(theta[a:b])

    def __call__(self, theta):
        import torch
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        n = theta.shape[0]
        mine = np.asarray(self.local(theta), dtype=np.float64)
        if self.world == 1:
            return mine
        pad = -(-n // self.world)  # equal-size slots
        device = 'cuda' if self._dist.get_backend(self.group) == 'nccl' else 'cpu'
        buf = torch.full((pad,), float('nan'), dtype=torch.float64, device=device)
        buf[:mine.size] = torch.from_numpy(mine).to(device)
        out = torch.empty(pad * self.world, dtype=torch.float64, device=device)
        self._dist.all_gather_into_tensor(out, buf, group=self.group)
        out = out.cpu().numpy().reshape(self.world, pad)
        sizes = shard_sizes(n, self.world)
        return np.concatenate([out[r, :sizes[r]] for r in range(self.world)])


def all_gather_rows(block, n_total, group=None):
    """All-gather unevenly sized row blocks (numpy (n_r, k)) into the full (n_total, k)
    array in rank order -- the per-half-step exchange of the sharded sampler."""
    import torch
    import torch.distributed as dist
    block = np.ascontiguousarray(block, dtype=np.float64)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return block
    world = dist.get_world_size(group)
    k = block.shape[1]
    pad = -(-int(n_total) // world)
    device = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
    buf = torch.zeros((pad, k), dtype=torch.float64, device=device)
    buf[:block.shape[0]] = torch.from_numpy(block).to(device)
    out = torch.empty((world * pad, k), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.cpu().numpy().reshape(world, pad, k)
    sizes = shard_sizes(n_total, world)
    return np.concatenate([out[r, :sizes[r]] for r in range(world)], axis=0)
