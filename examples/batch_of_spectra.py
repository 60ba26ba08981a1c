#!/usr/bin/env python3
"""A survey of many spectra inverted together (BASELINE config 5 in miniature): E synthetic
double-Cole-Cole spectra x 128 walkers each, all ensembles advanced by the same launches, the
chain kept in HBM and summarised there (mean / std / percentiles per spectrum).

One GPU:        python examples/batch_of_spectra.py
Several GPUs:   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                    examples/batch_of_spectra.py
(spectra shard as whole ensembles, E/N per rank; nothing is exchanged while they run, and one
gather at the end hands every rank the whole survey's summaries)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

import numpy as np

import bisip_amd
from bisip_amd.synthetic import synthetic_columns

rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
local = int(os.environ.get('LOCAL_RANK', 0))
if world > 1:
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))

E, Wp = 64, 128
tables = [synthetic_columns(32, i) for i in range(E)]          # or a list of data-file paths
batch = bisip_amd.SpectraBatch('PeltonColeCole', tables, nwalkers=Wp, nsteps=200, n_modes=2,
                               device=local, rank=rank, world=world)     # this rank's block of spectra
first, last = batch.spectrum_range
rng = np.random.RandomState(0)
p0 = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]) + 1e-3 * rng.randn(E, Wp, 7)
batch.fit(p0[first:last], seed=1 + rank, thin_by=5, chain='device')      # 1000 iterations, 200 stored

mean = batch.gather(batch.get_param_mean(discard=100))         # (E, ndim): computed on the device, gathered once
std = batch.gather(batch.get_param_std(discard=100))
pct = batch.get_param_percentile([16, 50, 84], discard=100)    # (3, E_rank, ndim)
p16, p50, p84 = np.moveaxis(batch.gather(np.moveaxis(pct, 1, 0)), 1, 0)
band = batch.get_model_percentile([2.5, 50, 97.5], discard=100)         # (3, E_rank, 2, N): the band a fit is plotted with
accept = batch.gather(batch.acceptance_fraction.mean(axis=1))
if rank == 0:
    print('parameters', batch.param_names)
    for e in (0, 1, E - 1):
        print(f'spectrum {e:3d}  mean {np.round(mean[e], 3)}  median {np.round(p50[e], 3)}')
    print('acceptance', round(float(accept.mean()), 3))
    print('95 % band of Re Z at the lowest frequency, first spectrum of this rank:', np.round(band[[0, 2], 0, 0, -1], 4))
batch.close()
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
