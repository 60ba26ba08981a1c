#!/usr/bin/env python3
"""One big ensemble, its walkers sharded over the GPUs of a node (BASELINE config 4's shape): one
process per GPU, every rank holds the whole ensemble and the same random stream, evaluates its
block of each half-step, and one RCCL all-gather per half-step rebuilds the state.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        examples/multi_gpu_sampler.py [python|rccl|rccl-own]

The argument picks who drives the half-steps: 'python' (default) = eval / all_gather_into_tensor / apply
from Python over torch.distributed; 'rccl' = all of a chunk enqueued by one C call
(bisip_stretch_run_sharded_dev) on the communicator torch.distributed made, 'rccl-own' = on one
of the sampler's own.  The C loop is 5x faster with one rank but has not yet run with more than
one rank on hardware, hence not the default.

Every rank ends with the same chain, bit for bit the chain a single GPU would have produced.
(At this size one GPU is the faster machine -- DESIGN.md §4 has the break-even; the sharded form
pays for far bigger ensembles or far more expensive log-probabilities.)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import numpy as np
import torch
import torch.distributed as dist

import bisip_amd
from bisip_amd.sampler import DeviceEnsembleSampler

rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
local = int(os.environ.get('LOCAL_RANK', 0))
torch.cuda.set_device(local)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29541')
dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))

W, nsteps = 32768, 100
model = bisip_amd.PolynomialDecomposition(bisip_amd.DataFiles()['SIP-K389175'], nwalkers=W, nsteps=nsteps, device=local)
centre = np.array([1.0, 0.005, -0.003, -0.001, 0.0005, 0.0002, 0.00001])
p0 = centre + 1e-4 * np.random.RandomState(2024).randn(W, 7)      # the same start on every rank

np.random.seed(7)                                                  # the same stream on every rank
sampler = DeviceEnsembleSampler(W, 7, model._context(), rng='philox', seed=11, distributed=True,
                                force_sharded_path=True, persistent=False, chain_on_device=True,
                                sharded_loop=sys.argv[1] if len(sys.argv) > 1 else 'python')
sampler.run_mcmc(p0, nsteps)
mean, std = sampler.param_moments(discard=nsteps // 2)
if rank == 0:
    print(f'{world} rank(s), driver {sampler.last_path}: acceptance {sampler.acceptance_fraction.mean():.3f}')
    print('posterior mean', np.round(mean[0], 5))
    print('posterior std ', np.round(std[0], 5))
sampler.close()
dist.barrier()
dist.destroy_process_group()
