#!/usr/bin/env python3
"""Walkers sharded over the GPUs of a node, one process per GPU (no collective on the
log-probability path; one all-gather only to hand every rank the full result):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        examples/multi_gpu_logprob.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import numpy as np
import torch
import torch.distributed as dist

import bisip_amd
from bisip_amd.dist import ShardedLogProb

rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
local = int(os.environ.get('LOCAL_RANK', 0))
torch.cuda.set_device(local)
if 'RANK' in os.environ:
    dist.init_process_group('nccl', device_id=torch.device('cuda', local))
model = bisip_amd.PeltonColeCole(bisip_amd.DataFiles()['SIP-K389175'], n_modes=2, device=local)
theta = np.random.RandomState(0).uniform(*model.param_bounds, (1 << 20, model.param_bounds.shape[1]))
logp = ShardedLogProb(model.log_prob)(theta) if world > 1 else model.log_prob(theta)
if rank == 0:
    print(f'{world} rank(s): {logp.size} log-probabilities, best {np.nanmax(logp):.3f}')
if dist.is_initialized():
    dist.destroy_process_group()
