#!/usr/bin/env python3
"""bisip_chain_moments_dev over a cfg5-shaped device chain (512 spectra x 256 walkers x 7) for a
growing number of samples: time per call and read rate (two passes over the used samples)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bisip_amd import _hip

E, Wp, nd = 512, 256, 7
W = E * Wp
nmax = 500
chain = torch.randn(nmax, W, nd, dtype=torch.float64, device='cuda')
mean = torch.empty(E, nd, dtype=torch.float64, device='cuda')
std = torch.empty_like(mean)
st = torch.cuda.current_stream().cuda_stream
for n in (32, 64, 125, 160, 200, 250, 320, 400, 500):
    work = torch.empty(max(1, _hip.chain_moments_workspace(n, E, nd)), dtype=torch.float64, device='cuda')
    for first in (0, nmax - n):
        ptr = chain.data_ptr() + 8 * first * W * nd
        for _ in range(2):
            _hip.chain_moments_dev(ptr, n, W * nd, E, Wp, nd, mean.data_ptr(), std.data_ptr(), work.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            _hip.chain_moments_dev(ptr, n, W * nd, E, Wp, nd, mean.data_ptr(), std.data_ptr(), work.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        gb = 2 * n * W * nd * 8 / 1e9
        ref = chain[first:first + n].reshape(n, E, Wp, nd).mean(dim=(0, 2))
        ok = torch.allclose(mean, ref, rtol=1e-10, atol=1e-12)
        print(f'n={n:4d} first={first:4d}  {ms:8.3f} ms  {gb / ms:7.2f} TB/s  ok={ok}', flush=True)
