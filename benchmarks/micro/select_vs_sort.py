#!/usr/bin/env python3
"""Percentiles of 2048 x 64 columns of 12,800 values (the model-space bands of 2048 spectra: 1.7e9 values):
milliseconds per call by selection (default) and by the segmented sort (BISIP_PERCENTILE_SORT=1)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bisip_amd import _hip

G, n, cols = 2048, 12800, 64
x = 1.0 + 0.01 * torch.randn(G, n, cols, dtype=torch.float64, device='cuda')
p = np.array([2.5, 50, 97.5])
nb = _hip.grouped_percentiles_workspace(G, n, cols, 3)
work = torch.empty(nb, dtype=torch.uint8, device='cuda')
out = torch.empty(3, G, cols, dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
res = {}
for mode in ('selection', 'sort'):
    if mode == 'sort':
        os.environ['BISIP_PERCENTILE_SORT'] = '1'
    for _ in range(3):
        _hip.grouped_percentiles_dev(x.data_ptr(), G, n, cols, p, out.data_ptr(), work.data_ptr(), nb, st)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        _hip.grouped_percentiles_dev(x.data_ptr(), G, n, cols, p, out.data_ptr(), work.data_ptr(), nb, st)
    torch.cuda.synchronize()
    res[mode] = out.cpu().numpy().copy()
    print(f'{mode}: {(time.perf_counter() - t) / 5 * 1e3:.1f} ms per call (gather + order statistics + interpolation)')
print('same doubles:', bool(np.array_equal(res['selection'], res['sort'])))
