#!/usr/bin/env python3
"""Percentiles of long columns (more than 40,960 values: selection from memory, k_segmented_select) against the
segmented sort: parameter percentiles of a 512 x 256 x 250-sample chain, and 448 columns of 500,000 values."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bisip_amd import _hip
for (G, n, cols, centre) in ((512, 64000, 7, 1.0), (512, 64000, 7, 0.3), (64, 500000, 7, 1.0), (1, 4000000, 7, 1.0), (1, 30000000, 7, 1.0), (1, 2000000, 40, 1.0)):
    x = centre + 0.01 * torch.randn(G, n, cols, dtype=torch.float64, device='cuda')
    p = np.array([2.5, 50, 97.5])
    nb = _hip.grouped_percentiles_workspace(G, n, cols, 3)
    work = torch.empty(nb, dtype=torch.uint8, device='cuda'); out = torch.empty(3, G, cols, dtype=torch.float64, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    for mode in ('selection', 'sort'):
        os.environ.pop('BISIP_PERCENTILE_SORT', None)
        if mode == 'sort': os.environ['BISIP_PERCENTILE_SORT'] = '1'
        for _ in range(2): _hip.grouped_percentiles_dev(x.data_ptr(), G, n, cols, p, out.data_ptr(), work.data_ptr(), nb, st)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3): _hip.grouped_percentiles_dev(x.data_ptr(), G, n, cols, p, out.data_ptr(), work.data_ptr(), nb, st)
        torch.cuda.synchronize(); print(G, n, cols, centre, mode, round((time.perf_counter() - t) / 3 * 1e3, 2), 'ms')
    del x, work
