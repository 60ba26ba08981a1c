#!/usr/bin/env python3
"""The post-run delay without any of this package's code: torch alone.  ~0.3 s of device work, a
synchronisation, a few device->host copies, then a small kernel timed from launch to the end of the
synchronisation.  (Pure HIP against the system runtime never shows the delay: idle_queue_wakeup.hip.)"""
import sys
import time

import torch

big = sys.argv[1] if len(sys.argv) > 1 else 'small'
x = torch.zeros(1 << 20, dtype=torch.float64, device='cuda')
a = torch.randn(4096, 4096, device='cuda')
chain = torch.empty((500 if big == 'big' else 8, 131072, 7), dtype=torch.float64, device='cuda')
torch.cuda.synchronize()
slow, worst = 0, 0.0
for rep in range(30):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(20):
            b = a @ a
        chain[rep % chain.shape[0]].fill_(1.0)
    torch.cuda.synchronize()
    _ = x[:4].cpu(); _ = chain[0, :1000].cpu()
    t1 = time.perf_counter()
    x.add_(1.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) * 1e3
    worst = max(worst, ms)
    slow += ms > 2.0
print(f'torch only ({big} footprint): small kernel launch->sync worst {worst:.3f} ms, {slow} of 30 above 2 ms')
