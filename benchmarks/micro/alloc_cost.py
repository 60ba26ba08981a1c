#!/usr/bin/env python3
"""Cost of the allocations a sampler run makes: device blocks through torch's caching
allocator (= hipMalloc on a miss) and pinned host blocks, by size.  Decides the chunk
budget in DeviceEnsembleSampler._chunk_steps."""
import time
import torch

torch.cuda.init()
torch.empty(1, device='cuda')
for mb in (64, 256, 1024, 2048, 4096, 8192):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = torch.empty(mb << 20, dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    del x
    torch.cuda.empty_cache()
    t2 = time.perf_counter()
    print(f'device {mb:5d} MiB: alloc {1e3 * (t1 - t0):8.2f} ms   free {1e3 * (t2 - t1):8.2f} ms')
for mb in (64, 256, 1024):
    t0 = time.perf_counter()
    x = torch.empty(mb << 20, dtype=torch.uint8, pin_memory=True)
    t1 = time.perf_counter()
    print(f'pinned {mb:5d} MiB: alloc {1e3 * (t1 - t0):8.2f} ms')
    del x
