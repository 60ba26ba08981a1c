"""Device->host copy rate of a chain slab into pageable vs pinned host memory (why the device
sampler returns big chains through pinned buffers)."""
import torch, time, numpy as np
n = 200*32768*7
dev = torch.empty(n, dtype=torch.float64, device='cuda'); dev.normal_()
torch.cuda.synchronize()
for trial in range(2):
    t0=time.perf_counter(); host=np.empty(n); t1=time.perf_counter()
    torch.from_numpy(host).copy_(dev); torch.cuda.synchronize(); t2=time.perf_counter()
    print('pageable: alloc %.1f ms copy %.1f ms (%.1f GB/s)'%((t1-t0)*1e3,(t2-t1)*1e3, n*8/(t2-t1)/1e9))
    t0=time.perf_counter(); pin=torch.empty(n, dtype=torch.float64, pin_memory=True); t1=time.perf_counter()
    pin.copy_(dev, non_blocking=True); torch.cuda.synchronize(); t2=time.perf_counter()
    print('pinned:   alloc %.1f ms copy %.1f ms (%.1f GB/s)'%((t1-t0)*1e3,(t2-t1)*1e3, n*8/(t2-t1)/1e9))
    del pin
