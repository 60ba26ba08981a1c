#!/usr/bin/env python3
"""A/B of two builds of libbisip_hip.so on ONE box (boxes of the pool differ by 5-10 % on the clock-throttled
kernels, more than most kernel changes are worth): the bulk log-probability kernels of the per-frequency
models at bench.py's shapes and one cfg5-shaped batch fit, each library in a process of its own.

    python benchmarks/micro/ab_library.py [path/to/libbisip_hip.so]     # one JSON line
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bisip_amd import _hip                                  # noqa: E402  (nothing is loaded yet)

if len(sys.argv) > 1:
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np                                          # noqa: E402
import torch                                                # noqa: E402
import bench                                                # noqa: E402

out = {'library': os.path.relpath(_hip.LIB_PATH, ROOT)}
data, _, _, _ = bench.make_problem()
stream = torch.cuda.current_stream()
for label, ctx, th, o, ndim in bench.zoo_contexts(data, 0, bench.ZOO_WALKERS):
    def fn():
        ctx.logprob_dev(th.data_ptr(), bench.ZOO_WALKERS, o.data_ptr(), stream.cuda_stream)
    bench.prime(fn, 0.3, torch)
    best = min(bench.time_launches(fn, 10, 2, torch, stream)[1] for _ in range(3))
    out[label] = {'evals_per_s': float('%.4g' % (bench.ZOO_WALKERS / (best * 1e-3))), 'kernel_ms': round(best, 4),
                  'clock_ghz': round(bench.engine_clock_ghz(fn, best, torch) or 0.0, 3)}
    ctx.close()

# cfg5's slice: 512 double-Cole-Cole spectra x 256 walkers on the persistent kernel
import bisip_amd                                            # noqa: E402
from bisip_amd.synthetic import synthetic_columns           # noqa: E402
E_, Wp, its = 512, 256, 4000
b = bisip_amd.SpectraBatch('PeltonColeCole', [synthetic_columns(32, i) for i in range(E_)], nwalkers=Wp,
                           nsteps=its // 40, n_modes=2)
p0 = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]) + 1e-3 * np.random.RandomState(0).randn(E_, Wp, 7)
b.fit(p0, seed=3, thin_by=40, chain='device')
best = None
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.fit(p0, seed=3, thin_by=40, chain='device')
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None or dt < best else best
out['cfg5_slice'] = {'us_per_half_step': round(best / its / 2 * 1e6, 3), 'path': b._sampler.last_path}
b.close()
print(json.dumps(out))
