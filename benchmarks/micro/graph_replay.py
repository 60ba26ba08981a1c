#!/usr/bin/env python3
"""Does replaying the launch-per-half-step loop from a captured hipGraph beat enqueueing it?
bisip_stretch_run_dev enqueues 2 launches per iteration (≈3.5-7 us of CPU each); a captured
graph replays them with one call.  Prints the time per half-step both ways for a few
ensemble sizes (PolynomialDecomposition reduced and double Cole-Cole, philox stream)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'benchmarks'))
import torch
from sweep import problem
from bisip_amd.sampler import HipStretchBackend, affine_splits

for name, model, kw in (('PD reduced', 'pd', {}), ('CC D2', 'cc', dict(n_modes=2))):
    ctx, bounds = problem(model, 32, **kw)
    ndim = bounds.shape[1]
    be = HipStretchBackend(ctx)
    for W in (512, 4096, 32768):
        n = 200
        nh = W // 2
        rng = np.random.RandomState(W)
        p0 = 0.5 * (bounds[0] + bounds[1]) + 0.05 * (bounds[1] - bounds[0]) * (rng.rand(W, ndim) - 0.5)
        st = dict(coords=be.tensor(p0, torch.float64), logp=be.empty((W,), torch.float64),
                  naccept=be.zeros((W,), torch.int32), status=be.zeros((1,), torch.int32), nh=nh, thin=1)
        be.logprob(st['coords'], st['logp'])
        for nm, dt in (('active', torch.int32), ('partner', torch.int32), ('zz', torch.float64),
                       ('factor', torch.float64), ('logu', torch.float64)):
            st[nm] = be.empty((n, 2, nh), dt)
        st['perm'] = be.tensor(affine_splits(7, W, 0, n))
        st['chain'] = be.empty((n, W, ndim), torch.float64)
        st['logp_chain'] = be.empty((n, W), torch.float64)
        be.draw(st, W, 2.0, 7, 0, n)
        torch.cuda.synchronize()

        def direct():
            be.run(st, n)
        for _ in range(2):
            direct()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); direct(); t_enq = time.perf_counter() - t0
        torch.cuda.synchronize(); t_dir = time.perf_counter() - t0

        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                ctx.stretch_run_dev(be._args(st, 0, 0, (W + 1) // 2, base=True), W, n, 1,
                                    torch.cuda.current_stream().cuda_stream)
        t_cap = time.perf_counter() - t0
        graph.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter(); graph.replay(); t_launch = time.perf_counter() - t0
        torch.cuda.synchronize(); t_rep = time.perf_counter() - t0
        print(f'{name:10s} W={W:6d}  direct {1e6 * t_dir / (2 * n):6.2f} us/half-step (enqueue {1e6 * t_enq / (2 * n):5.2f})   '
              f'graph replay {1e6 * t_rep / (2 * n):6.2f} us/half-step (launch call {1e3 * t_launch:6.2f} ms, capture+instantiate {1e3 * t_cap:6.1f} ms)')
    ctx.close()
