#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (bisip_logprob / bisip_forward):
never the headline `value`, reported in DESIGN.md §3.5."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'benchmarks'))
from sweep import problem
from bisip_amd.synthetic import synthetic_theta
import torch

for name, model, N, W, kw in [('PD reduced N32', 'pd', 32, 1 << 22, {}), ('CC D2 N32', 'cc', 32, 1 << 22, dict(n_modes=2)),
                              ('PD N20 (the bundled spectra)', 'pd', 20, 1 << 22, {}), ('CC D2 N64', 'cc', 64, 1 << 21, dict(n_modes=2))]:
    ctx, bounds = problem(model, N, **kw)
    theta = synthetic_theta(bounds[0], bounds[1], W)
    ctx.logprob(theta[:1024])
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); out = ctx.logprob(theta); ts.append(time.perf_counter() - t0)
    dt = min(ts)
    print(json.dumps({'case': name + ' bisip_logprob (host buffers, pageable)', 'W': W,
                      'evals_per_s': float('%.4g' % (W / dt)), 'GBs_over_pcie': round(W * 8 * (bounds.shape[1] + 1) / dt / 1e9, 2)}))
    # forward kernel, device resident
    Wf = min(1 << 21, W)
    th = torch.from_numpy(theta[:Wf]).cuda()
    Z = torch.empty((Wf, 2, N), dtype=torch.float64, device='cuda')
    st = torch.cuda.current_stream()
    t_prime = time.perf_counter()          # prime the clocks: ~0.3 s of back-to-back launches (as bench.py does)
    while time.perf_counter() - t_prime < 0.3:
        for _ in range(20):
            ctx.forward_dev(th.data_ptr(), Wf, Z.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20):
        ctx.forward_dev(th.data_ptr(), Wf, Z.data_ptr(), st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(json.dumps({'case': name + ' bisip_forward_dev', 'W': Wf, 'us': round(ms * 1e3, 1),
                      'rows_per_s': float('%.4g' % (Wf / ms * 1e3)),
                      'write_GBs': round(Wf * 16 * N / ms / 1e6, 1)}))
