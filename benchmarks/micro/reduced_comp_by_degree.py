#!/usr/bin/env python3
"""Bulk rate of the QR-reduced kernels by polynomial degree: a lone spectrum (2^23 walkers; compensated operands
as kernel arguments up to degree 5, from memory from degree 6 on) and a batch of 512 spectra x 16,384 walkers
(operands from memory), plain and compensated tier.  1e10 evals/s."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, bisip_amd
from bisip_amd import _hip
from bisip_amd.synthetic import synthetic_columns

W = 1 << 23
st = torch.cuda.current_stream().cuda_stream


def rate(ctx, th, out):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        ctx.logprob_dev(th.data_ptr(), W, out.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctx.logprob_dev(th.data_ptr(), W, out.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    return W / (e0.elapsed_time(e1) / 10) * 1e3 / 1e10


tables = [synthetic_columns(32, i) for i in range(512)]
for P in range(3, 11):
    b = bisip_amd.SpectraBatch('PolynomialDecomposition', tables, nwalkers=64, poly_deg=P)
    lo, hi = b.param_bounds
    th = torch.from_numpy(np.random.RandomState(0).uniform(lo, hi, (W, lo.size))).cuda()
    out = torch.empty(W, dtype=torch.float64, device='cuda')
    res = []
    for v in ('reduced', 'reduced_comp'):
        b.ctx.set_variant(v)
        res.append(f'batch {v} {rate(b.ctx, th, out):.2f}')
    b.close()
    data, taus, log_taus, bounds = bench.make_problem(poly_deg=P)
    ctx = _hip.HipContext(_hip.MODEL_POLYDECOMP, data['w'], data['zn'], data['zn_err'], bounds, poly_deg=P, c_exp=1.0,
                          taus=taus, log_taus=log_taus)
    for v in ('reduced', 'reduced_comp'):
        ctx.set_variant(v)
        res.append(f'single {v} {rate(ctx, th, out):.2f}')
    ctx.close()
    print('degree', P, '|', ' | '.join(res), flush=True)
    del th, out
