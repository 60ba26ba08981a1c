#!/usr/bin/env python3
"""Bulk log-probability of a batch of spectra (512 x 32768 walkers, device resident) against the
single-spectrum kernels' rates (profiles/r02_bench.json `kernels`): evals/s and fraction of HBM peak."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import bisip_amd
from bisip_amd.synthetic import synthetic_columns
E_, Wp = 512, 32768
tables = [synthetic_columns(32, i) for i in range(E_)]
st = torch.cuda.current_stream().cuda_stream
for model, kw, variants in (('PolynomialDecomposition', dict(poly_deg=5), ('auto', 'reduced_comp', 'collapsed')),
                            ('PeltonColeCole', dict(n_modes=1), ('auto',)), ('PeltonColeCole', dict(n_modes=2), ('auto',)),
                            ('Dias2000', {}, ('auto',)), ('Shin2015', {}, ('auto',))):
    b = bisip_amd.SpectraBatch(model, tables, nwalkers=64, **kw)
    lo, hi = b.param_bounds
    nd = lo.size
    W = E_ * Wp if model == 'PolynomialDecomposition' else E_ * Wp // 4
    th = torch.from_numpy(np.random.RandomState(0).uniform(lo, hi, (W, nd))).cuda()
    out = torch.empty(W, dtype=torch.float64, device='cuda')
    for variant in variants:
        b.ctx.set_variant(variant)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            b.ctx.logprob_dev(th.data_ptr(), W, out.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): b.ctx.logprob_dev(th.data_ptr(), W, out.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f'{model} {kw} {variant}: {b.ctx.kernel_name}  {W} rows  {ms*1e3:.1f} us  {W/ms*1e3:.3e} evals/s  {8*(nd+1)*W/ms/1e9*1e3/8000:.3f} of HBM', flush=True)
    b.close()
