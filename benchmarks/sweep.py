#!/usr/bin/env python3
"""Kernel-level sweep over the BASELINE configs and model families (1 GPU).

Not the headline bench (that is ../bench.py); this prints one JSON line per case with
evals/s and the HBM-roofline fraction from algorithmic bytes 8*(ndim+1) per eval.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def problem(model, n_freq, **kw):
    from bisip_amd.synthetic import synthetic_columns
    from bisip_amd.utils import columns_to_data
    from bisip_amd import _hip
    data = columns_to_data(synthetic_columns(n_freq, 0), 'mrad')
    if model == 'pd':
        P = kw.get('poly_deg', 5)
        per = np.log10(1. / data['w'])
        lt = np.linspace(np.floor(per.min() - 1), np.floor(per.max() + 1), 2 * n_freq)
        bounds = np.array([[0.9] + [-1.0] * (P + 1), [1.1] + [1.0] * (P + 1)])
        ctx = _hip.HipContext(0, data['w'], data['zn'], data['zn_err'], bounds, poly_deg=P,
                              c_exp=kw.get('c_exp', 1.0), taus=10 ** lt,
                              log_taus=np.array([lt ** i for i in range(P + 1)]),
                              variant=kw.get('variant', 'auto'))
    elif model == 'cc':
        D = kw.get('n_modes', 1)
        bounds = np.array([[0.9] + [0.0] * D + [-15.0] * D + [0.0] * D,
                           [1.1] + [1.0] * D + [5.0] * D + [1.0] * D])
        ctx = _hip.HipContext(1, data['w'], data['zn'], data['zn_err'], bounds, n_modes=D)
    elif model == 'dias':
        bounds = np.array([[0.9, 0, -20, 0, 0], [1.1, 1, 0, 150, 1.0]])
        ctx = _hip.HipContext(2, data['w'], data['zn'], data['zn_err'], bounds)
    else:
        bounds = np.array([[0, 0, -15, -7, 0, 0], [1, 1, -13, -5, 1, 1.0]])
        ctx = _hip.HipContext(3, data['w'], data['zn'], data['zn_err'], bounds)
    return ctx, bounds


CASES = [
    # name, model, N, W, kwargs
    ('metric PD P5 N32 reduced', 'pd', 32, 1 << 24, dict(variant='reduced')),
    ('metric PD P5 N32 collapsed', 'pd', 32, 1 << 24, dict(variant='collapsed')),
    ('metric PD P5 N32 faithful', 'pd', 32, 1 << 22, dict(variant='faithful')),
    ('metric PD P5 N32 wave-per-walker (north-star mapping)', 'pd', 32, 1 << 22, dict(variant='wave')),
    ('cfg3 PD P5 N64 W=65536 wave-per-walker', 'pd', 64, 65536, dict(variant='wave')),
    ('PD P5 N32 W=4096 wave-per-walker', 'pd', 32, 4096, dict(variant='wave')),
    ('PD P5 N32 W=4096 collapsed', 'pd', 32, 4096, dict(variant='collapsed')),
    ('PD P5 N32 reduced W=4096', 'pd', 32, 4096, dict(variant='reduced')),
    ('PD P5 N32 reduced W=65536', 'pd', 32, 65536, dict(variant='reduced')),
    ('PD P5 N32 reduced W=1M', 'pd', 32, 1 << 20, dict(variant='reduced')),
    ('cfg3 PD P5 N64 W=65536 reduced', 'pd', 64, 65536, dict(variant='reduced')),
    ('cfg3 PD P5 N64 W=65536 collapsed', 'pd', 64, 65536, dict(variant='collapsed')),
    ('cfg3 PD P5 N64 W=65536 faithful', 'pd', 64, 65536, dict(variant='faithful')),
    ('PD P5 N64 W=4M collapsed', 'pd', 64, 1 << 22, dict(variant='collapsed')),
    ('PD P5 N64 W=1M faithful', 'pd', 64, 1 << 20, dict(variant='faithful')),
    ('cfg4 PD P5 N20 W=4096 reduced', 'pd', 20, 4096, dict(variant='reduced')),
    ('cfg2 CC D1 N32 W=4096', 'cc', 32, 4096, dict(n_modes=1)),
    ('CC D1 N32 W=4M', 'cc', 32, 1 << 22, dict(n_modes=1)),
    ('CC D2 N32 W=16384', 'cc', 32, 16384, dict(n_modes=2)),
    ('CC D2 N32 W=65536', 'cc', 32, 65536, dict(n_modes=2)),
    ('cfg5 CC D2 N32 W=1M (=4096x256)', 'cc', 32, 1 << 20, dict(n_modes=2)),
    ('CC D2 N32 W=4M', 'cc', 32, 1 << 22, dict(n_modes=2)),
    ('CC D3 N32 W=4M', 'cc', 32, 1 << 22, dict(n_modes=3)),
    ('Dias N32 W=4M', 'dias', 32, 1 << 22, {}),
    ('Shin N32 W=4M', 'shin', 32, 1 << 22, {}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    import torch
    from bisip_amd.synthetic import synthetic_theta
    torch.cuda.set_device(0)
    for name, model, N, W, kw in CASES:
        if args.only and args.only not in name:
            continue
        ctx, bounds = problem(model, N, **kw)
        theta = torch.from_numpy(synthetic_theta(bounds[0], bounds[1], W)).cuda()
        out = torch.empty(W, dtype=torch.float64, device='cuda')
        st = torch.cuda.current_stream()
        # prime the clocks: the GPU idled while the host generated theta; ~0.25 s of back-to-back
        # launches before the timed ones (what bench.py does for the headline)
        import time
        t_prime = time.perf_counter()
        while time.perf_counter() - t_prime < 0.25:
            for _ in range(10):
                ctx.logprob_dev(theta.data_ptr(), W, out.data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(st)
        for _ in range(args.reps):
            ctx.logprob_dev(theta.data_ptr(), W, out.data_ptr(), st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        ndim = bounds.shape[1]
        evals = W / (ms * 1e-3)
        gbs = evals * 8 * (ndim + 1) / 1e9
        print(json.dumps({'case': name, 'kernel': ctx.kernel_name, 'N': N, 'W': W, 'ndim': ndim,
                          'us_per_launch': round(ms * 1e3, 2), 'evals_per_s': float('%.4g' % evals),
                          'hbm_GBs': round(gbs, 1), 'hbm_frac': round(gbs / 8000, 4)}), flush=True)
        ctx.close()
        del theta, out


if __name__ == '__main__':
    main()
