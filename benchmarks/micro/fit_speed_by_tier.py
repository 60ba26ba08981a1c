#!/usr/bin/env python3
"""fit() on the plain and on the compensated QR-reduced kernel: the price, in iterations/s, of the tier AUTO
picks where the plain triangle is not accurate enough on the shell logp = 0.  Bundled
spectrum SIP-K389175; best of three runs; the last column is fit()'s own after-the-fact measurement."""
import os
import sys, time, numpy as np, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bisip_amd
warnings.simplefilter('ignore')
path = bisip_amd.DataFiles()['SIP-K389175']
for kw in (dict(poly_deg=5,c_exp=1.0), dict(poly_deg=5,c_exp=0.5), dict(poly_deg=5,c_exp=0.5,variant='reduced'), dict(poly_deg=8,c_exp=1.0)):
    for W in (32, 256, 4096):
        m = bisip_amd.PolynomialDecomposition(path, nwalkers=W, nsteps=2000 if W<4096 else 200, **kw)
        np.random.seed(1); m.fit()
        best=1e9
        for _ in range(3):
            np.random.seed(1); t=time.perf_counter(); m.fit(); best=min(best,time.perf_counter()-t)
        print(kw, W, m._context().kernel_name, '%.0f it/s'%(m.nsteps/best), m.sampler.last_path, 'check %.1e'%m.reduced_check_)
