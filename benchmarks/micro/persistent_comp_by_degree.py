#!/usr/bin/env python3
"""The compensated QR-reduced kernel in the persistent sampler against one launch per half-step, by polynomial
degree and ensemble size (bundled spectrum SIP-K389175, 1000 iterations, best of three): the measurement behind
HipContext.persistent_walkers for 'reduced_comp' (persistent up to 1024 walkers at degree <= 5, up to 512 above)."""
import os
import sys, time, numpy as np, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bisip_amd
warnings.simplefilter('ignore')
path = bisip_amd.DataFiles()['SIP-K389175']
for kw in (dict(poly_deg=5,c_exp=0.5), dict(poly_deg=6,c_exp=1.0,variant='reduced_comp'), dict(poly_deg=7,c_exp=1.0), dict(poly_deg=8,c_exp=1.0), dict(poly_deg=9,c_exp=1.0), dict(poly_deg=10,c_exp=1.0)):
    for W in (32, 128, 256, 512, 1024):
        row=[]
        for pers in (True, False):
            m = bisip_amd.PolynomialDecomposition(path, nwalkers=W, nsteps=1000, **kw)
            np.random.seed(1); m.fit(persistent=pers)
            best=1e9
            for _ in range(3):
                np.random.seed(1); t=time.perf_counter(); m.fit(persistent=pers); best=min(best,time.perf_counter()-t)
            row.append('%s %.0f it/s'%(m.sampler.last_path, m.nsteps/best))
        print(kw, W, m._context().kernel_name, row, flush=True)
