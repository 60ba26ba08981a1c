#!/usr/bin/env python3
"""A 512 x 256 batch of PolynomialDecomposition spectra of degree 5-8 (AUTO runs the compensated kernel on such a
batch): persistent sampler kernel against one launch per half-step.  The other half of the measurement behind
HipContext.persistent_walkers for 'reduced_comp' (benchmarks/micro/persistent_comp_by_degree.py)."""
import sys, os, time, numpy as np, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bisip_amd
from bisip_amd.synthetic import synthetic_columns
warnings.simplefilter('ignore')
E, Wp = 512, 256
tables = [synthetic_columns(32, i) for i in range(E)]
for P in (5, 6, 7, 8, 9):
    row = []
    for pers in (True, False):
        b = bisip_amd.SpectraBatch('PolynomialDecomposition', tables, nwalkers=Wp, nsteps=50, poly_deg=P)
        b.fit(seed=3, thin_by=40, chain='device', persistent=pers)
        best = 1e9
        for _ in range(2):
            t = time.perf_counter(); b.fit(seed=3, thin_by=40, chain='device', persistent=pers); best = min(best, time.perf_counter() - t)
        row.append('%s %.2f us/half-step' % (b._sampler.last_path, best / (50 * 40 * 2) * 1e6))
        k = b.ctx.kernel_name + ' plain/comp %s' % (b.ctx.reduced_tiers,)
        b.close()
    print(P, k, row, flush=True)
