"""Geometric frequency grid (kernels.h: BOUNDS_GRID) against one exponential per (frequency, term):
same contexts, BISIP_NO_GRID=1 for the second of each pair.  Prints evaluations/s of the bulk launch and
the largest difference between the two, relative to max(1, |logp|)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import torch
from bisip_amd import _hip
from bisip_amd.batch import default_params
from bisip_amd.synthetic import synthetic_columns
from bisip_amd.utils import columns_to_data

MODELS = [('PeltonColeCole', 1, dict(n_modes=1)), ('PeltonColeCole', 1, dict(n_modes=2)), ('PeltonColeCole', 1, dict(n_modes=3)),
          ('Shin2015', 3, {})]
W = 1 << 22
import bisip_amd
from bisip_amd.utils import load_data
CASES = []
for N in (32, 30, 21):
    CASES.append((f'N={N} geometric grid', columns_to_data(synthetic_columns(N, 0), 'mrad'), 3))
# off any grid (frequencies as a data file rounds them, the bundled field spectrum): both columns run the
# same loop -- the control
cols = synthetic_columns(32, 0)
cols[:, 0] = np.array([float('%.6g' % f) for f in cols[:, 0]])
CASES.append(('N=32 grid rounded to 6 digits (no grid: control)', columns_to_data(cols, 'mrad'), 1))
CASES.append(('N=20 bundled SIP-K389175 (no grid: control)', load_data(bisip_amd.DataFiles()['SIP-K389175']), 1))
for label, d, flags in CASES:
    for name, mid, kw in MODELS:
        bounds = np.array(list(default_params(name, **kw).values()), float).T
        rng = np.random.RandomState(1)
        theta = rng.uniform(bounds[0], bounds[1], (W, bounds.shape[1]))
        th = torch.from_numpy(theta).cuda()
        out = {}
        for grid in (True, False):
            if grid: os.environ.pop('BISIP_NO_GRID', None)
            else: os.environ['BISIP_NO_GRID'] = '1'
            ctx = _hip.HipContext(mid, d['w'], d['zn'], d['zn_err'], bounds, **kw)
            assert ctx.loop_flags == (flags if grid else 1)
            lp = torch.empty(W, dtype=torch.float64, device='cuda')
            for _ in range(3): ctx.logprob_dev(th.data_ptr(), W, lp.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): ctx.logprob_dev(th.data_ptr(), W, lp.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
            out[grid] = (lp.cpu().numpy(), W / dt)
            small = ctx.logprob(theta[:1000])           # L = 4 lanes per walker
            mid_ = ctx.logprob(theta[:20000])           # L = 2 or 1
            assert np.array_equal(small, out[grid][0][:1000]), 'L=4 differs from L=1'
            assert np.array_equal(mid_, out[grid][0][:20000]), 'mid-size launch differs'
            ctx.close()
        a, b = out[True][0], out[False][0]
        rel = np.max(np.abs(a - b) / np.maximum(1, np.abs(b)))
        print(f'{label} {name} {kw}: grid {out[True][1]:.3e} evals/s, direct {out[False][1]:.3e} evals/s, ratio {out[True][1]/out[False][1]:.2f}, max rel diff {rel:.2e}', flush=True)
