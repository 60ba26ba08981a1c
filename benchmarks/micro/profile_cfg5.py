#!/usr/bin/env python3
"""cProfile of benchmarks/cfg5_batch.py: where the host side of a batch run spends its time.
Writes gpurun_out/cfg5_profile.txt."""
import cProfile, io, os, pstats, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ['cfg5_batch.py'] + sys.argv[1:]
import torch, numpy, bisip_amd  # noqa: imports out of the profile
from bisip_amd import sampler as _s, batch as _b, _hip as _h, synthetic as _y  # noqa
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(ROOT, 'benchmarks', 'cfg5_batch.py'), run_name='__main__')
finally:
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(40)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    open(os.path.join(ROOT, 'gpurun_out', 'cfg5_profile.txt'), 'w').write(s.getvalue())
