"""Kernel time of the headline launch on the null stream vs a side stream, round after round:
shows the 5-8 % warm-up effect (first ~0.1 s after idle) that bench.py primes away."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'benchmarks'))
import torch, numpy as np
from sweep import problem
from bisip_amd.synthetic import synthetic_theta
ctx, bounds = problem('pd', 32, variant='reduced')
W = 1 << 24
theta = torch.from_numpy(synthetic_theta(bounds[0], bounds[1], W)).cuda()
out = torch.empty(W, dtype=torch.float64, device='cuda')
def run(stream, reps=50):
    with torch.cuda.stream(stream):
        st = torch.cuda.current_stream()
        for _ in range(5): ctx.logprob_dev(theta.data_ptr(), W, out.data_ptr(), st.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(st)
        for _ in range(reps): ctx.logprob_dev(theta.data_ptr(), W, out.data_ptr(), st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
side = torch.cuda.Stream()
for r in range(4):
    print('null stream %.1f us   side stream %.1f us' % (run(torch.cuda.default_stream()), run(side)))
