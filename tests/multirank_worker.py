"""One rank of the multi-rank sampler check (started by tests/test_gpu_multirank.py under
`python -m torch.distributed.run`): the sharded stretch move -- every rank evaluates its block of each
half-step, ONE all-gather per half-step rebuilds the ensemble everywhere (bisip_amd/csrc/comm_rccl.hip, the
replacement of the reference's only parallel hook, fit(pool=...) -> emcee pool.map,
/root/reference/src/bisip/models.py:84,91-94,115) -- must give, on every rank, the chain of the fused
single-GPU run, bit for bit, whoever drives the loop:

  python     eval -> torch.distributed all_gather_into_tensor -> apply, per half-step, from Python
  rccl-own   bisip_stretch_run_sharded_dev (eval -> in-place ncclAllGather -> apply, enqueued from C) on a
             communicator of the sampler's own
  rccl       the same on the communicator torch.distributed built

over RCCL (one rank per GPU), or -- `--backend gloo --same-device`, what a one-GPU box can run -- the Python
loop with every rank on cuda:0 and the exchange through host memory.  Ensemble shapes: an odd number of walkers
(uneven halves and shards), fewer slots per half than ranks (a rank with nothing to evaluate), and BASELINE
config 4's 32,768 walkers.  Each rank writes its verdict to <out>/rank<r>.json.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'])
    ap.add_argument('--same-device', action='store_true')
    ap.add_argument('--out', required=True)
    ap.add_argument('--big', type=int, default=32768)
    args = ap.parse_args()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = 0 if args.same_device else int(os.environ.get('LOCAL_RANK', rank))
    torch.cuda.set_device(local)
    if args.backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    import bisip_amd
    from bisip_amd.sampler import DeviceEnsembleSampler

    verdict = {'rank': rank, 'world': world, 'backend': args.backend, 'cases': []}
    path = bisip_amd.DataFiles()['SIP-K389175']
    centre = np.array([1.0, 0.005, -0.003, -0.001, 0.0005, 0.0002, 0.00001])
    loops = ['python'] + (['rccl-own', 'rccl'] if args.backend == 'nccl' else [])
    # (walkers, stored steps, stream, thin_by): odd ensemble; fewer slots per half than ranks; cfg4
    cases = [(33, 12, 'numpy', 1), (2 * (world - 1) if world > 1 else 2, 9, 'philox', 1), (257, 6, 'philox', 2),
             (args.big, 6, 'philox', 1)]
    for W, steps, rng, thin in cases:
        m = bisip_amd.PolynomialDecomposition(path, nwalkers=max(W, 32), nsteps=steps, device=local)
        ctx = m._context()
        p0 = centre + 1e-4 * np.random.RandomState(7 + W).randn(W, 7)

        def run(**kw):
            np.random.seed(11)
            s = DeviceEnsembleSampler(W, 7, ctx, rng=rng, seed=5 if rng == 'philox' else None, persistent=False,
                                      live_dangerously=True, **kw)
            s.run_mcmc(p0, steps, thin_by=thin)
            out = (s.get_chain().copy(), s.get_log_prob().copy(), np.asarray(s.acceptance_fraction).copy(), s.last_path)
            s.close()
            return out

        fused = run()                                   # this rank alone: one launch per half-step
        rec = {'walkers': W, 'steps': steps, 'rng': rng, 'thin_by': thin, 'fused_path': fused[3], 'loops': {}}
        for loop in loops:
            got = run(distributed=True, sharded_loop=loop)
            same = bool(np.array_equal(got[0], fused[0]) and np.array_equal(got[1], fused[1]) and np.array_equal(got[2], fused[2]))
            # ... and the same on every rank (an all-reduce of a checksum of the final state)
            t = torch.from_numpy(np.ascontiguousarray(got[0][-1])).to(f'cuda:{local}' if args.backend == 'nccl' else 'cpu')
            lo, hi = t.clone(), t.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            rec['loops'][loop] = {'path': got[3], 'equals_fused_chain': same, 'identical_on_every_rank': bool(torch.equal(lo, hi))}
        verdict['cases'].append(rec)
        ctx.close()
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, f'rank{rank}.json'), 'w') as fh:
        json.dump(verdict, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
