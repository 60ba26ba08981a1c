#!/usr/bin/env python3
"""Headline benchmark: log-prob evals/sec (walker-steps/s), PolynomialDecomposition,
32 frequencies, fp64 (BASELINE.json `metric`).

A "step" is one pass of the hot path -- prior + forward model + Gaussian
log-likelihood for every walker of the rank's batch -- i.e. ONE launch of the
log-probability kernel over theta (W, 7) already resident in HBM.  Walkers shard
across ranks with no data-path collective, so scaling is weak:
each GPU gets the same W.

    python bench.py                                   # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
N_FREQ = 32
POLY_DEG = 5
C_EXP = 1.0


def make_problem(n_freq=N_FREQ, poly_deg=POLY_DEG, c_exp=C_EXP, spectrum_index=0):
    """Synthetic spectrum of SURVEY.md §8(d) + the PolynomialDecomposition precompute."""
    from bisip_amd.synthetic import synthetic_columns
    from bisip_amd.utils import columns_to_data
    data = columns_to_data(synthetic_columns(n_freq, spectrum_index), 'mrad')
    period = np.log10(1. / data['w'])
    log_tau = np.linspace(np.floor(period.min() - 1), np.floor(period.max() + 1), 2 * n_freq)
    log_taus = np.array([log_tau ** i for i in range(poly_deg + 1)])
    taus = 10 ** log_tau
    bounds = np.array([[0.9] + [-1.0] * (poly_deg + 1), [1.1] + [1.0] * (poly_deg + 1)])
    return data, taus, log_taus, bounds


def time_launches(ctx, theta_t, out_t, steps, warmup, torch, dist=None):
    """Times `steps` launches; returns (wall seconds bracketed by barrier+sync, mean
    kernel duration in ms from HIP events on the launch stream)."""
    W = theta_t.shape[0]
    stream = torch.cuda.current_stream()
    for _ in range(warmup):
        ctx.logprob_dev(theta_t.data_ptr(), W, out_t.data_ptr(), stream.cuda_stream)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)            # HIP events on the stream the kernel is launched on
    for _ in range(steps):
        ctx.logprob_dev(theta_t.data_ptr(), W, out_t.data_ptr(), stream.cuda_stream)
    ev1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / steps   # back-to-back launches: mean launch duration
    return wall, kern_ms


def cpu_baseline(data, taus, log_taus, bounds, theta, gpu_logp):
    """Oracle (CPU restatement of the reference loop, libm cpow) on a bounded sample of
    the same workload, on this box's host cores.  Also the parity spot-check."""
    import oracle
    prob = oracle.OracleProblem('PolynomialDecomposition', data['w'], data['zn'], data['zn_err'],
                                bounds, taus=taus, log_taus=log_taus, c_exp=C_EXP)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 16, oracle.max_threads()))
    # calibrate on 1 core, then size the threaded sample for ~20 s of CPU work
    n1 = min(4096, theta.shape[0])
    t0 = time.perf_counter()
    ref1 = oracle.logprob(prob, theta[:n1], n_threads=1)
    t1 = time.perf_counter() - t0
    rate1 = n1 / t1
    n = int(min(theta.shape[0], max(n1, rate1 * 20.0)))
    t0 = time.perf_counter()
    ref = oracle.logprob(prob, theta[:n], n_threads=threads)
    tn = time.perf_counter() - t0
    fin = np.isfinite(ref)
    err = float(np.max(np.abs(gpu_logp[:n][fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin]))))
    same_inf = bool(np.array_equal(np.isneginf(gpu_logp[:n]), np.isneginf(ref)))
    return {
        'value': n / tn, 'unit': 'evals/s', 'cores': threads, 'kind': 'port',
        'sample': f'first {n} walkers of the same theta batch, oracle/bisip_oracle.c '
                  f'(loop-faithful, glibc cpow) with {threads} OpenMP threads',
        'value_1core': rate1, 'sample_1core': f'first {n1} walkers, 1 thread',
        'host_cpus_visible': cores,
    }, err, same_inf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--walkers', type=int, default=1 << 24, help='walkers per GPU per step')
    ap.add_argument('--variant', default='auto')
    ap.add_argument('--prime-seconds', type=float, default=0.5,
                    help='untimed device warm-up before the W warm-up steps')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='process-group backend; gloo + --same-device rehearses the multi-rank '
                         'control flow on a one-GPU box (RCCL refuses two ranks on one device)')
    ap.add_argument('--same-device', action='store_true', help='all ranks use cuda:0 (rehearsal only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-variants', action='store_true')
    args = ap.parse_args()

    # read by the HSA runtime when the GPU is first touched: set before anything initialises it
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if 'RANK' in os.environ:   # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f'warning: --gpus {args.gpus} but WORLD_SIZE {world}', file=sys.stderr)

    from bisip_amd import _hip
    from bisip_amd.synthetic import synthetic_theta
    from bisip_amd.utils import respect_cpu_quota
    # host thread pools sized by the node's core count would be throttled by the container's
    # CPU quota at some random moment, possibly inside the timed region
    respect_cpu_quota()

    data, taus, log_taus, bounds = make_problem()
    ctx = _hip.HipContext(_hip.MODEL_POLYDECOMP, data['w'], data['zn'], data['zn_err'], bounds,
                          device=local_rank, poly_deg=POLY_DEG, c_exp=C_EXP, taus=taus,
                          log_taus=log_taus, variant=args.variant)
    ndim = POLY_DEG + 2
    W = int(args.walkers)
    # walker positions: uniform in the prior box (100 % in-prior: no early-exit savings)
    theta = synthetic_theta(bounds[0], bounds[1], W, seed=2024 + rank)
    theta_t = torch.from_numpy(theta).to(f'cuda:{local_rank}')
    out_t = torch.empty(W, dtype=torch.float64, device=f'cuda:{local_rank}')

    # Device warm-up (setup, untimed): the first ~0.1 s of launches after an idle period run
    # 5-8 % slower (fabric/memory clocks ramping, first touch of a 1 GB buffer); prime the
    # device so that the W warm-up steps and the K timed steps see steady state.
    prime_t0 = time.perf_counter()
    while time.perf_counter() - prime_t0 < args.prime_seconds:
        for _ in range(50):
            ctx.logprob_dev(theta_t.data_ptr(), W, out_t.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()

    wall, kern_ms = time_launches(ctx, theta_t, out_t, args.steps, args.warmup, torch, dist)
    if dist is not None:
        t = torch.tensor([wall, kern_ms], dtype=torch.float64,
                         device=f'cuda:{local_rank}' if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kern_ms = float(t[0]), float(t[1])

    if rank == 0:
        total_evals = float(W) * world * args.steps
        value = total_evals / wall
        bytes_per_eval = 8 * (ndim + 1)            # read one theta row, write one logp
        achieved = bytes_per_eval * W / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath))
                if rec.get('walkers') == W and rec.get('kernel') == ctx.kernel_name:
                    traffic = rec.get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        result = {
            'metric': 'log-prob evals/sec (walker-steps/s), PolynomialDecomposition 32 freq fp64',
            'value': value, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': wall / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'PolynomialDecomposition poly_deg=5 c_exp=1.0, 32 synthetic '
                                   'frequencies (S=64 taus), ndim 7, theta uniform in the prior box',
                       'walkers_per_gpu': W, 'global_walkers': W * world,
                       'parallelism': f'walker-sharded x{world}, no data-path collective',
                       'kernel': ctx.kernel_name, 'variant': ctx.variant},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'bytes_per_eval': bytes_per_eval, 'kernel_ms': kern_ms,
                         'traffic_source': 'profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / '
                                           'WRITE_SIZE passes of this command (FETCH_SIZE x2, gfx950)'
                                           if traffic is not None else None},
        }
        if world == 1:
            gpu_logp = out_t.cpu().numpy()
            if not args.no_variants:
                variants = {}
                for v in ('reduced', 'collapsed', 'faithful', 'wave'):
                    ctx.set_variant(v)
                    k = max(3, min(args.steps, 10))
                    _, ms = time_launches(ctx, theta_t, out_t, k, 2, torch)
                    variants[v] = {'evals_per_s': W / (ms * 1e-3), 'kernel_ms': ms,
                                   'kernel': ctx.kernel_name}
                ctx.set_variant(args.variant)
                result['variants'] = variants
            if not args.no_cpu_baseline:
                cb, err, same_inf = cpu_baseline(data, taus, log_taus, bounds, theta, gpu_logp)
                result['cpu_baseline'] = cb
                result['parity'] = {'max_rel_err_vs_oracle': err, 'neg_inf_rows_match': same_inf,
                                    'tolerance': 1e-10}
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
