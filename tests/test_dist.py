"""Multi-rank paths on CPU: gloo, world_size 2.  The log-probability evaluator is the
CPU oracle here (the HIP kernels need a GPU); what is under test is the sharding, the
per-half-step all-gather and that a sharded run reproduces the single-rank chain."""

import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_cases


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, outdir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle
    from bisip_amd.dist import ShardedLogProb, all_gather_rows, shard_range
    from bisip_amd.sampler import EnsembleSampler

    g = np.load(case)
    prob = oracle.OracleProblem.from_golden(g, 'PeltonColeCole')
    fn = lambda t: oracle.logprob(prob, t)  # noqa: E731

    # (1) sharded log-prob: contiguous blocks, outputs in rank order
    theta = g['theta'][:37]
    sharded = ShardedLogProb(fn)
    full = sharded(theta)
    a, b = shard_range(len(theta), world, rank)
    local = sharded.local(theta)
    # (2) uneven all-gather
    blk = np.full((b - a, 3), float(rank)) + np.arange(a, b)[:, None]
    gathered = all_gather_rows(blk, len(theta))
    # (3) sharded sampler == single-rank sampler, bit for bit
    lo, hi = g['bounds']
    np.random.seed(11)
    p0 = np.random.uniform(lo, hi, (18, lo.size))
    np.random.seed(12)
    s = EnsembleSampler(18, lo.size, fn, distributed=True)
    s.run_mcmc(p0, 25)
    # (4) a survey's per-spectrum summaries, gathered in spectrum order (SpectraBatch.gather;
    #     the batch object itself needs a GPU, its gather does not)
    from bisip_amd.batch import SpectraBatch
    sb = object.__new__(SpectraBatch)
    sb.n_spectra_total = 11
    lo_s, hi_s = shard_range(11, world, rank)
    sb.n_spectra = hi_s - lo_s
    mine = np.arange(lo_s, hi_s, dtype=float)[:, None, None] * 100 + np.arange(6.0).reshape(2, 3)
    survey = sb.gather(mine)
    np.savez(os.path.join(outdir, f'rank{rank}.npz'), full=full, local=local, a=a, b=b,
             gathered=gathered, chain=s.get_chain(), logp=s.get_log_prob(),
             acc=s.acceptance_fraction, survey=survey)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from bisip_amd.dist import shard_range, shard_sizes
    for n in (0, 1, 7, 8, 9, 4096, 32768):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(shard_sizes(n, world)) - min(shard_sizes(n, world)) <= 1


def test_two_rank_gloo(tmp_path):
    import torch.multiprocessing as mp
    import oracle
    from bisip_amd.sampler import EnsembleSampler
    case = [p for p in golden_cases() if 'PeltonColeCole_SIP-K389175' in p][0]
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f'rank{i}.npz') for i in range(world)]

    g = np.load(case)
    prob = oracle.OracleProblem.from_golden(g, 'PeltonColeCole')
    want = oracle.logprob(prob, g['theta'][:37])
    for i in range(world):
        assert np.array_equal(r[i]['full'], want)                       # same on every rank
        assert np.array_equal(r[i]['local'], want[int(r[i]['a']):int(r[i]['b'])])
        assert np.array_equal(r[i]['gathered'][:, 0], np.concatenate(
            [np.arange(int(r[k]['a']), int(r[k]['b'])) + k for k in range(world)]))
    assert int(r[0]['b']) == int(r[1]['a'])

    lo, hi = g['bounds']
    np.random.seed(11)
    p0 = np.random.uniform(lo, hi, (18, lo.size))
    np.random.seed(12)
    s = EnsembleSampler(18, lo.size, lambda t: oracle.logprob(prob, t))
    s.run_mcmc(p0, 25)
    for i in range(world):
        assert np.array_equal(r[i]['chain'], s.get_chain())
        assert np.array_equal(r[i]['logp'], s.get_log_prob())
        assert np.array_equal(r[i]['acc'], s.acceptance_fraction)
        # every rank ends with the whole survey's summaries, spectrum e in row e
        assert np.array_equal(r[i]['survey'], np.arange(11.0)[:, None, None] * 100 + np.arange(6.0).reshape(2, 3))


def _device_driver_worker(rank, world, port, case, outdir, nwalkers=19):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle
    from bisip_amd.sampler import DeviceEnsembleSampler
    from numpy_stretch_backend import NumpyStretchBackend
    g = np.load(case)
    prob = oracle.OracleProblem.from_golden(g, 'PeltonColeCole')
    fn = lambda t: oracle.logprob(prob, t)  # noqa: E731
    lo, hi = g['bounds']
    np.random.seed(11)
    p0 = np.random.uniform(lo, hi, (nwalkers, lo.size))   # odd ensemble: uneven halves and shards
    np.random.seed(12)
    s = DeviceEnsembleSampler(nwalkers, lo.size, backend=NumpyStretchBackend(fn), distributed=True, chunk=6)
    s.run_mcmc(p0, 20)
    np.savez(os.path.join(outdir, f'dev_rank{rank}.npz'), chain=s.get_chain(), logp=s.get_log_prob(),
             acc=s.acceptance_fraction)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,nwalkers', [(2, 19), (8, 15)])
def test_sharded_device_driver_two_rank_gloo(tmp_path, world, nwalkers):
    """DeviceEnsembleSampler's multi-rank path (eval -> all_gather_into_tensor -> apply)
    over gloo with a NumPy stand-in for the kernels: every rank ends with the chain of
    the single-rank host sampler.  Two ranks, and the eight of the scaling run with an ensemble whose halves
    (8 and 7 slots) leave the last rank of the second half with nothing to evaluate."""
    import torch.multiprocessing as mp
    import oracle
    from bisip_amd.sampler import EnsembleSampler
    case = [p for p in golden_cases() if 'PeltonColeCole_SIP-K389175' in p][0]
    mp.spawn(_device_driver_worker, args=(world, _free_port(), case, str(tmp_path), nwalkers), nprocs=world,
             join=True)
    g = np.load(case)
    prob = oracle.OracleProblem.from_golden(g, 'PeltonColeCole')
    lo, hi = g['bounds']
    np.random.seed(11)
    p0 = np.random.uniform(lo, hi, (nwalkers, lo.size))
    np.random.seed(12)
    s = EnsembleSampler(nwalkers, lo.size, lambda t: oracle.logprob(prob, t))
    s.run_mcmc(p0, 20)
    for i in range(world):
        r = np.load(tmp_path / f'dev_rank{i}.npz')
        assert np.array_equal(r['chain'], s.get_chain())
        assert np.array_equal(r['logp'], s.get_log_prob())
        assert np.array_equal(r['acc'], s.acceptance_fraction)


def test_batch_replica_sharding():
    """Batch of spectra shards as whole replicas: the rank blocks tile the spectrum list."""
    from bisip_amd.dist import shard_range
    n = 4096
    blocks = [shard_range(n, 8, r) for r in range(8)]
    assert blocks[0] == (0, 512) and blocks[-1] == (3584, 4096)
    assert all(b - a == 512 for a, b in blocks)


def _run_bench(*flags, env=None, launcher=()):
    import subprocess
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, *launcher, os.path.join(ROOT, 'bench.py'), *flags], capture_output=True,
                          text=True, timeout=600, cwd=ROOT, env=e)


def _result_lines(r):
    return [ln for ln in r.stdout.splitlines() if ln.strip()]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no RANK in the environment must itself become a 2-rank run
    (the parent starts torch.distributed.run before touching any GPU) and say so: n_gpus 2,
    ranks_seen 2 from an all-reduce over the ranks.  --rehearse-cpu runs the whole control flow
    (launch, rendezvous, barriers, reductions, the one JSON line) with no kernel -- value is null.
    The extras (cfg4 / cfg5) run AFTER the line, in a second group of ranks: other processes."""
    import json
    r = _run_bench('--gpus', '2', '--backend', 'gloo', '--rehearse-cpu', '--steps', '3', '--warmup', '1')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _result_lines(r)
    assert len(lines) == 1                      # rank 0's line only; everything else went to stderr
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['ranks_seen'] == 2
    assert rec['value'] is None and 'not a measurement' in rec['rehearsal']
    assert rec['steps'] == 3 and rec['warmup'] == 1 and rec['scaling'] == 'weak'
    extras = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith('{"extras_rehearsal"')]
    assert len(extras) == 1, r.stderr[-2000:]
    assert extras[0]['extras_rehearsal']['ranks'] == 2
    assert extras[0]['extras_rehearsal']['pid'] != rec['rank0_pid']      # not the headline's processes


def test_bench_headline_survives_a_crash_in_the_extras():
    """A rank of the extras group aborts (SIGABRT, as a failed assertion inside a native library
    would): the run still exits 0 with its one result line, and says what happened on stderr."""
    import json
    r = _run_bench('--gpus', '2', '--backend', 'gloo', '--rehearse-cpu', '--steps', '2', '--warmup', '1',
                   env={'BISIP_BENCH_INJECT_EXTRAS_ABORT': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _result_lines(r)
    assert len(lines) == 1 and json.loads(lines[0])['ranks_seen'] == 2
    assert 'the extras group' in r.stderr and 'not affected' in r.stderr
    assert not [ln for ln in r.stderr.splitlines() if ln.startswith('{"extras_rehearsal"')]


@pytest.mark.parametrize('crash', [False, True])
def test_bench_under_the_drivers_own_launcher(crash):
    """The driver starts the ranks itself (`python -m torch.distributed.run ... bench.py --gpus 2`):
    there is no parent of ours, so rank 0 -- after the line is out and the process group is gone --
    starts the extras as a fresh child process and exits 0 whatever becomes of it."""
    import json
    launcher = ('-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                '--master-port', str(_free_port()))
    r = _run_bench('--gpus', '2', '--backend', 'gloo', '--rehearse-cpu', '--steps', '2', '--warmup', '1',
                   launcher=launcher, env={'BISIP_BENCH_INJECT_EXTRAS_ABORT': '1'} if crash else {'OMP_NUM_THREADS': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in _result_lines(r) if ln.startswith('{"metric"')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['ranks_seen'] == 2
    extras = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith('{"extras_rehearsal"')]
    if crash:
        assert not extras and 'not affected' in r.stderr
    else:
        assert len(extras) == 1 and extras[0]['extras_rehearsal']['pid'] != rec['rank0_pid']


def test_bench_refuses_more_gpus_than_enumerated(tmp_path, monkeypatch):
    """The parent counts GPUs from the KFD topology in sysfs (no GPU call) and refuses an
    impossible --gpus with a clear message."""
    sys.path.insert(0, ROOT)
    import bench
    n = bench.gpus_enumerated()
    assert n is None or n >= 0
    if n is None:
        return               # no amdgpu driver here: nothing to refuse against (the GPU box has one)
    r = _run_bench('--gpus', str(n + 1), '--no-extras')
    assert r.returncode == 2 and 'enumerates' in r.stderr


def test_bench_fails_when_a_rank_fails():
    """A failed child is a failed run: non-zero exit, no result line, no retry."""
    r = _run_bench('--gpus', '2', '--backend', 'nccl', '--rehearse-cpu')   # every rank rejects this combination
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert 'failed with status' in r.stderr

