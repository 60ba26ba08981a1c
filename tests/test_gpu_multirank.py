"""The multi-rank paths on real GPUs.

RCCL has, on this pool, only ever seen one rank (the pool hands out one GPU per box and RCCL refuses two ranks
on one device): the tests that need two GPUs are here and SKIP themselves -- decided from the KFD topology in
sysfs, before anything touches a device -- so that they run the day a box has two.  What a one-GPU box can run
runs: the same worker with every rank on cuda:0 over gloo (the Python half-step loop, exchange through host
memory, the real kernels), and `bench.py` with four ranks on one device (with the test runner and the launcher --
torch.distributed.run opens the device too -- that is the six processes the pool allows on a card; eight ranks
are rehearsed on the CPU in tests/test_dist.py).

What the sharded sampler replaces: fit(pool=...) -> emcee.EnsembleSampler(pool=pool),
/root/reference/src/bisip/models.py:84,91-94,115."""

import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402  (gpus_enumerated reads sysfs only)

pytestmark = pytest.mark.gpu

N_GPUS = bench.gpus_enumerated() or 0
needs_two = pytest.mark.skipif(N_GPUS < 2, reason=f'needs 2 GPUs on this box for RCCL ranks; the KFD topology lists {N_GPUS}')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_worker(world, outdir, *flags):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'tests', 'multirank_worker.py'), '--out', str(outdir), *flags]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=bench.rank_group_env())
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    return [json.load(open(os.path.join(outdir, f'rank{i}.json'))) for i in range(world)]


def _check(verdicts, world, loops):
    assert len(verdicts) == world
    for v in verdicts:
        assert v['world'] == world and len(v['cases']) == 4
        for case in v['cases']:
            assert case['fused_path'] == 'launch-per-half-step'
            assert sorted(case['loops']) == sorted(loops), case
            for loop, res in case['loops'].items():
                assert res['path'] == ('sharded' if loop == 'python' else 'sharded-rccl'), (case['walkers'], loop, res)
                assert res['equals_fused_chain'] and res['identical_on_every_rank'], (v['rank'], case['walkers'], loop, res)
    # the shapes that matter: an odd ensemble, a half with fewer slots than ranks, BASELINE config 4
    walkers = [c['walkers'] for c in verdicts[0]['cases']]
    assert walkers[0] % 2 == 1 and walkers[1] // 2 < world and walkers[3] == 32768


@needs_two
def test_rccl_two_gpus_sharded_chains_equal_the_fused_chain(tmp_path):
    """Two ranks over RCCL (backend 'nccl'), one GPU each: the Python loop, the C loop on a communicator of its
    own and the C loop on torch.distributed's communicator -- the in-place ncclAllGather of comm_rccl.hip with a
    second rank at last -- give the fused single-GPU chain on both ranks, for an odd ensemble, for a half with
    fewer slots than ranks and for 32,768 walkers."""
    _check(_run_worker(2, tmp_path), 2, ['python', 'rccl-own', 'rccl'])


@needs_two
def test_bench_two_gpus_over_rccl():
    """`python bench.py --gpus 2`: two ranks seen, each checked against the oracle, and the extras' three
    drivers of config 4 (python / rccl-own / rccl) all equal to the single-GPU chain."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--walkers', str(1 << 22),
                        '--steps', '5', '--warmup', '2'], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert rec['n_gpus'] == 2 and rec['ranks_seen'] == 2 and rec['config']['backend'] == 'nccl'
    assert rec['parity']['ranks_checked'] == 2 and rec['parity']['max_rel_err_vs_oracle'] <= 1e-10
    cfg4 = [json.loads(ln)['sampler_cfg4'] for ln in r.stderr.splitlines() if ln.startswith('{"sampler_cfg4"')]
    assert sorted(c['sharded_loop'] for c in cfg4) == ['python', 'rccl', 'rccl-own'], r.stderr[-3000:]
    for c in cfg4:
        assert c['n_gpus'] == 2 and c['state_identical_on_every_rank'] and c['equals_single_gpu_fused_chain'], c


def test_three_ranks_on_one_device_sharded_chains_equal_the_fused_chain(tmp_path):
    """The same worker on what this pool has: three ranks on cuda:0 over gloo -- the Python half-step loop with
    the real eval / apply kernels, uneven shards, a rank with no slot at all -- against the fused chain."""
    _check(_run_worker(3, tmp_path, '--backend', 'gloo', '--same-device'), 3, ['python'])


def test_bench_four_ranks_on_one_device():
    """`bench.py --gpus 4 --backend gloo --same-device`: the whole multi-rank control flow of the scaling run
    (self-launch, rendezvous, barriers, the all-reduced timing, a parity check per rank, the extras in a second
    group of ranks) with the real kernels and more than two ranks -- four, because the pool allows six
    processes on a card and this test runner and the launcher are two of them (five ranks: the run is killed)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--backend', 'gloo',
                        '--same-device', '--walkers', str(1 << 20), '--steps', '4', '--warmup', '1',
                        '--prime-seconds', '0.05'], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 4 and rec['ranks_seen'] == 4 and rec['scaling'] == 'weak'
    assert rec['config']['global_walkers'] == 4 << 20 and len(rec['roofline']['per_rank_kernel_ms']) == 4
    # both readings of "4 GPUs" are in the line: weak (every rank its own 2^20 rows: the value) and strong (2^20 rows
    # in all, contiguous blocks of shard_range)
    assert rec['weak']['value'] == rec['value'] and rec['weak']['global_walkers'] == 4 << 20
    assert rec['strong']['global_walkers'] == 1 << 20 and rec['strong']['walkers_per_gpu'] == [1 << 18] * 4
    assert len(rec['strong']['per_rank_kernel_ms']) == 4 and rec['strong']['value'] > 0
    assert rec['parity']['ranks_checked'] == 4 and rec['parity']['neg_inf_rows_match'] is True
    assert rec['parity']['max_rel_err_vs_oracle'] <= 1e-10
    cfg4 = [json.loads(ln)['sampler_cfg4'] for ln in r.stderr.splitlines() if ln.startswith('{"sampler_cfg4"')]
    assert len(cfg4) == 1 and cfg4[0]['n_gpus'] == 4 and cfg4[0]['sharded_loop'] == 'python', r.stderr[-3000:]
    assert cfg4[0]['state_identical_on_every_rank'] is True and cfg4[0]['equals_single_gpu_fused_chain'] is True
