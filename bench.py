#!/usr/bin/env python3
"""Headline benchmark: log-prob evals/sec (walker-steps/s), PolynomialDecomposition,
32 frequencies, fp64 (BASELINE.json `metric`).

A "step" is one pass of the hot path -- prior + forward model + Gaussian
log-likelihood for every walker of the rank's batch -- i.e. ONE launch of the
log-probability kernel over theta (W, 7) already resident in HBM.  Walkers shard
across ranks with no data-path collective, so scaling is weak:
each GPU gets the same W.

    python bench.py                       # 1 GPU
    python bench.py --gpus N              # N GPUs: this process starts the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no RANK in the environment the parent -- before it imports torch or
touches a GPU -- starts `python -m torch.distributed.run ... bench.py <same flags> --no-extras`
as a child process, relays rank 0's JSON line and exits with that group's status (a failed rank
is a failed run; nothing is retried).  Prints ONE JSON line on rank 0.

The sampler-mode measurements of BASELINE configs 4 and 5 ("extras", stderr only) never share a
process with the headline: they run in a SECOND group of ranks (`--extras-only`), started as a
fresh child process only after the headline line is out, under a wall-clock limit (180 s), and their
exit status is ignored -- a crash, abort or hang there cannot cost the measured line.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MAX_CLOCK_HZ = 2.4e9           # same guide: max engine clock
N_SIMD = 256 * 4               # 256 CUs x 4 SIMDs
# A wave64 fp64 (or fp32) VALU instruction occupies its SIMD's 16-lane pipe for 4 cycles, so the
# chip issues at most N_SIMD * clock / 4 VALU wave-instructions per second.
VALU_PEAK_WAVE_INSTR_S = N_SIMD * MAX_CLOCK_HZ / 4.0
N_FREQ = 32
POLY_DEG = 5
C_EXP = 1.0
METRIC = 'log-prob evals/sec (walker-steps/s), PolynomialDecomposition 32 freq fp64'


def make_problem(n_freq=N_FREQ, poly_deg=POLY_DEG, c_exp=C_EXP, spectrum_index=0):
    """Synthetic spectrum of SURVEY.md §8(d) + the PolynomialDecomposition precompute."""
    import numpy as np
    from bisip_amd.synthetic import synthetic_columns
    from bisip_amd.utils import columns_to_data
    data = columns_to_data(synthetic_columns(n_freq, spectrum_index), 'mrad')
    period = np.log10(1. / data['w'])
    log_tau = np.linspace(np.floor(period.min() - 1), np.floor(period.max() + 1), 2 * n_freq)
    log_taus = np.array([log_tau ** i for i in range(poly_deg + 1)])
    taus = 10 ** log_tau
    bounds = np.array([[0.9] + [-1.0] * (poly_deg + 1), [1.1] + [1.0] * (poly_deg + 1)])
    return data, taus, log_taus, bounds


# The other log-probability kernels of the path at the BASELINE shapes (N = 32 frequencies),
# measured after the headline so that BENCH_r*.json carries an achieved fraction for the
# compute-bound ones too.  label -> (model id name, context kwargs, bounds)
def kernel_zoo():
    import numpy as np
    b = {
        'colecole1': ('COLECOLE', dict(n_modes=1), [[0.9, 0, -15, 0], [1.1, 1, 5, 1]]),
        'colecole2': ('COLECOLE', dict(n_modes=2), [[0.9, 0, 0, -15, -15, 0, 0], [1.1, 1, 1, 5, 5, 1, 1]]),
        'dias2000': ('DIAS2000', {}, [[0.9, 0, -20, 0, 0], [1.1, 1, 0, 150, 1]]),
        'shin2015': ('SHIN2015', {}, [[0, 0, -15, -7, 0, 0], [1, 1, -13, -5, 1, 1]]),
    }
    return {k: (m, kw, np.array(v, dtype=np.float64)) for k, (m, kw, v) in b.items()}


def load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', name)))
    except (OSError, ValueError):
        return None


# The PMC counters this file replays (profiles/pmc_traffic.json, valu_counts.json) were collected from a
# particular build of the log-probability kernels: the files carry the SHA-256 of these sources as they were
# on the GPU box, and a line printed from other sources reports the counters as stale instead of quoting them.
# (everything that decides WHICH kernel runs or what it compiles to: the kernels, their dispatch, the host's
# loop / tier choices and constants, the compiler flags)
KERNEL_SOURCES = ('bisip_amd/csrc/kernels.h', 'bisip_amd/csrc/sampler_kernels.h', 'bisip_amd/csrc/dispatch_logprob.hip',
                  'bisip_amd/csrc/bisip_hip.hip', 'bisip_amd/csrc/host.h', 'bisip_amd/csrc/philox.h',
                  'bisip_amd/csrc/stretch_launch.h', 'bisip_amd/csrc/Makefile')


def kernel_sources_sha256():
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def load_counters(name):
    """A counter file of profiles/ if it was collected from the kernel sources of this tree, else None;
    second value: True when the file exists but belongs to other sources."""
    rec = load_json(name)
    if rec is None:
        return None, False
    if rec.get('kernel_sources_sha256') != kernel_sources_sha256():
        return None, True
    return rec, False


def valu_roofline(label, W, kernel_ms, counts):
    """fp64 VALU-issue roofline of one launch from the PMC-measured instruction count
    (profiles/valu_counts.json, written by benchmarks/summarize_pmc.py from a
    `rocprofv3 --pmc SQ_INSTS_VALU` pass of `bench.py --pmc-pass`)."""
    rec = (counts or {}).get(label)
    if not rec:
        return None
    per_eval = float(rec['valu_wave_instr_per_eval'])
    achieved = per_eval * W / (kernel_ms * 1e-3)
    out = {'bound': 'fp64-valu-issue', 'achieved': achieved, 'peak': VALU_PEAK_WAVE_INSTR_S,
           'unit': 'VALU wave-instr/s', 'frac': achieved / VALU_PEAK_WAVE_INSTR_S,
           'valu_wave_instr_per_eval': per_eval, 'kernel': rec.get('kernel'),
           'source': 'profiles/valu_counts.json (SQ_INSTS_VALU per launch / walkers); peak = 1024 SIMDs x 2.4 GHz / 4 cycles'}
    if 'issue_slots_per_eval' in rec:
        # the same ceiling with every instruction at its own price: a quarter-rate fp64 transcendental (v_rcp_f64, ...;
        # SQ_INSTS_VALU_TRANS_F64) occupies the issue port for four slots
        slots = float(rec['issue_slots_per_eval'])
        out.update({'trans_f64_wave_instr_per_eval': float(rec['trans_f64_wave_instr_per_eval']),
                    'issue_slots_per_eval': slots,
                    'frac_issue_slots': slots * W / (kernel_ms * 1e-3) / VALU_PEAK_WAVE_INSTR_S})
        if 'wave_cycle_fractions' in rec:
            out['wave_cycle_fractions'] = rec['wave_cycle_fractions']
    return out


def prime(fn, seconds, torch):
    """Untimed back-to-back launches: the first ~0.1 s after an idle period run 5-25 % slower
    (clock / fabric ramp, first touch of the buffers)."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()


def time_launches(fn, steps, warmup, torch, stream, dist=None):
    """Times `steps` calls of fn (one kernel launch each); returns (wall seconds bracketed by
    barrier+sync, mean kernel duration in ms from HIP events on the launch stream)."""
    for _ in range(warmup):
        fn()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)            # HIP events on the stream the kernel is launched on
    for _ in range(steps):
        fn()
    ev1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / steps   # back-to-back launches: mean launch duration
    return wall, kern_ms


def engine_clock_ghz(fn, kernel_ms, torch, launches=40):
    """The engine clock the chip holds while `fn`'s kernel runs: a one-wave probe
    (bisip_clock_probe_dev) on a side stream brackets a window inside a back-to-back series of
    `fn` launches with reads of the shader clock and of the constant 100 MHz counter.  Untimed;
    the timed blocks never have a probe beside them."""
    from bisip_amd import _hip
    side = torch.cuda.Stream()
    buf = torch.zeros(4, dtype=torch.int64, device='cuda')
    window_us = min(max(kernel_ms * 1e3 * launches / 4.0, 50.0), 5000.0)
    torch.cuda.synchronize()
    for _ in range(launches // 4):
        fn()
    _hip.clock_probe_dev(buf.data_ptr(), window_us, side.cuda_stream)
    for _ in range(launches - launches // 4):
        fn()
    torch.cuda.synchronize()
    t0, t1, r0, r1 = (int(x) for x in buf.cpu())
    if r1 <= r0:
        return None
    return (t1 - t0) / (r1 - r0) * 0.1


def fp64_stream_ceiling(torch, stream, seconds=0.3, rounds=400):
    """What a stream of independent fp64 FMAs reaches on this chip, measured the way the kernels are (primed, ten
    back-to-back launches of ~0.2 ms, HIP events on the launch stream; bisip_fp64_stream_probe_dev: 8 waves per
    SIMD on every compute unit, operands with full mantissas): the ceiling the compute-bound kernels are held to.
    The nominal issue peak (1024 SIMDs x 2.4 GHz / 4 cycles) is not reachable: under fp64-dense load the chip
    holds 2.2-2.3 GHz and a wave-instruction takes 4.25-4.45 cycles (benchmarks/micro/fp64_stream_ceiling.hip)."""
    from bisip_amd import _hip
    buf = torch.empty(_hip.fp64_stream_probe_lanes(), dtype=torch.float64, device='cuda')
    n = [0]

    def fn():
        n[0] = _hip.fp64_stream_probe_dev(buf.data_ptr(), rounds, stream.cuda_stream)
    prime(fn, seconds, torch)
    _, ms = time_launches(fn, 10, 2, torch, stream)
    rate = n[0] / (ms * 1e-3)
    return {'wave_instr_per_s': rate, 'frac_of_nominal_peak': rate / VALU_PEAK_WAVE_INSTR_S, 'kernel_ms': ms,
            'clock_ghz': engine_clock_ghz(fn, ms, torch),
            'what': 'independent v_fma_f64 (d = d * s + v), 8 waves per SIMD on 256 CUs, full-mantissa operands, '
                    f'{32 * rounds} per wave and launch'}


def against_stream(rv, stream_rec):
    """The kernel's issue-slot rate as a fraction of the measured FMA stream's."""
    if rv and stream_rec and 'frac_issue_slots' in rv:
        rv['frac_of_fma_stream'] = rv['frac_issue_slots'] * VALU_PEAK_WAVE_INSTR_S / stream_rec['wave_instr_per_s']
    return rv


def with_clock(rv, clock_ghz):
    """fp64-issue fraction at the clock the chip actually held (DVFS) beside the one at 2.4 GHz."""
    if rv and clock_ghz:
        rv['clock_ghz'] = clock_ghz
        rv['frac_at_clock'] = rv['achieved'] / (N_SIMD * clock_ghz * 1e9 / 4.0)
        if 'frac_issue_slots' in rv:
            rv['frac_issue_slots_at_clock'] = rv['frac_issue_slots'] * 2.4 / clock_ghz
        rv['cycles_per_valu_instr_per_simd'] = N_SIMD * clock_ghz * 1e9 / rv['achieved']
    return rv


def cpu_baseline(data, taus, log_taus, bounds, theta, gpu_logp):
    """Oracle (CPU restatement of the reference loop, libm cpow) on a bounded sample of
    the same workload, on this box's host cores.  Also the parity spot-check."""
    import numpy as np
    import oracle
    from bisip_amd.utils import cpu_quota
    prob = oracle.OracleProblem('PolynomialDecomposition', data['w'], data['zn'], data['zn_err'],
                                bounds, taus=taus, log_taus=log_taus, c_exp=C_EXP)
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    quota = cpu_quota()       # what the container may actually burn (CFS quota), <= visible
    threads = max(1, min(quota, oracle.max_threads()))
    # calibrate on 1 core, then size the threaded sample for ~20 s of CPU work
    n1 = min(4096, theta.shape[0])
    t0 = time.perf_counter()
    oracle.logprob(prob, theta[:n1], n_threads=1)
    rate1 = n1 / (time.perf_counter() - t0)
    n = int(min(theta.shape[0], max(n1, rate1 * 20.0)))
    t0 = time.perf_counter()
    ref = oracle.logprob(prob, theta[:n], n_threads=threads)
    tn = time.perf_counter() - t0
    fin = np.isfinite(ref)
    err = float(np.max(np.abs(gpu_logp[:n][fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin]))))
    same_inf = bool(np.array_equal(np.isneginf(gpu_logp[:n]), np.isneginf(ref)))
    out = {
        'value': n / tn, 'unit': 'evals/s', 'cores': threads, 'kind': 'port',
        'sample': f'first {n} walkers of the same theta batch, oracle/bisip_oracle.c '
                  f'(loop-faithful, glibc cpow) with {threads} OpenMP threads',
        'value_1core': rate1, 'sample_1core': f'first {n1} walkers, 1 thread',
        'host_cpus_visible': visible, 'cpu_quota': quota,
    }
    cal = load_json('cpu_calibration.json')
    if cal:
        # the real reference (Cython + NumPy per-walker call) against this same oracle, one core,
        # measured in the build container by benchmarks/cpu_calibration.py
        r = float(cal['reference_over_oracle'])
        out['reference_calibration'] = {
            'reference_over_oracle_1core': r,
            'reference_evals_per_s_1core_there': cal['reference_evals_per_s_1core'],
            'oracle_evals_per_s_1core_there': cal['oracle_evals_per_s_1core'],
            'estimated_reference_1core_here': rate1 * r,
            'estimated_reference_all_cores_here': (n / tn) * r,
            'source': 'profiles/cpu_calibration.json (build container, ' + cal['machine']['cpu'] + ')',
        }
    return out, err, same_inf


def rank_parity(data, taus, log_taus, bounds, theta, gpu_logp, rows=4096):
    """This rank's first `rows` log-probabilities against the oracle: (max relative error over
    finite rows, 1.0 if the -inf rows agree else 0.0).  The checker of the N > 1 line."""
    import numpy as np
    import oracle
    prob = oracle.OracleProblem('PolynomialDecomposition', data['w'], data['zn'], data['zn_err'],
                                bounds, taus=taus, log_taus=log_taus, c_exp=C_EXP)
    n = min(rows, theta.shape[0])
    ref = oracle.logprob(prob, theta[:n], n_threads=1)
    got = np.asarray(gpu_logp[:n])
    fin = np.isfinite(ref)
    same_inf = np.array_equal(np.isneginf(got), np.isneginf(ref)) and not np.isnan(got).any()
    err = float(np.max(np.abs(got[fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin])))) if fin.any() else 0.0
    return err, float(same_inf), n


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


EXTRAS_TIMEOUT_S = 180.0      # the whole second group: ~20 s of start-up, ~20 s of sampling when all is well
_ELASTIC_VARS = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'ROLE_RANK', 'ROLE_NAME',
                 'ROLE_WORLD_SIZE', 'GROUP_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID',
                 'TORCHELASTIC_RESTART_COUNT', 'TORCHELASTIC_MAX_RESTARTS', 'TORCHELASTIC_USE_AGENT_STORE',
                 'TORCHELASTIC_ERROR_FILE', 'TORCH_NCCL_ASYNC_ERROR_HANDLING', 'NCCL_ASYNC_ERROR_HANDLING',
                 'OMP_NUM_THREADS')


def gpus_enumerated():
    """GPUs the kernel driver lists (KFD topology nodes with SIMDs), read from sysfs so that the
    parent of a multi-GPU run can refuse an impossible --gpus without touching a GPU.  None when
    the topology is not there to read (no amdgpu driver: the CPU rehearsal)."""
    import glob
    nodes = glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties')
    if not nodes:
        return None
    n = 0
    for path in nodes:
        try:
            props = dict(ln.split()[:2] for ln in open(path) if len(ln.split()) >= 2)
        except OSError:       # a node this user may not read: not one of ours
            continue
        n += int(props.get('simd_count', '0')) > 0
    return n


def rank_group_env():
    """Environment of a fresh group of ranks: nothing inherited from a rendezvous this process
    may itself be part of.  HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of these machines only
    supports dmabuf IPC; with the legacy mode RCCL's intra-node P2P set-up (and any sharing of
    device memory between processes) fails with `hipIpcGetMemHandle: invalid argument`.  It is
    read when the HSA runtime starts, so it has to be in the environment of the ranks."""
    env = {k: v for k, v in os.environ.items() if k not in _ELASTIC_VARS}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env['OMP_NUM_THREADS'] = os.environ.get('OMP_NUM_THREADS', '1')   # torchrun would set it anyway, with a warning
    return env


def launch_group(n, argv, timeout=None):
    """One group of n ranks of this file under torch.distributed.run, as a child process in its
    own session; returns (status, stdout).  At the limit the group -- exactly the process group
    started here -- is ended and the status is -9."""
    import signal
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
           os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=rank_group_env(), start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGTERM)
            out, _ = proc.communicate(timeout=15)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            out, _ = proc.communicate()
        except ProcessLookupError:
            out, _ = proc.communicate()
        return -9, out
    return proc.returncode, out


def run_extras_group(n, argv):
    """The cfg4 / cfg5 sampler measurements in their own group of ranks.  Whatever happens to that
    group -- non-zero exit, abort, hang -- is reported on stderr and goes no further."""
    argv = [a for a in argv if a not in ('--no-extras', '--extras-only')] + ['--extras-only']
    try:
        rc, out = launch_group(n, argv, timeout=EXTRAS_TIMEOUT_S)
    except Exception as ex:      # noqa: BLE001 -- nothing here may change the run's status
        print(f'bench.py: the extras group could not be started: {type(ex).__name__}: {ex}', file=sys.stderr)
        return
    for ln in out.splitlines():
        print(ln, file=sys.stderr)
    if rc != 0:
        what = f'ended at its {EXTRAS_TIMEOUT_S:.0f} s limit' if rc == -9 else f'ended with status {rc}'
        print(f'bench.py: the extras group (cfg4 / cfg5 sampler measurements) {what}; the result line '
              'above is not affected', file=sys.stderr, flush=True)


def self_launch(n, args, argv):
    """Parent of a multi-GPU run: start the ranks, relay rank 0's line, then -- in a second group,
    after the line is out -- the extras; exit with the FIRST group's status.  Nothing here imports
    torch or touches the GPU."""
    have = gpus_enumerated()
    if not args.rehearse_cpu and not args.same_device and have is not None and have < n:
        print(f'bench.py: --gpus {n} but this machine enumerates {have} GPU(s) '
              '(/sys/class/kfd/kfd/topology/nodes/*/properties with simd_count > 0)', file=sys.stderr)
        sys.exit(2)
    if args.extras_only:
        rc, out = launch_group(n, argv, timeout=EXTRAS_TIMEOUT_S)
        for ln in out.splitlines():
            print(ln, file=sys.stderr)
        sys.exit(rc if rc >= 0 else 1)
    rc, out = launch_group(n, [a for a in argv if a != '--no-extras'] + ['--no-extras'])
    lines = out.splitlines()
    result = [ln for ln in lines if ln.startswith('{"metric"')]
    for ln in lines:
        if not ln.startswith('{"metric"'):
            print(ln, file=sys.stderr)
    if rc != 0:
        print(f'bench.py: the {n}-rank run failed with status {rc}', file=sys.stderr)
        sys.exit(rc if rc > 0 else 1)
    if len(result) != 1:
        print(f'bench.py: expected one result line from rank 0, got {len(result)}', file=sys.stderr)
        sys.exit(1)
    print(result[0], flush=True)
    if not args.no_extras:
        run_extras_group(n, argv)
    sys.exit(0)


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
    ap.add_argument('--rehearse-cpu', action='store_true',
                    help='control flow only (launch, rendezvous, barriers, reductions, the JSON line) with '
                         'NO kernel: value is null.  For the CPU test of the multi-rank path; never a measurement')
    ap.add_argument('--no-extras', '--no-sampler-extra', dest='no_extras', action='store_true',
                    help='N > 1: skip the cfg4 / cfg5 sampler measurements that follow the result line '
                         '(they run in a second group of ranks, never in the processes of the headline)')
    ap.add_argument('--extras-only', action='store_true',
                    help='N > 1: ONLY the cfg4 / cfg5 sampler measurements (what the second group runs); '
                         'lines on stderr, no result line')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help="N > 1: which measurement is the line's `value`.  weak (default, the contract's per-GPU work "
                         "fixed): every rank its own --walkers rows.  strong: --walkers rows IN ALL, split into contiguous "
                         "blocks by shard_range (SURVEY 8e) -- the north star's '>= 6x at 8 GPUs' reading.  Both are "
                         "measured in every N > 1 run; the other one is in the line's `weak` / `strong` object.")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-variants', action='store_true')
    ap.add_argument('--pmc-pass', action='store_true',
                    help='one launch of every kernel this file times, in a fixed order, nothing else: the '
                         'target of `rocprofv3 --pmc SQ_INSTS_VALU` (benchmarks/collect_profiles.sh)')
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    if args.gpus > 1 and 'RANK' not in os.environ:
        self_launch(args.gpus, args, sys.argv[1:])     # does not return
    if args.extras_only and args.gpus == 1:
        ap.error('--extras-only needs --gpus N > 1')

    # read by the HSA runtime when the GPU is first touched: set before anything initialises it
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import numpy as np
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if args.same_device:
        local_rank = 0
    rehearse = args.rehearse_cpu
    if rehearse and args.backend != 'gloo':
        ap.error('--rehearse-cpu needs --backend gloo')
    if not rehearse:
        if not args.same_device and torch.cuda.device_count() < world:     # counting devices does not initialise one
            if rank == 0:
                print(f'bench.py: {world} ranks but torch sees {torch.cuda.device_count()} GPU(s)', file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
    if 'RANK' in os.environ:   # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
    if args.gpus != world:
        # a line that says n_gpus = N must come from N ranks
        if rank == 0:
            print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}', file=sys.stderr)
        sys.exit(2)
    coll_dev = f'cuda:{local_rank}' if (args.backend == 'nccl' and not rehearse) else 'cpu'

    if args.extras_only:
        # the second group of a multi-GPU run: nothing of the headline lives in these processes
        if os.environ.get('BISIP_BENCH_INJECT_EXTRAS_ABORT') and rank == world - 1:
            os.abort()                                 # tests: a rank of the extras dies hard
        if rehearse:
            dist.barrier()
            if rank == 0:
                print(json.dumps({'extras_rehearsal': {'pid': os.getpid(), 'ranks': world}}), file=sys.stderr, flush=True)
        else:
            sampler_extra(args, dist, torch, rank, world, local_rank)
        dist.barrier()
        dist.destroy_process_group()
        return

    ndim = POLY_DEG + 2
    W = int(args.walkers)
    if rehearse:
        wall, kern_ms = 0.0, 0.0
        if dist is not None:
            dist.barrier()
            t0 = time.perf_counter()
            dist.barrier()
            wall = time.perf_counter() - t0
        ctx = None
    else:
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
        # walker positions: uniform in the prior box (100 % in-prior: no early-exit savings)
        theta = synthetic_theta(bounds[0], bounds[1], W, seed=2024 + rank)
        theta_t = torch.from_numpy(theta).to(f'cuda:{local_rank}')
        out_t = torch.empty(W, dtype=torch.float64, device=f'cuda:{local_rank}')
        stream = torch.cuda.current_stream()

        def step():
            ctx.logprob_dev(theta_t.data_ptr(), W, out_t.data_ptr(), stream.cuda_stream)

        if args.pmc_pass:
            pmc_pass(ctx, step, theta_t, out_t, data, torch)
            return
        stalled_blocks = []
        prime(step, args.prime_seconds, torch)
        wall, kern_ms = time_launches(step, args.steps, args.warmup, torch, stream, dist)
        # A process on these boxes now and then waits 10-40 ms for the first device work after a
        # synchronisation although the device-side markers show the kernels ran back to back
        # (DESIGN.md section 3.5c): 20 steps of 0.16 ms would then read 7x slow.  When the host clock
        # of the block is far above the device's, the block -- exactly K steps between barriers, as
        # before -- is timed again (at most twice) and the stalled blocks are reported, not hidden.
        for _ in range(2):
            stalled = wall * 1e3 / max(args.steps, 1) > 1.5 * kern_ms + 0.1
            if dist is not None:
                flag = torch.tensor([float(stalled)], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)      # every rank repeats, or none
                stalled = bool(flag[0] > 0)
            if not stalled:
                break
            stalled_blocks.append(wall / max(args.steps, 1) * 1e3)
            wall, kern_ms = time_launches(step, args.steps, 0, torch, stream, dist)
        # Strong scaling beside it (N > 1): the SAME W rows in all, rank r its contiguous block shard_range(W, N, r)
        # (SURVEY 8e), timed exactly as above -- K launches between barriers, the slowest rank's clock.
        strong = None
        if dist is not None and world > 1:
            from bisip_amd.dist import shard_range
            lo_s, hi_s = shard_range(W, world, rank)
            n_s = hi_s - lo_s

            def step_strong():
                ctx.logprob_dev(theta_t.data_ptr(), n_s, out_t.data_ptr(), stream.cuda_stream)
            prime(step_strong, min(args.prime_seconds, 0.1), torch)
            wall_s, kern_s = time_launches(step_strong, args.steps, args.warmup, torch, stream, dist)
            strong = (wall_s, kern_s, n_s)

    ranks_seen, per_rank_kernel_ms = 1, [kern_ms]
    parity = None
    if dist is not None and world > 1 and not rehearse:
        # every rank checks its own shard (its own theta, seed 2024 + rank): a value summed over
        # N GPUs comes with evidence from each of them
        err, same_inf, n_chk = rank_parity(data, taus, log_taus, bounds, theta, out_t[:4096].cpu().numpy())
        e = torch.tensor([err], dtype=torch.float64, device=coll_dev)
        ok = torch.tensor([same_inf], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        parity = {'max_rel_err_vs_oracle': float(e[0]), 'neg_inf_rows_match': bool(ok[0] > 0), 'tolerance': 1e-10,
                  'rows_per_rank': n_chk, 'ranks_checked': world,
                  'reduction': 'all_reduce MAX of the error, MIN of the -inf agreement, over all ranks'}
    if dist is not None:
        t = torch.tensor([wall, kern_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        one = torch.ones(1, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)     # every rank that ran adds itself
        mine = torch.tensor([kern_ms], dtype=torch.float64, device=coll_dev)
        every = torch.empty(world, dtype=torch.float64, device=coll_dev)
        dist.all_gather_into_tensor(every, mine)
        wall, kern_ms = float(t[0]), float(t[1])       # slowest rank
        ranks_seen = int(round(float(one[0])))
        per_rank_kernel_ms = [float(x) for x in every.cpu()]
        if not rehearse and world > 1:
            ts = torch.tensor([strong[0]], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            mine_s = torch.tensor([strong[1], float(strong[2])], dtype=torch.float64, device=coll_dev)
            every_s = torch.empty(2 * world, dtype=torch.float64, device=coll_dev)
            dist.all_gather_into_tensor(every_s, mine_s)
            every_s = every_s.cpu().view(world, 2)
            strong = {'wall': float(ts[0]), 'per_rank_kernel_ms': [float(x) for x in every_s[:, 0]],
                      'walkers_per_gpu': [int(x) for x in every_s[:, 1]]}

    if rank == 0:
        result = {
            'metric': METRIC, 'value': None, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': None,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic', 'ranks_seen': ranks_seen,
        }
        if rehearse:
            result['rehearsal'] = 'control flow only: no kernel ran, not a measurement'
            result['rank0_pid'] = os.getpid()
            result['config'] = {'workload': 'none (--rehearse-cpu)', 'backend': args.backend}
        else:
            bytes_per_eval = 8 * (ndim + 1)            # read one theta row, write one logp
            achieved = bytes_per_eval * W / (kern_ms * 1e-3) / 1e9    # per GPU, slowest rank
            traffic = None
            rec, traffic_stale = load_counters('pmc_traffic.json')
            if rec and rec.get('walkers') == W and rec.get('kernel') == ctx.kernel_name:
                traffic = rec.get('hbm_bytes_per_launch')
            result.update({
                'value': float(W) * world * args.steps / wall,
                'ms_per_step': wall / args.steps * 1e3,
                'stalled_blocks_ms_per_step': stalled_blocks,     # timed blocks discarded as host-side stalls (rank 0's view)
                'config': {'workload': 'PolynomialDecomposition poly_deg=5 c_exp=1.0, 32 synthetic '
                                       'frequencies (S=64 taus), ndim 7, theta uniform in the prior box',
                           'walkers_per_gpu': W, 'global_walkers': W * world,
                           'parallelism': f'walker-sharded x{world}, no data-path collective',
                           'backend': args.backend if dist is not None else None,
                           'kernel': ctx.kernel_name, 'variant': ctx.variant},
                'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                             'bytes_per_eval': bytes_per_eval, 'kernel_ms': kern_ms,
                             'per': 'GPU (slowest rank)', 'per_rank_kernel_ms': per_rank_kernel_ms,
                             'traffic_source': 'profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / '
                                               'WRITE_SIZE passes of this command (FETCH_SIZE x2, gfx950)'
                                               if traffic is not None else None},
            })
            result['kernel_sources_sha256'] = kernel_sources_sha256()
            if world > 1 and isinstance(strong, dict):
                # both readings of "N GPUs" in one line: `value` is the one --scaling names
                weak_obj = {'value': result['value'], 'ms_per_step': result['ms_per_step'], 'global_walkers': W * world,
                            'walkers_per_gpu': W, 'per_rank_kernel_ms': per_rank_kernel_ms}
                strong_obj = {'value': float(W) * args.steps / strong['wall'], 'ms_per_step': strong['wall'] / args.steps * 1e3,
                              'global_walkers': W, 'walkers_per_gpu': strong['walkers_per_gpu'],
                              'per_rank_kernel_ms': strong['per_rank_kernel_ms'],
                              'partition': 'contiguous blocks shard_range(W, N, rank), no data-path collective'}
                result['weak'], result['strong'] = weak_obj, strong_obj
                if args.scaling == 'strong':
                    slow = max(range(world), key=lambda r: strong['per_rank_kernel_ms'][r])
                    k_ms, n_slow = strong['per_rank_kernel_ms'][slow], strong['walkers_per_gpu'][slow]
                    ach = bytes_per_eval * n_slow / (k_ms * 1e-3) / 1e9
                    result.update({'scaling': 'strong', 'value': strong_obj['value'], 'ms_per_step': strong_obj['ms_per_step']})
                    result['config'].update({'walkers_per_gpu': strong['walkers_per_gpu'], 'global_walkers': W})
                    result['roofline'].update({'achieved': ach, 'frac': ach / HBM_PEAK_GBS, 'kernel_ms': k_ms, 'traffic': None,
                                               'traffic_source': None, 'per_rank_kernel_ms': strong['per_rank_kernel_ms']})
            if traffic_stale:
                result['roofline']['counters_stale'] = True     # collected from other kernel sources: not quoted
            if parity is not None:
                result['parity'] = parity
            if world == 1:
                result['roofline']['clock_ghz'] = engine_clock_ghz(step, kern_ms, torch)
                counts, counts_stale = load_counters('valu_counts.json')
                if counts_stale:
                    result['valu_counters_stale'] = True        # no roofline_valu below
                gpu_logp = out_t.cpu().numpy()
                if not args.no_variants:
                    fma = result['fp64_fma_stream'] = fp64_stream_ceiling(torch, stream, min(args.prime_seconds, 0.3))
                    result['variants'] = time_variants(ctx, args, step, W, torch, stream, counts, fma)
                    result['kernels'] = time_zoo(data, args, torch, stream, counts, local_rank, fma)
                if not args.no_cpu_baseline:
                    cb, err, same_inf = cpu_baseline(data, taus, log_taus, bounds, theta, gpu_logp)
                    result['cpu_baseline'] = cb
                    result['parity'] = {'max_rel_err_vs_oracle': err, 'neg_inf_rows_match': same_inf,
                                        'tolerance': 1e-10}
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if world > 1 and not args.no_extras and rank == 0:
        # Started by the driver's own torch.distributed.run (no parent of ours to do it): AFTER the
        # result line is out and the process group is gone, rank 0 lets go of its device memory and
        # starts the extras as a fresh child process -- `bench.py --extras-only`, which launches its
        # own group of ranks -- waits for it within its limit and exits 0 whatever became of it.
        if not rehearse:
            del theta_t, out_t
            ctx.close()
            torch.cuda.empty_cache()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] +
                               [a for a in sys.argv[1:] if a != '--extras-only'] + ['--extras-only'],
                               env=rank_group_env(), stdout=subprocess.PIPE, text=True, timeout=EXTRAS_TIMEOUT_S + 60)
            for ln in r.stdout.splitlines():
                print(ln, file=sys.stderr)
            if r.returncode != 0:
                print(f'bench.py: the extras group ended with status {r.returncode}; the result line above is '
                      'not affected', file=sys.stderr, flush=True)
        except Exception as ex:      # noqa: BLE001 -- includes the timeout; never the run's status
            print(f'bench.py: extras not completed: {type(ex).__name__}: {ex}', file=sys.stderr, flush=True)


def sampler_extra(args, dist, torch, rank, world, local_rank, walkers=32768, steps=200):
    """BASELINE config 4 on a group of ranks of its own: Debye decomposition (S = 40 relaxation
    times, the bundled 20-frequency grid, poly_deg 5), 32768 walkers sharded over the GPUs, one
    all-gather per stretch half-step.  Three drivers of the same sharded half-step, least risky
    first, each with its own JSON line on rank 0's STDERR as soon as it is done: the Python loop
    over torch.distributed, then (RCCL groups) the C loop bisip_stretch_run_sharded_dev on a
    communicator of its own and on the one torch.distributed built.  Between them BASELINE
    config 5 (batch_extra: whole replicas, no collective)."""
    import threading
    import numpy as np
    watchdog = threading.Timer(EXTRAS_TIMEOUT_S - 30, lambda: (print('bench.py: sampler extra timed out', file=sys.stderr,
                                                                     flush=True), os._exit(0)))
    watchdog.daemon = True
    watchdog.start()
    try:
        import bisip_amd
        from bisip_amd.sampler import DeviceEnsembleSampler
        m = bisip_amd.PolynomialDecomposition(bisip_amd.DataFiles()['SIP-K389175'], nwalkers=walkers,
                                              nsteps=steps, device=local_rank)
        ctx = m._context()
        centre = np.array([1.0, 0.005, -0.003, -0.001, 0.0005, 0.0002, 0.00001])
        p0 = centre + 1e-4 * np.random.RandomState(2024).randn(walkers, 7)

        def run(n, **kw):
            np.random.seed(7)
            # the single-GPU reference run takes what the sampler itself would take (at cfg4's size the multi-workgroup
            # kernel); with several ranks rehearsing on ONE device its workgroups could not all be resident: launches
            s = DeviceEnsembleSampler(walkers, 7, ctx, rng='philox', seed=11,
                                      persistent=(None if not kw and not args.same_device else False),
                                      chain_on_device=True, **kw)
            if kw:
                s._sharded_comm()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.run_mcmc(p0, n)
            torch.cuda.synchronize()
            dist.barrier()
            return s, time.perf_counter() - t0

        if args.same_device:
            steps = 40                                 # rehearsal through host memory: keep it short
        run(20)
        f, dtf = run(steps)                        # every rank alone, fused half-steps, same stream
        fused = np.array(f._coords)
        loops = ['python'] + (['rccl-own', 'rccl'] if args.backend == 'nccl' else [])
        for i, loop in enumerate(loops):
            sharded = dict(distributed=True, force_sharded_path=True, sharded_loop=loop)
            run(20, **sharded)[0].close()              # clocks, communicator, allocations
            s, dt = run(steps, **sharded)
            final = torch.from_numpy(np.ascontiguousarray(s._coords))
            if args.backend == 'nccl':
                final = final.to(f'cuda:{local_rank}')
            every = torch.empty((world * final.shape[0], final.shape[1]), dtype=final.dtype, device=final.device)
            dist.all_gather_into_tensor(every, final)      # rank r's state = rows [r*W, (r+1)*W)
            every = every.view(world, final.shape[0], final.shape[1])
            same = bool(all(torch.equal(every[0], every[r]) for r in range(world)))
            rec = {'sampler_cfg4': {
                'config': 'Debye decomposition S=40 N=20 poly_deg=5, stretch move', 'walkers': walkers,
                'n_gpus': world, 'iterations': steps, 'driver': s.last_path, 'sharded_loop': loop,
                'us_per_half_step': dt / steps / 2 * 1e6, 'walker_steps_per_s': walkers * steps / dt,
                'payload_bytes_per_rank_per_half_step': -(-(walkers // 2) // world) * 9 * 8,
                'state_identical_on_every_rank': same,
                'equals_single_gpu_fused_chain': bool(np.array_equal(s._coords, fused)),
                'single_gpu_fused_us_per_half_step': dtf / steps / 2 * 1e6,
                'single_gpu_fused_walker_steps_per_s': walkers * steps / dtf}}
            s.close()
            if rank == 0:
                print(json.dumps(rec), file=sys.stderr, flush=True)
            if i == 0 and not args.same_device:
                batch_extra(dist, torch, rank, world, local_rank)
    except Exception as ex:      # reported; the group's status is ignored by whoever started it
        print(f'bench.py: sampler extra failed on rank {rank}: {type(ex).__name__}: {ex}', file=sys.stderr, flush=True)
    finally:
        watchdog.cancel()


def batch_extra(dist, torch, rank, world, local_rank, spectra_per_gpu=512, walkers=256, iterations=4000):
    """BASELINE config 5 on the ranks of this run: every GPU inverts its own block of 512 synthetic spectra
    (double Cole-Cole, 32 frequencies, 256 walkers each) -- whole replicas, no collective while they run; the
    spectrum offset keys each spectrum's stream by its survey index.  One JSON line on rank 0's STDERR."""
    import numpy as np
    import bisip_amd
    from bisip_amd.synthetic import synthetic_columns
    E = spectra_per_gpu
    first = rank * E
    tables = [synthetic_columns(32, first + i) for i in range(E)]
    b = bisip_amd.SpectraBatch('PeltonColeCole', tables, nwalkers=walkers, nsteps=iterations // 40, n_modes=2, device=local_rank)
    b.ctx.set_spectrum_offset(first)
    p0 = np.array([1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6]) + 1e-3 * np.random.RandomState(rank).randn(E, walkers, 7)
    b.nsteps = 5
    b.fit(p0, seed=3, thin_by=40, chain='device')        # kernels loaded, allocator warm
    b.nsteps = iterations // 40
    best = None
    for _ in range(2):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.fit(p0, seed=3, thin_by=40, chain='device')
        mean = b.get_param_mean(discard=b.nsteps // 2)
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    t = torch.tensor([best], dtype=torch.float64, device=f'cuda:{local_rank}')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        dt = float(t[0])
        print(json.dumps({'sampler_cfg5': {
            'config': 'double Cole-Cole, 32 frequencies, 256 walkers per spectrum, stretch move, chain and posterior means on the device',
            'n_gpus': world, 'spectra': E * world, 'spectra_per_gpu': E, 'iterations': b.nsteps * 40, 'seconds': dt,
            'walker_steps_per_s': E * world * walkers * b.nsteps * 40 / dt, 'path': b._sampler.last_path,
            'collectives_on_the_data_path': 0, 'finite_means': bool(np.isfinite(mean).all())}}), file=sys.stderr, flush=True)
    b.close()


def time_variants(ctx, args, step, W, torch, stream, counts, fma=None):
    """The other formulations of the same log-probability (same theta, same W), each primed
    like the headline before it is timed."""
    variants = {}
    for v in ('reduced', 'reduced_comp', 'collapsed', 'faithful', 'wave'):
        ctx.set_variant(v)
        prime(step, min(args.prime_seconds, 0.3), torch)
        k = max(3, min(args.steps, 10))
        _, ms = time_launches(step, k, 2, torch, stream)
        rec = {'evals_per_s': W / (ms * 1e-3), 'kernel_ms': ms, 'kernel': ctx.kernel_name,
               'hbm_frac': 64 * W / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if v == 'wave':
            rec['note'] = ('not a product path: the north star\'s one-wave-per-walker mapping, kept as a measured '
                           'comparison; AUTO never selects it')
        rec['clock_ghz'] = engine_clock_ghz(step, ms, torch)
        rv = against_stream(with_clock(valu_roofline(v, W, ms, counts), rec['clock_ghz']), fma)
        if rv:
            rec['roofline_valu'] = rv
        variants[v] = rec
    ctx.set_variant(args.variant)
    return variants


def zoo_contexts(data, local_rank, W):
    import torch
    from bisip_amd import _hip
    from bisip_amd.synthetic import synthetic_theta
    for label, (model, kw, bounds) in kernel_zoo().items():
        ctx = _hip.HipContext(getattr(_hip, 'MODEL_' + model), data['w'], data['zn'], data['zn_err'], bounds,
                              device=local_rank, **kw)
        th = torch.from_numpy(synthetic_theta(bounds[0], bounds[1], W, seed=7)).to(f'cuda:{local_rank}')
        out = torch.empty(W, dtype=torch.float64, device=f'cuda:{local_rank}')
        yield label, ctx, th, out, bounds.shape[1]


ZOO_WALKERS = 1 << 22


def time_zoo(data, args, torch, stream, counts, local_rank, fma=None):
    """The transcendental-bound models at N = 32 (cfg2 / cfg5's kernels), 4M walkers each."""
    out = {}
    for label, ctx, th, o, ndim in zoo_contexts(data, local_rank, ZOO_WALKERS):
        def fn():
            ctx.logprob_dev(th.data_ptr(), ZOO_WALKERS, o.data_ptr(), stream.cuda_stream)
        prime(fn, min(args.prime_seconds, 0.3), torch)
        _, ms = time_launches(fn, 10, 2, torch, stream)
        rec = {'evals_per_s': ZOO_WALKERS / (ms * 1e-3), 'kernel_ms': ms, 'kernel': ctx.kernel_name,
               'walkers': ZOO_WALKERS, 'n_freq': N_FREQ,
               'hbm_frac': 8 * (ndim + 1) * ZOO_WALKERS / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        rec['clock_ghz'] = engine_clock_ghz(fn, ms, torch)
        rv = against_stream(with_clock(valu_roofline(label, ZOO_WALKERS, ms, counts), rec['clock_ghz']), fma)
        if rv:
            rec['roofline_valu'] = rv
        out[label] = rec
        ctx.close()
    return out


def pmc_pass(ctx, step, theta_t, out_t, data, torch):
    """ONE launch per kernel in a fixed order (the four PolynomialDecomposition formulations at
    the headline W, then the zoo): under `rocprofv3 --pmc SQ_INSTS_VALU` each dispatch's counter
    is that kernel's VALU wave-instruction count per launch."""
    order = []
    for v in ('reduced', 'reduced_comp', 'collapsed', 'faithful', 'wave'):
        ctx.set_variant(v)
        step()
        torch.cuda.synchronize()
        order.append({'label': v, 'kernel': ctx.kernel_name, 'walkers': int(theta_t.shape[0])})
    stream = torch.cuda.current_stream()
    for label, zctx, th, o, _ in zoo_contexts(data, ctx.device, ZOO_WALKERS):
        zctx.logprob_dev(th.data_ptr(), ZOO_WALKERS, o.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        order.append({'label': label, 'kernel': zctx.kernel_name, 'walkers': ZOO_WALKERS})
        zctx.close()
    print(json.dumps({'pmc_pass': order, 'kernel_sources_sha256': kernel_sources_sha256()}))


if __name__ == '__main__':
    main()
