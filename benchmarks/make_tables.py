#!/usr/bin/env python3
"""Regenerates every measured table of DESIGN.md and the whole of profiles/README.md from the
files under profiles/ -- numbers in the prose are never retyped by hand.

    python benchmarks/make_tables.py            # rewrite DESIGN.md's table blocks + profiles/README.md
    python benchmarks/make_tables.py --check    # exit 1 if either file is out of date (used by tests/)

A table block in DESIGN.md looks like

    <!-- TABLE:bench -->
    ...generated...
    <!-- /TABLE:bench -->

and is replaced as a whole.  ROUND selects the profile set (profiles/rNN_*).
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'profiles')
ROUNDS = ['r05', 'r04', 'r03']     # a measurement is quoted from the newest round that repeated it


def rn(suffix):
    """profiles/<round>_<suffix> of the newest round that has it (globs allowed); several, comma separated, each on its own."""
    import glob
    if ', ' in suffix:
        return ', '.join(rn(x) for x in suffix.split(', '))
    for r in ROUNDS:
        if glob.glob(os.path.join(PROF, f'{r}_{suffix}')):
            return f'{r}_{suffix}'
    return f'{ROUNDS[0]}_{suffix}'


ROUND = ROUNDS[0]


def all_rounds(suffix):
    """(round, record) of every line of profiles/<round>_<suffix>, oldest round first: campaigns add up over the
    rounds (a later round re-runs a seed only where a kernel it exercises changed)."""
    for r in reversed(ROUNDS):
        path = os.path.join(PROF, f'{r}_{suffix}')
        if os.path.exists(path):
            for rec in jlines(f'{r}_{suffix}'):
                yield r, rec


def jload(name):
    with open(os.path.join(PROF, name)) as fh:
        return json.load(fh)


def jlines(name):
    with open(os.path.join(PROF, name)) as fh:
        return [json.loads(ln) for ln in fh if ln.strip().startswith('{')]


def sci(x, d=3):
    return f'{x:.{d}g}' if abs(x) < 1e4 else f'{x:.{d - 1}e}'.replace('e+', 'e').replace('e0', 'e')


def md(rows, header):
    out = ['| ' + ' | '.join(header) + ' |', '|' + '---|' * len(header)]
    out += ['| ' + ' | '.join(str(c) for c in r) + ' |' for r in rows]
    return '\n'.join(out)


def kernel_stats(name, needle):
    with open(os.path.join(PROF, name)) as fh:
        for r in csv.DictReader(fh):
            if needle in r['Name']:
                return r
    return None


# ---------------------------------------------------------------------------------------
def table_bench():
    b = jload(rn('bench.json'))
    u = jload(rn('bench_under_rocprof.json'))
    ks = kernel_stats(rn('bench_kernel_stats.csv'), 'k_logprob_pd_reduced')
    t = jload('pmc_traffic.json')
    r = b['roofline']
    cb = b['cpu_baseline']
    cal = cb.get('reference_calibration', {})
    rows = [
        ['value (whole job, 1 GPU)', f"{sci(b['value'], 4)} evals/s", f"`{rn('bench.json')}`"],
        ['ms per step (W = 2^24 walkers per launch)', f"{b['ms_per_step']:.4f}", ''],
        ['kernel duration, HIP events on the launch stream', f"{r['kernel_ms'] * 1e3:.1f} us", 'un-profiled run'],
        ['achieved / peak', f"{r['achieved']:.0f} / {r['peak']:.0f} GB/s = **{r['frac']:.3f}**", '64 B per eval'],
        ['same command under `rocprofv3 --kernel-trace --stats`',
         f"{float(ks['AverageNs']) / 1e3:.2f} us average over {ks['Calls']} launches",
         f"`{rn('bench_kernel_stats.csv')}`"],
        ['in-process HIP-event clock in that profiled run', f"{u['roofline']['kernel_ms'] * 1e3:.2f} us", f"`{rn('bench_under_rocprof.json')}`"],
        ['HBM traffic per launch (PMC, FETCH_SIZE x2 + WRITE_SIZE)',
         f"{t['hbm_bytes_per_launch'] / 1e6:.2f} MB vs {t['algorithmic_bytes_per_launch'] / 1e6:.2f} MB algorithmic "
         f"= x{t['hbm_bytes_per_launch'] / t['algorithmic_bytes_per_launch']:.5f}", f"`pmc_traffic.json`, `{rn('bench_pmc_summary.csv')}`"],
        ['CPU baseline (oracle, kind "port")', f"{sci(cb['value'])} evals/s on {cb['cores']} threads; {sci(cb['value_1core'])} on one",
         f"quota {cb.get('cpu_quota')} of {cb['host_cpus_visible']} visible CPUs"],
    ]
    if cal:
        rows.append(['real reference / oracle, one core (build container)', f"{cal['reference_over_oracle_1core']:.3f}",
                     f"=> reference here ~{sci(cal['estimated_reference_1core_here'])} evals/s per core, "
                     f"~{sci(cal['estimated_reference_all_cores_here'])} on {cb['cores']}; `cpu_calibration.json`"])
    rows.append(['parity spot check inside the run', f"max rel err {b['parity']['max_rel_err_vs_oracle']:.1e} (tolerance 1e-10)", ''])
    return md(rows, ['quantity', 'measured', 'source / note'])


def table_variants():
    b = jload(rn('bench.json'))
    rows = []
    fma = b.get('fp64_fma_stream')
    if fma:
        rows.append(['independent `v_fma_f64` (the ceiling; `bisip_fp64_stream_probe_dev`)', '8 waves per SIMD, full-mantissa operands',
                     f"({sci(fma['wave_instr_per_s'])} wave-instr/s)", f"{fma['kernel_ms'] * 1e3:.1f}", '-', '-',
                     f"{fma['frac_of_nominal_peak']:.2f}", f"{fma['clock_ghz']:.2f}" if fma.get('clock_ghz') else '-', '-', '-', '**1.00**', '-'])
    for group, title in (('variants', 'PolynomialDecomposition P=5, N=32, W=2^24'), ('kernels', 'N=32, W=2^22')):
        for label, v in b[group].items():
            rv = v.get('roofline_valu')
            rows.append([f'`{label}`' + (' (not a product path)' if label == 'wave' else ''), title, f"{sci(v['evals_per_s'])}", f"{v['kernel_ms'] * 1e3:.1f}",
                         f"{v['hbm_frac']:.3f}", f"{rv['valu_wave_instr_per_eval']:.2f}" if rv else '-',
                         f"{rv['frac']:.2f}" if rv else '-',
                         f"{v['clock_ghz']:.2f}" if v.get('clock_ghz') else '-',
                         f"{rv['trans_f64_wave_instr_per_eval']:.2f}" if rv and 'trans_f64_wave_instr_per_eval' in rv else '-',
                         f"{rv['frac_issue_slots']:.2f}" if rv and rv.get('frac_issue_slots') else '-',
                         f"**{rv['frac_of_fma_stream']:.2f}**" if rv and rv.get('frac_of_fma_stream') else '-',
                         ' / '.join(f"{x:.2f}" for x in list(rv['wave_cycle_fractions'].values())[:3]) if rv and rv.get('wave_cycle_fractions') else '-'])
    return md(rows, ['kernel', 'workload', 'evals/s', 'us per launch', 'fraction of HBM peak',
                     'VALU wave-instr per eval (PMC)', 'fraction of the nominal fp64 issue peak (1024 SIMDs x 2.4 GHz / 4)',
                     'engine clock, GHz (in-run probe)', 'of them quarter-rate fp64 (v_rcp_f64 ...: SQ_INSTS_VALU_TRANS_F64)',
                     'ISSUE SLOTS (a quarter-rate instruction = 4) / nominal peak',
                     'issue slots / the MEASURED FMA stream of the same run',
                     "a wave's cycles: parked at a wait / stalled at issue / issuing"])


def table_sweep():
    rows = [[d['case'], f"`{d['kernel']}`", d['N'], d['W'], f"{d['us_per_launch']:.2f}", sci(d['evals_per_s']),
             f"{d['hbm_frac']:.3f}"] for d in jlines(rn('sweep.jsonl'))]
    return md(rows, ['case', 'kernel', 'N', 'W', 'us per launch', 'evals/s', 'fraction of HBM peak'])


def table_forward():
    rows = []
    for d in jlines(rn('host_path.jsonl')):
        if 'forward' in d['case']:
            rows.append([d['case'].replace(' bisip_forward_dev', ''), d['W'], d['us'], sci(d['rows_per_s']),
                         f"{d['write_GBs'] / 1e3:.2f}"])
    return md(rows, ['batched forward (device resident)', 'rows', 'us per launch', 'rows/s', 'TB/s of Z written'])


def table_host_path():
    rows = [[d['case'].replace(' bisip_logprob (host buffers, pageable)', ''), d['W'], sci(d['evals_per_s']), d['GBs_over_pcie']]
            for d in jlines(rn('host_path.jsonl')) if 'host buffers' in d['case']]
    return md(rows, ['`bisip_logprob`, host buffers (PCIe inclusive; never `value`)', 'W', 'evals/s', 'GB/s over the link'])


def table_samplers():
    data = jlines(rn('sampler_bench.jsonl'))
    cases, samplers = [], ['device-philox-persistent', 'device-philox', 'device', 'device-launches', 'host']
    for d in data:
        if d['case'] not in cases:
            cases.append(d['case'])
    rows = []
    for c in cases:
        row = [c]
        for s in samplers:
            m = [d for d in data if d['case'] == c and d['sampler'] == s]
            row.append(f"{m[0]['it_per_s'] / 1e3:.1f} k ({m[0].get('path') or 'host loop'})" if m else '-')
        rows.append(row)
    return md(rows, ['`fit()` workload: iterations/s'] + [f'`{s}`' for s in samplers])


def table_cfg4():
    rows = []
    for f, label in ((rn('cfg4_fused_device_chain.json'), 'single GPU, no collective (what the sampler picks itself)'),
                     (rn('cfg4_sharded_rccl.json'), "sharded: C loop, eval -> ncclAllGather -> apply, torch's communicator"),
                     (rn('cfg4_sharded_rccl-own.json'), 'sharded: C loop, communicator from bisip_rccl_comm_create'),
                     (rn('cfg4_sharded_python.json'), 'sharded: Python loop over torch.distributed (round 1)')):
        d = jload(f)
        rows.append([label, d['n_gpus'], d['driver'], d['us_per_half_step'], d.get('us_per_half_step_long_run') or '-',
                     sci(d['walker_steps_per_s']), d['acceptance']])
    return md(rows, ['cfg4: 32,768 walkers, Debye S=40, P=5', 'ranks', 'driver', 'us per half-step, 200 iterations (enqueue + waits)',
                     'per further half-step of a long run', 'walker-steps/s, 200 iterations end to end', 'acceptance'])


def table_cfg5():
    rows = []
    for f, label in ((rn('cfg5_device_chain.json'), 'persistent kernel (automatic for this shape), chain in HBM'),
                     (rn('cfg5_device_chain_launches.json'), 'one launch per half-step, chain in HBM'),
                     (rn('cfg5_host_chain.json'), 'persistent kernel, chain copied to pinned host memory')):
        d = jload(f)
        rows.append([label, d['path'], d['iterations'], d['seconds'], d.get('us_per_half_step', '-'), sci(d['walker_steps_per_s'])])
    for f, label in ((rn('cfg5_4096_spectra.json'), 'ALL 4096 spectra (1,048,576 walkers) on one GPU, persistent kernel'),
                     (rn('cfg5_4096_spectra_launches.json'), 'ALL 4096 spectra on one GPU, one launch per half-step')):
        if os.path.exists(os.path.join(PROF, f)):
            d = jload(f)
            rows.append([label, d['path'], d['iterations'], d['seconds'], d.get('us_per_half_step', '-'), sci(d['walker_steps_per_s'])])
    return md(rows, ["cfg5: 512 spectra (one GPU's share) x 256 walkers unless stated, double Cole-Cole, N=32", 'path', 'iterations', 'seconds (incl. summaries)',
                     'us per half-step', 'walker-steps/s'])


def table_ingest():
    d = jload(rn('ingest.json'))
    th = max(int(k.split('_')[1]) for k in d if k.startswith('batch_') and k.endswith('_threads_s'))
    rows = [['np.loadtxt + per-file arithmetic, file after file (the reference\'s way)', d['per_file_numpy_s'], round(1e6 * d['per_file_numpy_s'] / d['spectra'], 1), '1'],
            ['`load_data_batch`, 1 thread', d['batch_1_threads_s'], round(1e6 * d['batch_1_threads_s'] / d['spectra'], 1), d['batch_1_threads_speedup']],
            [f'`load_data_batch`, {th} threads', d[f'batch_{th}_threads_s'], round(1e6 * d[f'batch_{th}_threads_s'] / d['spectra'], 1), d[f'batch_{th}_threads_speedup']]]
    return md(rows, [f"ingest of {d['spectra']} spectrum files (N = {d['n_freq']}), GPU box host ({d['cpus']} CPUs); bit-identical: {str(d['bit_identical']).lower()}", 'seconds', 'us per file', 'speed-up'])


def table_batch_models():
    rows = []
    for d in jlines(rn('batch_models.jsonl')):
        a, b = d['persistent'], d['launch_per_half_step']
        rows.append([d['model'], a['us_per_half_step'], sci(a['walker_steps_per_s']), b['us_per_half_step'], sci(b['walker_steps_per_s']), a['acceptance']])
    d0 = jlines(rn('batch_models.jsonl'))[0]
    return md(rows, [f"{d0['spectra']} spectra x {d0['walkers_per_spectrum']} walkers, 32 frequencies, {d0['iterations']} iterations, chain in HBM: model",
                     'persistent kernel: us per half-step', 'walker-steps/s', 'launch per half-step: us per half-step', 'walker-steps/s', 'acceptance'])


def table_survey():
    rows = [[d['model'], d['ingest_and_context_s'], d['fit_s'], d['parameter_summaries_s'], d['model_bands_s'], d['total_s'], sci(d['walker_steps_per_s_end_to_end'])]
            for d in jlines(rn('survey.jsonl'))]
    d0 = jlines(rn('survey.jsonl'))[0]
    return md(rows, [f"a survey end to end on one GPU: {d0['spectra']} spectrum files, {d0['walkers_per_spectrum']} walkers x {d0['iterations']} iterations each; model",
                     'ingest + batch context, s', '`fit()`, s', 'mean, std, 3 percentiles of every parameter, s', 'model-space bands of every spectrum, s', 'total, s', 'walker-steps/s end to end'])


def table_batch_setup():
    rows = [[d['model'], d['kernel'], d['create_s'], d['fit_100_iterations_s'], d['summaries_s'], d.get('model_percentiles_s', '-')] for d in jlines(rn('batch_setup.jsonl'))]
    return md(rows, ['512 spectra x 256 walkers: model', 'kernel', 'batch context creation, s', '`fit()` of 100 iterations, s', 'mean + std + 3 percentiles on the device, s',
                     'model-space percentile bands of every spectrum (50 x 256 samples each), s'])


def table_fuzz():
    rows = []
    for r, d in all_rounds('fuzz_parity_summary.jsonl'):
        a = d['auto_on_polydecomp']
        rows.append([r, f"parity, seed {d['seed']}" + (f", prior boxes widened x{d['widen']}" if d.get('widen') else ''), d['cases'], d['violations'], f"{d['worst_logp_rel_err']:.1e}", f"{d['worst_Z_rel_err']:.1e}",
                     f"{a['reduced']} + {a['reduced_comp']} + {a['collapsed']} of {a['problems']} (reduced + compensated + collapsed)"])
    for name, f in (('sampler', 'fuzz_sampler_summary.jsonl'), ('batch of spectra', 'fuzz_batch_summary.jsonl')):
        for r, d in all_rounds(f):
            rows.append([r, f"{name}, seed {d['seed']}", d['cases'], d['failures'], '-', '-', '-'])
    return md(rows, ['round', 'campaign', 'cases', 'violations', 'worst log-prob rel. err (tol 1e-10)', 'worst Z rel. err (tol 1e-12)',
                     'AUTO on PolynomialDecomposition designs with 2N >= P+2'])


def table_valley():
    rows = []
    for r, d in all_rounds('fuzz_valley_summary.jsonl'):
        v, a = d['valley_rows'], d['auto_on_polydecomp']
        rows.append([r, d['seed'], d['cases'], v['cases'], d['violations'], f"{v['worst_auto_vs_exact']:.1e}", f"{v['worst_comp_vs_exact']:.1e}",
                     f"{v['worst_reference_vs_exact']:.1e}", v['cases_where_reference_is_off'], f"{v['worst_per_frequency_vs_exact']:.1e}",
                     f"{a['reduced']} + {a['reduced_comp']} + {a['collapsed']}"])
    return md(rows, ['round', 'seed', 'cases', 'PolynomialDecomposition cases with valley / shell rows', 'violations (more than 1e-10 from the reference AND from the exact value)',
                     'AUTO: worst distance from the exact value', 'compensated kernel: the same', "the REFERENCE's arithmetic (oracle): the same",
                     'cases where the reference is more than 1e-10 from the exact value', 'per-frequency forms (collapsed / faithful / wave): the same, not judged',
                     'AUTO ran plain + compensated + collapsed'])


def table_latency():
    rows = []
    for d in jlines(rn('micro_small_call_latency.jsonl')):
        u = d['us_per_call']
        rows.append([d['model'], f"`{d['kernel']}`"] + [f"{u[w]['ctx.logprob']:.1f} / {u[w]['model.log_prob']:.1f}" for w in ('16', '32', '64', '256', '4096')])
    return md(rows, ['one log-probability call, host buffers in and out: us per call, `ctx.logprob` / `model.log_prob`', 'kernel',
                     '16 rows', '32 rows', '64 rows', '256 rows', '4096 rows'])


def table_auto_by_degree():
    tot = {}
    for _, d in all_rounds('fuzz_parity_summary.jsonl'):
        for deg, v in d['auto_on_polydecomp']['by_degree'].items():
            t = tot.setdefault(int(deg), dict(problems=0, reduced=0, reduced_comp=0, collapsed=0))
            for k in t:
                t[k] += v[k]
    rows = [[deg, v['problems'], v['reduced'], v['reduced_comp'], v['collapsed'],
             f"{(v['reduced'] + v['reduced_comp']) / v['problems']:.2f}"] for deg, v in sorted(tot.items())]
    return md(rows, ['poly_deg', 'problems', 'plain reduced', 'compensated reduced', 'collapsed', 'fraction on a reduced kernel'])


def table_group_sampler():
    rows = []
    for d in jlines(rn('micro_group_sampler.jsonl')):
        g, l = d.get('persistent-multi-workgroup'), d.get('launch-per-half-step')
        if not g or not l:
            continue
        rows.append([d['case'], d['walkers'], f"{g['device_us_per_half_step']:.2f}", f"{l['device_us_per_half_step']:.2f}",
                     f"{l['device_us_per_half_step'] / g['device_us_per_half_step']:.2f}", f"{g['it_per_s'] / 1e3:.1f}", f"{l['it_per_s'] / 1e3:.1f}"])
    return md(rows, ['one ensemble beyond a workgroup (philox stream, chain on the device, 1000 iterations)', 'walkers',
                     'us per half-step: ONE launch per chunk, a barrier among the workgroups per half-step (`k_stretch_group`)',
                     'us per half-step: a launch each', 'ratio', 'k iterations/s end to end: multi-workgroup', 'launches'])


def table_big_ensemble():
    rows = []
    for d in jlines(rn('micro_ab_big_ensemble_packed.jsonl')):
        for W in ('524288', '1048576'):       # (131,072 walkers are in the file as well)
            r = d[W]
            rows.append([{'cc2': 'double Cole-Cole', 'pd': 'PolynomialDecomposition (reduced)'}[d['model']], int(W),
                         'one 64-byte row per walker' if d['packed_state'] else 'coords (W, ndim) + logp (W,)',
                         {'in place': 'drawn in place', 'arrays': 'arrays (a draw kernel per chunk)'}.get(r.get('stream'), '-'),
                         r.get('us_per_half_step_launches', '-'), r['wall_ms']])
    return md(rows, ['big single ensembles, 200 iterations, philox stream, chain thinned by 50 on the device', 'walkers', 'state of a chunk',
                     'philox stream', "us per half-step: a chunk's launches (HIP events; the draw kernel of the array form not included)",
                     'run_mcmc end to end, ms'])


def table_guard():
    rows = []
    for d in jlines(rn('guard_overhead_cfg4.jsonl')):
        g = d.get('guard') or {}
        runs = sorted(d['seconds_all_runs'])
        rows.append(['on' if g.get('checks') else 'off', f"{runs[0] * 1e3:.2f}", f"{runs[len(runs) // 2] * 1e3:.2f}", g.get('checks', 0), g.get('rows', 0),
                     f"{g.get('worst', 0.0):.1e}" if g.get('checks') else '-'])
    return md(rows, ["cfg4's ensemble (32,768 walkers, 200 iterations, one GPU, fused half-steps): the sampler's guard of the reduced tier", 'fastest of 7 runs, ms', 'median, ms',
                     'checks', 'rows measured', 'worst relative error seen (tolerance 2e-11)'])


TABLES = {
    'group_sampler': table_group_sampler, 'big_ensemble': table_big_ensemble, 'guard': table_guard,
    'bench': table_bench, 'variants': table_variants, 'sweep': table_sweep, 'forward': table_forward,
    'host_path': table_host_path, 'samplers': table_samplers, 'cfg4': table_cfg4, 'cfg5': table_cfg5,
    'fuzz': table_fuzz, 'auto_by_degree': table_auto_by_degree, 'ingest': table_ingest,
    'batch_setup': table_batch_setup, 'batch_models': table_batch_models, 'survey': table_survey,
    'valley': table_valley, 'latency': table_latency,
}

FILES = [
    (rn('bench.json'), '`python bench.py`', 'the headline line: value, roofline (HIP-event kernel time), variants and other kernels with `roofline_valu`, cpu_baseline + reference calibration, parity'),
    (rn('bench_kernel_stats.csv'), '`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-variants`', 'average duration of the headline kernel under the profiler'),
    (rn('bench_under_rocprof.json'), 'stdout of that run', 'the in-process HIP-event clock under the profiler (agrees with the trace)'),
    (rn('bench_pmc_summary.csv, pmc_traffic.json'), 'two more runs with `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes), `benchmarks/summarize_pmc.py`', 'HBM bytes per launch; bench.py copies it into `roofline.traffic`'),
    ('valu_counts.json', '`rocprofv3 --pmc SQ_INSTS_VALU -- python3 bench.py --pmc-pass`, `benchmarks/summarize_pmc.py`', 'VALU wave-instructions per eval of every kernel bench.py times: the numerators of `roofline_valu`'),
    ('cpu_calibration.json', '`python benchmarks/cpu_calibration.py` (build container only: imports the real reference)', 'real reference vs oracle on one core, same rows: bit-identical values, rate ratio'),
    (rn('sweep.jsonl, sweep_kernel_stats.csv'), '`python benchmarks/sweep.py` (and under `rocprofv3 --kernel-trace --stats`)', 'every log-probability kernel at the BASELINE config shapes'),
    (rn('host_path.jsonl'), '`python benchmarks/host_path.py`', 'PCIe-inclusive `bisip_logprob` rate; `bisip_forward_dev` store rate'),
    (rn('sampler_bench.jsonl, sampler_kernel_stats.csv'), '`python benchmarks/sampler_bench.py`', 'end-to-end `fit()` iterations/s per sampler / stream mode'),
    (rn('cfg4_*.json'), '`python benchmarks/cfg4_sampler.py --steps 200 --chain device [--fused] [--loop rccl|rccl-own|python]`', 'BASELINE config 4 on one GPU: fused vs the sharded half-step driven from C over RCCL vs from Python'),
    (rn('cfg5_*.json, cfg5_kernel_stats.csv'), '`python benchmarks/cfg5_batch.py --chain device --steps 500 --thin-by 40 [--no-persistent]`', "BASELINE config 5, one GPU's share (512 spectra x 256 walkers)"),
    (rn('bench_2ranks_one_device*.json'), '`python bench.py --gpus 2 --backend gloo --same-device --walkers 1048576 --steps 4`', 'the self-launched 2-rank run on one GPU (gloo): result line and the sharded-sampler extra (state identical on both ranks = single-GPU chain)'),
    (rn('ingest.json'), '`python benchmarks/ingest.py` (host only)', 'survey ingest: the C parser + batched arithmetic against np.loadtxt file after file, same bits'),
    (rn('batch_models.jsonl, batch_models_kernel_stats.csv'), '`python benchmarks/batch_models.py` (and `--only Polynomial` under `rocprofv3 --kernel-trace --stats`)', 'the batch-of-spectra sampler for every model at the cfg5 shape; the trace shows the stream draw next to the sampler kernel'),
    (rn('survey.jsonl'), '`python benchmarks/survey.py [--model PeltonColeCole]`', 'a 4096-spectrum survey end to end: files to posterior summaries and model bands'),
    (rn('batch_setup.jsonl'), '`python benchmarks/batch_setup.py`', 'what surrounds a survey run: batch context creation (host precompute on threads), a short fit, the summaries'),
    (rn('soak.json'), '`python benchmarks/soak.py`', 'twenty 100,000-iteration fits and twenty 10,000-iteration batch fits in one process: device memory constant, posterior means within 0.02 sigma of each other, identical summaries for identical seeds'),
    ('r03_fuzz_*_summary.jsonl, r04_fuzz_*_summary.jsonl', '`python benchmarks/fuzz_parity.py --cases N --seed S [--widen X] [--valley]`, `fuzz_sampler.py`, `fuzz_batch.py` (r03: parity seeds 41-63, valley 301-310, sampler 2-23, batch 1-24; r04, all run again at the final kernel sources, i.e. after the paired reciprocals of ColeCole / Shin (`benchmarks/collect_r04_pairs.sh`): parity 46 (`--widen 3`, the seed with round 3\'s one violation), 64, 65 (`--widen 1.5`), 66, 67 (`--widen 1.5`), 68 (`--widen 3`), 69, 70 (`--widen 1.5`), 71 (`--widen 3`), valley 311-313, sampler 25, 27, 29, batch 26, 28, 30)', 'randomised campaigns (one line per seed in the tables below; earlier rounds: `r02_fuzz_*`, 41,000 problems): violations, worst errors, which kernel AUTO ran'),
    ('r03_fuzz_valley_summary.jsonl, r04_fuzz_valley_summary.jsonl', '`python benchmarks/fuzz_parity.py --cases 1500 --seed S --valley` (S = 301..303; 3000 cases each at S = 304..306, 4000 at S = 307..309, 6000 at S = 310; r04: 3000 at S = 311 with the binary128 yardstick and operands)', 'half of the checked PolynomialDecomposition rows along the valley of chi^2 / on the shell logp = 0: distances of every formulation AND of the reference from the exact value'),
    (rn('valley_rows.jsonl, valley_rows_before.jsonl'), '`python benchmarks/valley_rows.py`', 'the kernel AUTO picks, measured on 3000 valley rows per scale (1 ... 1000 sigma) of 216 designs of degree 5-10 (designs above 1e-11 are listed): with this round\'s estimate, and -- `_before` -- with round 2\'s, against plain long double'),
    (rn('micro_collapsed_r3.txt'), '`benchmarks/micro/collapsed_r3`', 'PDCollapsed: the shipped kernel against LDS-staged records, 4 rows per lane, single-wave workgroups and persistent waves, each with the engine clock it ran at and cycles per VALU instruction'),
    (rn('micro_exp2_variants.txt'), '`benchmarks/micro/exp2_variants`', 'a table-driven exp2 (16 / 32 entries in LDS) against the shipped degree-11 polynomial: cycles per exp2 per SIMD, clock, worst ulp -- not adopted'),
    (rn('micro_fit_speed_by_tier.txt'), '`python benchmarks/micro/fit_speed_by_tier.py`', 'fit() iterations/s on the plain and on the compensated reduced kernel (32 / 256 / 4096 walkers), with fit()\'s own measurement of the kernel'),
    (rn('micro_rcp_accuracy.txt'), '`benchmarks/micro/rcp_accuracy` (hipcc from `rcp_accuracy.hip`)', '`v_rcp_f64` is good to 2^-24.4 on gfx950; one Newton step leaves 20 ulp, two 1.00 ulp, ONE cubic step `r(1 + e + e^2)` 1.00 ulp with one instruction less and a shorter chain: what `rcp_nr` does; the reciprocals that the two / four denominators of a pair of frequencies take from one `rcp_nr` of their product (`rcp_joint`): 2.4 / 3.7 ulp at worst over [1, 2^221)'),
    (rn('micro_grid_vs_direct.txt'), '`python benchmarks/micro/grid_vs_direct.py`', 'ColeCole (1-3 modes) and Shin bulk launches on geometric frequency grids (exponentials stepped by multiplication in blocks of eight: 1.38-1.65x) against one exponential per frequency (`BISIP_NO_GRID=1`), with the largest difference between the two; rounded grid and bundled field spectrum as controls (no grid: the same loop twice)'),
    (rn('micro_grid_small_ensembles.txt'), '`python benchmarks/micro/grid_small_ensembles.py`', '`fit()` of one spectrum with 32-1024 walkers and emcee-sized calls, stepped against direct: 1.08-1.31x on an exact grid (lanes that share a walker take whole half-blocks), 1.0 on the bundled spectrum (no grid)'),
    (rn('micro_grid_small_ensembles_before.txt'), 'the same script on two designs that were taken out', 'one frequency per lane and round on a grid: 0.92-1.0x; a series-corrected tier for grids rounded in data files: 0.71-0.94x on the bundled spectrum (1.09-1.16x in bulk launches) -- why field data keeps the direct loop'),
    (rn('micro_persistent_comp_by_degree.txt'), '`python benchmarks/micro/persistent_comp_by_degree.py`, `batch_comp_by_degree.py`', 'the compensated kernel in the persistent sampler against one launch per half-step, by degree, with its triangle in scalar registers (single spectrum) and its low words staged in LDS: single ensembles -- persistent wins at every degree up to 512 walkers (1.6-2.3x at 32 walkers); 512 x 256 batches -- persistent wins up to degree 8, ties at 9; the automatic rules follow'),
    (rn('micro_small_call_latency.jsonl'), '`python benchmarks/micro/small_call_latency.py`', 'one emcee-sized log-probability call with host buffers, every model, 16 ... 4096 rows'),
    (rn('micro_issue_latency.txt'), '`benchmarks/micro/issue_latency`', 'cycles per fp64 FMA for 1/2/4/8 independent chains at 1, 2, 4 waves per SIMD; placement of 1-, 2-, 4-, 8-, 16-wave workgroups'),
    (rn('micro_row_latency.txt'), '`benchmarks/micro/row_latency`', 'cycles of one log-probability row at one wave per SIMD, records from the scalar cache vs staged in LDS'),
    (rn('micro_half_step_phases.txt'), '`benchmarks/micro/half_step_phases`', 'phases of the cfg5 half-step launch by s_memtime'),
    (rn('micro_cfg5_pmc.txt'), '`bash benchmarks/micro/cfg5_pmc.sh`', 'SQ counters of the cfg5 sampler kernel (VALU busy vs waiting)'),
    (rn('micro_batch_pd_pmc.txt'), '`bash benchmarks/micro/batch_pd_pmc.sh`', 'SQ counters of the persistent kernel on a PolynomialDecomposition batch, per wave (divide by the half-steps of a launch)'),
    (rn('micro_batch_logprob_rate.txt'), '`python benchmarks/micro/batch_logprob_rate.py`', 'bulk log-probability of a 512-spectrum batch, every model: the single-spectrum kernels\' rates'),
    (rn('micro_select_vs_sort.txt'), '`python benchmarks/micro/select_vs_sort.py`', 'percentiles of 131,072 columns by radix selection and by the segmented sort: time, and that the doubles are the same'),
    (rn('micro_model_percentile_single.txt'), '`python benchmarks/micro/model_percentile_single.py`', 'get_model_percentile of one model over a 160,000-row chain: the device call against forward + np.percentile'),
    (rn('micro_forward_rows_variants.txt'), '`benchmarks/micro/forward_rows_variants`', 'output path of the batched forward kernels (whole rows at N = 20, 16-frequency tiles at N = 32 / 64)'),
    (rn('micro_persistent_crossover.txt'), '`python benchmarks/micro/persistent_crossover.py`', 'persistent kernel vs launch per half-step by ensemble size and model: the automatic rule'),
    (rn('micro_grid_barrier.txt'), '`benchmarks/micro/grid_barrier`', 'cost of a device-wide barrier (with and without a row exchange) for 64 / 128 / 256 workgroups'),
    (rn('micro_reduced_comp_by_degree.txt'), '`python benchmarks/micro/reduced_comp_by_degree.py`', 'round 4: bulk rate of the QR-reduced kernels by degree, lone spectrum and batch; the compensated tier with the prior decided first and, where the operands come from memory, loaded again instead of spilled (batch: +22 / +31 / +33 % at degree 5 / 7 / 9; a lone spectrum from degree 6 on takes the same route: +14 ... +29 %); before / after and the two variants not kept in the file'),
    (rn('micro_ab_paired_reciprocals.jsonl'), '`python benchmarks/micro/ab_library.py <library>`, the build before the paired reciprocals and the present one alternately (before, after, before, after) on ONE box', 'round 4, last kernel change: ColeCole<1> / ColeCole<2> / Shin take the denominators of frequencies 2k, 2k+1 from one reciprocal: 4.28 -> 4.62e10 (+8 %), 2.77 -> 2.84e10 (+2 %), 2.73 -> 2.91e10 (+7 %) evals/s in bulk, a cfg5-shaped batch fit 5.43 -> 5.26 us per half-step; Dias (untouched) 3.87 / 3.79e10: the run-to-run spread'),
    (rn('micro_ab_big_ensemble.txt'), '`rocprofv3 --kernel-trace --stats -- python3 benchmarks/micro/ab_big_ensemble.py <library>` (builds before / after the paired reciprocals), and `run_mcmc`\'s own `timing` before / after the shortcut of `walkers_independent`', 'ensembles of 131,072 - 1,048,576 double Cole-Cole walkers: the one-lane-per-walker half-step kernel keeps two waves per SIMD for four after the pairing and is no slower (34.5 -> 33.7 us on average); the initial-state test (singular values of the (W, ndim) positions) cost more than 200 iterations of sampling: 3.0 -> 0.9 ms at 32,768 walkers, 152 -> 36 ms at a million'),
    (rn('micro_xcd_barrier.txt'), '`benchmarks/micro/xcd_barrier`', 'round 4: a barrier among the workgroups of ONE XCD (0.7-0.9 us for 8-32 workgroups, no fences: the counter and the rows go through that XCD\'s L2 with sc1 loads; 0 stale rows) and what a stretch half-step\'s row exchange costs on top, naive (every lane writes and gathers 72-B rows) and laid out for it (one lane per walker, 64-B rows, 16-B accesses): 1.4 us at 2,048 walkers, 2.1 at 8,192, **6.1 at cfg4\'s 32,768** on one XCD, 5.2-5.9 spread over the chip with write-through rows -- no better than the 6.4 us kernel boundary it would replace: the persistent multi-workgroup sampler was not built (kill criterion of VERDICT r3 #3)'),
    (rn('guard_overhead_cfg4.jsonl'), '`python benchmarks/cfg4_sampler.py --steps 200 --fused --chain device --repeat 7 [--no-guard]`, alternately, three times', "round 5: what the sampler's guard of the QR-reduced tier costs at cfg4's size (selection of the rows nearest to the shell, their copy, the yardstick): 2-3 %"),
    (rn('micro_group_sampler.jsonl'), '`python benchmarks/micro/group_sampler.py`', 'round 5: `k_stretch_group` (one ensemble of 1,025 ... 8,192 walkers over several workgroups, a barrier of their own per half-step) against one launch per half-step, every model; cfg2 is the first case'),
    (rn('micro_group_phases.txt'), 'a temporary build of `k_stretch_group` with `s_memrealtime` timers around its phases', "round 5: where cfg2's half-step goes inside the kernel: gather 0.58, evaluation 2.15, commit + drain 0.5, barrier 1.07 us"),
    (rn('micro_ab_big_ensemble_packed.jsonl'), '`BIG_MODEL=cc2|pd [BISIP_NO_PACKED_STATE=1 | BISIP_NO_INLINE_DRAW=1] python benchmarks/micro/ab_big_ensemble.py`', 'round 5: ensembles of 131,072 ... 1,048,576 walkers on the plain layout, on the packed state (one 64-byte row per walker) with stream arrays, and on the packed state with the philox stream drawn in place; a chunk\'s launches timed by HIP events, and end to end'),
    (rn('micro_ab_big_ensemble_rows.jsonl'), 'the same script on a build with `k_stretch_half_rows` (rows moved by four lanes each through LDS; not kept)', 'round 5: PolynomialDecomposition 65 -> 63 us, double Cole-Cole 66 -> 73 us per half-step of 524,288 proposals: requests per instruction were not the bound'),
    (rn('big_ensemble_host_setup.json'), '`BIG_MODEL=pd python benchmarks/micro/ab_big_ensemble.py` after the set-up moved to the device', "round 5: run_mcmc's own timing at 131,072 / 524,288 / 1,048,576 walkers: check_s 0 (the independence test runs on the device), 70.7 -> 33.3 ms end to end at a million walkers"),
    (rn('big_ensemble_kernels.txt'), '`BIG_MODEL=pd rocprofv3 --kernel-trace --stats -- python3 benchmarks/micro/ab_big_ensemble.py`, then two `--pmc` passes (FETCH_SIZE, WRITE_SIZE) of the same command', 'round 5, final sources: the big-ensemble half-step with the stream drawn in place -- 29-31 us per 524,288 proposals, FETCH_SIZE 125 B per proposal for 128 algorithmic (no re-reads), WRITE_SIZE 50; the repacking kernels 21-23 us per million walkers'),
    (rn('fuzz_group_summary.jsonl'), '`bash benchmarks/collect_final.sh r05 campaign-group` (`benchmarks/fuzz_group.py`)', 'round 5, final sources: the multi-workgroup sampler (one XCD and all XCDs, 4 ... 256 workgroups, the two-level barrier) against one launch per half-step on random models, sizes 1,025 ... 32,768, thinning, chunking, lanes, streams: chains bit-equal, 0 failures'),
    (rn('extreme_shapes.txt'), '`python benchmarks/extreme_shapes.py`', 'round 5, final sources: frequency counts 100 ... 4096 (the reference has 20), every model and PolynomialDecomposition formulation: log-probability and forward against the oracle (worst 8.9e-15), a device sampler run against the host loop (same chain)'),
    (rn('micro_random_lines.txt'), '`benchmarks/micro/random_lines 1048576`, `... 8388608` (hipcc from `random_lines.hip`)', 'round 5: what the chip delivers for the access pattern of a big ensemble\'s half-step -- whole 64-byte rows at random places: two read per proposal 20.4 us per 524,288, and one written 25.8 (3.3-4.1 TB/s; sequential rows 6-7.4): the half-step (29-31 us with its evaluation and its draw) is at 0.85-0.9 of it'),
    (rn('micro_fp64_stream_ceiling.txt'), '`benchmarks/micro/fp64_stream_ceiling 0.5` (hipcc from `fp64_stream_ceiling.hip`)', 'round 5: what a stream of independent fp64 FMAs reaches (0.82-0.89 of the nominal issue peak with full-mantissa operands, the waves at 2.2-2.3 GHz; 0.87-0.94 on small integers): the ceiling `bench.py` measures in every run (`fp64_fma_stream`)'),
    (rn('micro_fma_operands.txt'), '`benchmarks/micro/fma_operands` (hipcc from `fma_operands.hip`)', 'round 5: a fp64 FMA with three vector sources costs 2-4 % more issue time than one with a scalar source, whatever the banks: not what holds Dias2000 back'),
    (rn('micro_host_pipeline.txt'), 'a scratch experiment with chunk size, staging and thread count as knobs', 'round 5: the host-buffer entry is bound by the host copy out of pageable memory (54 GB/s from a cache-resident 64 MB source, 37-41 GB/s from DRAM at 256 MB however it is staged or overlapped)'),
    (rn('micro_post_run_stall.txt'), '`python benchmarks/micro/post_run_stall.py kernel`, `upload_cost.py plain`', 'the sporadic 20-30 ms delay of the first device work after a synchronisation early in a process'),
]


def readme():
    out = [f'# profiles/ -- measured evidence, round {int(ROUND[1:])} (one MI355X, ROCm 7.2)', '',
           'Generated by `python benchmarks/make_tables.py` from the files in this directory; the commands are what',
           '`bash benchmarks/collect_profiles.sh all` runs on the GPU box (rocprofv3 from `/tmp`, `TMPDIR=/tmp`).',
           'Files named `r01_*` / `r02_*` are earlier rounds\' and are kept for comparison.', '',
           md([[f'`{f}`', c, w] for f, c, w in FILES], ['file', 'command', 'what to read']), '']
    for name, title in (('bench', 'Headline'), ('variants', 'Formulations and the other kernels, with both rooflines'),
                        ('sweep', 'Every log-probability kernel at the BASELINE shapes'), ('forward', 'Batched forward'),
                        ('host_path', 'Host-buffer entry'), ('samplers', '`fit()` workloads'), ('guard', "The device sampler's guard, cost"), ('group_sampler', 'One ensemble over several workgroups (BASELINE config 2 and its neighbours)'), ('big_ensemble', 'Big single ensembles'), ('cfg4', 'BASELINE config 4'),
                        ('cfg5', 'BASELINE config 5'), ('batch_models', 'Batch of spectra, every model'), ('survey', 'A survey end to end'), ('ingest', 'Survey ingest'), ('batch_setup', 'Survey set-up'), ('fuzz', 'Randomised campaigns'), ('valley', 'Valley / shell rows: every formulation and the reference against the exact value'), ('auto_by_degree', 'Which kernel AUTO ran, by polynomial degree'), ('latency', 'One call with host buffers')):
        out += [f'## {title}', '', TABLES[name](), '']
    return '\n'.join(out)


def design_with_tables(text):
    def repl(m):
        name = m.group(1)
        if name not in TABLES:
            raise SystemExit(f'DESIGN.md names an unknown table {name!r}')
        return f'<!-- TABLE:{name} -->\n{TABLES[name]()}\n<!-- /TABLE:{name} -->'
    return re.sub(r'<!-- TABLE:(\w+) -->.*?<!-- /TABLE:\1 -->', repl, text, flags=re.S)


def main():
    check = '--check' in sys.argv
    stale = []
    dpath = os.path.join(ROOT, 'DESIGN.md')
    old = open(dpath).read()
    new = design_with_tables(old)
    if new != old:
        stale.append('DESIGN.md')
        if not check:
            open(dpath, 'w').write(new)
    rpath = os.path.join(PROF, 'README.md')
    old = open(rpath).read() if os.path.exists(rpath) else ''
    new = readme() + '\n'
    if new != old:
        stale.append('profiles/README.md')
        if not check:
            open(rpath, 'w').write(new)
    if check and stale:
        print('out of date (run python benchmarks/make_tables.py):', ', '.join(stale))
        return 1
    print('rewrote' if stale else 'up to date:', ', '.join(stale) if stale else 'DESIGN.md tables, profiles/README.md')
    return 0


if __name__ == '__main__':
    sys.exit(main())
