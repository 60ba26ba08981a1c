#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE; see
benchmarks/collect_profiles.sh) into profiles/rNN_bench_pmc_summary.csv and
profiles/pmc_traffic.json (the number bench.py reports as roofline.traffic).

    python benchmarks/summarize_pmc.py gpurun_out profiles r01

When gpurun_out/pmc_valu/ holds a `rocprofv3 --pmc SQ_INSTS_VALU` pass of `bench.py --pmc-pass`
(one launch per kernel, order printed by the run itself in gpurun_out/pmc_valu.json) it also
writes profiles/valu_counts.json: VALU wave-instructions per eval of every kernel bench.py
times, the numerator of `roofline_valu`.

HBM bytes per launch = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024: on gfx950 the
fetch counter tallies the 128-byte requests of a 16-B/lane coalesced stream at 64 B
(MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is taken as is.
"""
import csv
import glob
import json
import os
import sys

src, dst, tag = sys.argv[1:4]


def sources_hash(path, key):
    """The SHA-256 of the kernel sources as the profiled run on the GPU box reported it (bench.py prints it
    in its result line and in the --pmc-pass line): what bench.py compares with the tree it runs from."""
    for line in open(path):
        if line.startswith('{"' + key + '"'):
            h = json.loads(line).get('kernel_sources_sha256')
            if h:
                return h
    sys.exit(f'{path}: no {key} line with kernel_sources_sha256 (collected by an older bench.py?)')


def newest_run(pattern):
    """The counter CSV of the LAST run only: gpurun merges every call's files into the same local
    directory under PID-derived names, so earlier runs' files sit next to the new one."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


KERNEL = 'k_logprob_pd_reduced'
rows = []
mean = {}
grid = None
for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = newest_run(os.path.join(src, f'pmc_{counter}', '*', '*counter_collection.csv'))
    if not files:
        sys.exit(f'no counter_collection.csv for {counter} under {src}')
    vals, name = [], None
    for f in files:
        for r in csv.DictReader(open(f)):
            if KERNEL in r['Kernel_Name'] and r['Counter_Name'] == counter:
                vals.append(float(r['Counter_Value']))
                name, grid = r['Kernel_Name'], int(r['Grid_Size'])
    mean[counter] = sum(vals) / len(vals)
    rows.append([counter, len(vals), '%.3f' % min(vals), '%.3f' % max(vals), '%.3f' % mean[counter], name, grid])
with open(os.path.join(dst, f'{tag}_bench_pmc_summary.csv'), 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(['counter', 'dispatches', 'min_KB', 'max_KB', 'mean_KB', 'kernel', 'grid_size'])
    w.writerows(rows)
fetch = 2.0 * mean['FETCH_SIZE'] * 1024.0
write = mean['WRITE_SIZE'] * 1024.0
W = grid
hashes = {sources_hash(os.path.join(src, f'pmc_{c}.json'), 'metric') for c in ('FETCH_SIZE', 'WRITE_SIZE')}
if len(hashes) != 1:
    sys.exit('the FETCH_SIZE and WRITE_SIZE passes ran different kernel sources')
rec = {
    'kernel_sources_sha256': hashes.pop(),
    'walkers': W, 'kernel': KERNEL, 'hbm_bytes_per_launch': fetch + write,
    'fetch_bytes_corrected': fetch, 'write_bytes': write,
    'FETCH_SIZE_KB_raw': mean['FETCH_SIZE'], 'WRITE_SIZE_KB_raw': mean['WRITE_SIZE'],
    'algorithmic_bytes_per_launch': 64 * W,
    'note': f'separate rocprofv3 --pmc passes (profiles/{tag}_bench_pmc_summary.csv); FETCH_SIZE doubled per '
            'MI355X_MICROARCH.md HBM section (gfx950 tallies 128-B requests of a 16-B/lane coalesced stream '
            'at 64 B); WRITE_SIZE taken as is (equals 8 B x W exactly)',
}
json.dump(rec, open(os.path.join(dst, 'pmc_traffic.json'), 'w'), indent=1)
print(json.dumps(rec))


def valu_counts():
    order_file = os.path.join(src, 'pmc_valu.json')
    files = newest_run(os.path.join(src, 'pmc_valu', '*', '*counter_collection.csv'))
    if not files or not os.path.exists(order_file):
        return
    order = None
    for line in open(order_file):
        if line.startswith('{"pmc_pass"'):
            order = json.loads(line)['pmc_pass']
    if order is None:
        sys.exit(f'no pmc_pass line in {order_file}')
    disp = []
    other = {}          # dispatch id -> the other SQ counters of the pass (summed over the rows of a dispatch)
    for f in files:
        for r in csv.DictReader(open(f)):
            if 'k_logprob' not in r['Kernel_Name']:
                continue
            if r['Counter_Name'] == 'SQ_INSTS_VALU':
                disp.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value']), int(r['Grid_Size'])))
            else:
                o = other.setdefault(int(r['Dispatch_Id']), {})
                o[r['Counter_Name']] = o.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    disp.sort()
    # a dispatch may appear once per XCD/agent row: sum rows of one dispatch id
    merged = {}
    for d, name, val, grid in disp:
        m = merged.setdefault(d, [name, 0.0, grid])
        m[1] += val
    ids = sorted(merged)
    rows = [merged[k] for k in ids]
    if len(rows) != len(order):
        sys.exit(f'{len(rows)} log-prob dispatches in the counter file, {len(order)} in the pass order')
    out = {}
    for o, (name, val, grid) in zip(order, rows):
        short = o['kernel'].split('<')[0]
        if short.endswith('_comp'):      # the compensated tier is a template flag of the same kernel
            short = short[:-5]
        if short not in name:
            sys.exit(f'dispatch order mismatch: expected {o["kernel"]}, saw {name}')
        out[o['label']] = {'kernel': name, 'walkers': o['walkers'], 'SQ_INSTS_VALU_per_launch': val,
                           'valu_wave_instr_per_eval': val / o['walkers']}
        extra = other.get(ids[len(out) - 1], {})
        if 'SQ_INSTS_VALU_TRANS_F64' in extra:
            # v_rcp_f64 / v_sqrt_f64 / v_rsq_f64 issue at a quarter of the FMA rate (benchmarks/micro/valu_rates.hip):
            # one of them takes four issue slots
            t = extra['SQ_INSTS_VALU_TRANS_F64'] / o['walkers']
            out[o['label']].update({'trans_f64_wave_instr_per_eval': t,
                                    'issue_slots_per_eval': val / o['walkers'] + 3.0 * t})
        wc = extra.get('SQ_WAVE_CYCLES')
        if wc:
            out[o['label']]['wave_cycle_fractions'] = {
                'parked_at_a_wait (SQ_WAIT_ANY)': round(extra.get('SQ_WAIT_ANY', 0.0) / wc, 3),
                'stalled_at_issue (SQ_WAIT_INST_ANY)': round(extra.get('SQ_WAIT_INST_ANY', 0.0) / wc, 3),
                'issuing (SQ_ACTIVE_INST_ANY)': round(extra.get('SQ_ACTIVE_INST_ANY', 0.0) / wc, 3),
                'issuing_valu (SQ_ACTIVE_INST_VALU)': round(extra.get('SQ_ACTIVE_INST_VALU', 0.0) / wc, 3)}
            out[o['label']]['smem_instr_per_wave'] = round(extra.get('SQ_INSTS_SMEM', 0.0) / (o['walkers'] / 64.0), 1)
    out['kernel_sources_sha256'] = sources_hash(order_file, 'pmc_pass')
    out['_note'] = (f'rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_* SQ_ACTIVE_* SQ_WAVE_CYCLES SQ_INSTS_SMEM on '
                    f'`python3 bench.py --pmc-pass` ({tag}); one launch per kernel; wave-instructions summed over all waves of the '
                    'launch; issue_slots_per_eval = all VALU + 3 x the quarter-rate fp64 transcendentals')
    json.dump(out, open(os.path.join(dst, 'valu_counts.json'), 'w'), indent=1)
    print(json.dumps(out))


valu_counts()

