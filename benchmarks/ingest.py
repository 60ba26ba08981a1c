#!/usr/bin/env python3
"""Ingest of a survey: E spectrum files -> the stacked operands of a batch context.
Times the reference's way (np.loadtxt + per-file arithmetic, one file after the other -- what a loop
over Inversion objects does, src/bisip/utils.py:108-146) against load_data_batch (C parser on a few
threads + arithmetic batched over the spectra) and checks that both give the same bits.  Host only.

    python benchmarks/ingest.py [--spectra 4096] [--nfreq 32]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument('--spectra', type=int, default=4096)
ap.add_argument('--nfreq', type=int, default=32)
args = ap.parse_args()

from bisip_amd.synthetic import synthetic_columns
from bisip_amd.utils import cpu_quota, load_data, load_data_batch

d = tempfile.mkdtemp(prefix='bisip_ingest_')
try:
    paths = []
    for i in range(args.spectra):
        p = os.path.join(d, f'spectrum{i:05d}.csv')
        np.savetxt(p, synthetic_columns(args.nfreq, i), delimiter=',', header='freq,amp,pha,amp_err,pha_err')
        paths.append(p)
    load_data_batch(paths[:8])                       # library loaded, page cache warm for those
    t0 = time.perf_counter()
    ref = [load_data(p) for p in paths]
    t_ref = time.perf_counter() - t0
    out = {'spectra': args.spectra, 'n_freq': args.nfreq, 'cpus': cpu_quota(),
           'per_file_numpy_s': round(t_ref, 4), 'per_file_numpy_us_per_file': round(1e6 * t_ref / args.spectra, 1)}
    for threads in (1, min(cpu_quota(), 16)):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            b = load_data_batch(paths, threads=threads)
            best = min(best, time.perf_counter() - t0)
        out[f'batch_{threads}_threads_s'] = round(best, 4)
        out[f'batch_{threads}_threads_speedup'] = round(t_ref / best, 1)
    same = all(np.array_equal(b[k][i], r[k]) for i, r in enumerate(ref) for k in ('w', 'zn', 'zn_err'))
    out['bit_identical'] = bool(same and all(b['norm_factor'][i] == r['norm_factor'] for i, r in enumerate(ref)))
    tabs = [np.loadtxt(p, skiprows=1, delimiter=',') for p in paths[:512]]
    t0 = time.perf_counter()
    load_data_batch(tabs)
    out['tables_512_s'] = round(time.perf_counter() - t0, 4)
    print(json.dumps(out))
    sys.exit(0 if out['bit_identical'] else 1)
finally:
    shutil.rmtree(d, ignore_errors=True)
