#!/bin/bash
# SQ counters of the persistent sampler kernel on a 512 x 256 PolynomialDecomposition batch (400 iterations
# in chunks; 1024 waves): per wave and per 800 half-steps -- a launch holds fewer, compare runs, not absolutes.  Run on the GPU box from the repo root.
out=$PWD/gpurun_out; mkdir -p $out; repo=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  d=/tmp/pmc_pd_$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 $repo/benchmarks/batch_models.py --only ${MODEL:-Polynomial} --iterations 400 > /dev/null 2>&1
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_stretch_persistent' in r['Kernel_Name']:
            tot[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in tot.items():
    m = sum(v) / len(v)
    print(f'{k:24s} dispatches {len(v):3d}  per dispatch {m:14.1f}  per wave per half-step {m / 1024 / 800:10.2f}')
PY
done
