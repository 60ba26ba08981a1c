#!/bin/bash
# PMC counters of the cfg5 half-step kernel (short run); prints per-wave averages
out=$PWD/gpurun_out; mkdir -p $out; repo=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  d=/tmp/pmc_cfg5_$(echo $set | cut -c1-12 | tr ' ' '_')
  env $CFG5_ENV rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 $repo/benchmarks/cfg5_batch.py --chain device --steps 10 --thin-by 10 > /dev/null 2>&1
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_stretch_half' in r['Kernel_Name'] or 'k_stretch_persistent' in r['Kernel_Name']:
            tot[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in tot.items():
    print(f'{k:24s} dispatches {len(v):4d}  mean per dispatch {sum(v)/len(v):14.1f}  per wave(1024) {sum(v)/len(v)/1024:10.1f}')
PY
done
