#!/bin/bash
# kernel-trace of the cfg5 half-step kernel under the variants named in $@ (env assignments), e.g.
#   bash benchmarks/micro/cfg5_trace.sh "" "BISIP_NO_LDS_STAGING=1"
out=$PWD/gpurun_out; mkdir -p $out; repo=$PWD
i=0
for v in "$@"; do
  i=$((i+1))
  (cd /tmp && env $v TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg5_v$i -- python3 $repo/benchmarks/cfg5_batch.py --chain device > $out/cfg5_v$i.json 2>/dev/null)
  echo "== variant $i: '$v'"
  find /tmp/prof_cfg5_v$i -name '*kernel_stats.csv' -exec grep k_stretch_half {} \; | cut -c1-40,150-400
done
