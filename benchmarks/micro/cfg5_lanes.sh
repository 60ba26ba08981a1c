#!/bin/bash
# cfg5 half-step under 1 / 2 / 4 lanes per walker + kernel trace; run on the GPU box from the repo root
set -e
out=$PWD/gpurun_out; mkdir -p $out
for L in 1 2 4; do
  BISIP_STRETCH_LANES=$L python3 benchmarks/cfg5_batch.py --chain device 2>/dev/null | tee $out/cfg5_L$L.json
done
python3 benchmarks/cfg5_batch.py --chain device --persistent 2>/dev/null | tee $out/cfg5_persistent.json
for L in 1 2; do
  (cd /tmp && BISIP_STRETCH_LANES=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg5_$L -- python3 $OLDPWD/benchmarks/cfg5_batch.py --chain device > /dev/null 2>&1)
  find /tmp/prof_cfg5_$L -name '*kernel_stats.csv' -exec cp {} $out/cfg5_L${L}_kernel_stats.csv \;
  head -4 $out/cfg5_L${L}_kernel_stats.csv | cut -c1-260
done
python3 -c "
import cProfile, pstats, sys, io
sys.argv=['cfg5_batch.py','--chain','device']
sys.path.insert(0,'benchmarks')
import runpy
pr=cProfile.Profile(); pr.enable(); runpy.run_path('benchmarks/cfg5_batch.py', run_name='__main__'); pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue())
" > $out/cfg5_cprofile.txt 2>&1
