#!/bin/bash
# The one evidence run of a round, at the FINAL kernel sources, in stages that each fit one gpurun call (<= 20 min):
#   bash benchmarks/collect_final.sh <tag> tests|bench|sweep|sampler|micro|campaign-a|campaign-b|campaign-c|campaign-group
# Everything lands under gpurun_out/ (scratch); `bash benchmarks/import_profiles.sh <tag>` then copies what is judged
# into profiles/ and regenerates the tables.  (Replaces the per-round collect_rNN_*.sh scripts.)
set -o pipefail
tag=${1:?round tag, e.g. r05}
what=${2:?stage}
out=gpurun_out
mkdir -p $out
fz() {   # fz <kind> <script> <args...>: one campaign, its summary line appended to the kind's file
  local kind=$1 script=$2; shift 2
  python3 benchmarks/$script "$@" 2>> $out/fuzz.err | tail -1 >> $out/fuzz_${kind}_summary.jsonl
  echo "$script $* done"
}
case $what in
tests)
  python3 -m pytest tests -m gpu -q > $out/${tag}_gpu_tests_final.txt 2>&1; echo "pytest rc=$?" >> $out/${tag}_gpu_tests_final.txt
  tail -3 $out/${tag}_gpu_tests_final.txt ;;
bench)
  # counters summarised ON THE BOX so that the bench line taken afterwards quotes them
  bash benchmarks/collect_profiles.sh bench > $out/collect_bench.log 2>&1
  python3 benchmarks/summarize_pmc.py $out profiles $tag > /dev/null && python3 bench.py > $out/bench_final.json 2> $out/bench_final.err
  echo "bench rc=$?"; cut -c1-300 $out/bench_final.json
  # the ceiling bench.py holds the compute-bound kernels to, in the open: cold / sustained, the clock the waves saw
  [ -x benchmarks/micro/fp64_stream_ceiling ] && timeout -k 5 120 benchmarks/micro/fp64_stream_ceiling 0.5 > $out/micro_fp64_stream_ceiling.txt 2>&1 ;;
sweep|sampler|micro)
  bash benchmarks/collect_profiles.sh $what > $out/collect_$what.log 2>&1; echo "rc=$?"; tail -3 $out/progress.log
  if [ $what = micro ]; then
    python3 benchmarks/micro/group_sampler.py 2> /dev/null > $out/micro_group_sampler.jsonl
    for m in cc2 pd; do      # plain layout / packed state, stream arrays / packed state, stream drawn in place
      BIG_MODEL=$m BISIP_NO_PACKED_STATE=1 python3 benchmarks/micro/ab_big_ensemble.py 2> /dev/null | grep '^{'
      BIG_MODEL=$m BISIP_NO_INLINE_DRAW=1 python3 benchmarks/micro/ab_big_ensemble.py 2> /dev/null | grep '^{'
      BIG_MODEL=$m python3 benchmarks/micro/ab_big_ensemble.py 2> /dev/null | grep '^{'
    done > $out/micro_ab_big_ensemble_packed.jsonl
    [ -x benchmarks/micro/random_lines ] && (timeout -k 5 120 benchmarks/micro/random_lines 1048576 && timeout -k 5 120 benchmarks/micro/random_lines 8388608) > $out/micro_random_lines.txt 2>&1
    for i in 1 2 3; do
      python3 benchmarks/cfg4_sampler.py --steps 200 --fused --chain device --repeat 7 | grep '^{'
      python3 benchmarks/cfg4_sampler.py --steps 200 --fused --chain device --repeat 7 --no-guard | grep '^{'
    done > $out/guard_overhead_cfg4.jsonl 2> /dev/null
  fi ;;
campaign-a)
  fz parity fuzz_parity.py --cases 1500 --seed 46 --widen 3
  fz parity fuzz_parity.py --cases 4000 --seed 64
  fz parity fuzz_parity.py --cases 2000 --seed 65 --widen 1.5
  fz valley fuzz_parity.py --cases 3000 --seed 311 --valley
  fz sampler fuzz_sampler.py --cases 1500 --seed 25
  fz batch fuzz_batch.py --cases 600 --seed 26
  fz parity fuzz_parity.py --cases 10000 --seed 66 ;;
campaign-b)
  fz parity fuzz_parity.py --cases 10000 --seed 67 --widen 1.5
  fz parity fuzz_parity.py --cases 4000 --seed 68 --widen 3
  fz valley fuzz_parity.py --cases 6000 --seed 312 --valley
  fz sampler fuzz_sampler.py --cases 6000 --seed 27
  fz batch fuzz_batch.py --cases 2000 --seed 28 ;;
campaign-c)      # a third set of seeds, taken after the last kernel change of the round (the in-place stream)
  fz parity fuzz_parity.py --cases 12000 --seed 69
  fz parity fuzz_parity.py --cases 6000 --seed 70 --widen 2
  fz valley fuzz_parity.py --cases 6000 --seed 313 --valley
  fz sampler fuzz_sampler.py --cases 6000 --seed 29
  fz batch fuzz_batch.py --cases 2000 --seed 30 ;;
campaign-group)  # the multi-workgroup sampler against one launch per half-step (every case: hundreds to thousands of barriers)
  fz group fuzz_group.py --cases 1500 --seed 41
  fz group fuzz_group.py --cases 1500 --seed 42 ;;
*) echo "unknown stage $what" >&2; exit 2 ;;
esac
