# Round 4, after the paired reciprocals of ColeCole / Shin: every measurement that runs a changed kernel again, in
# four gpurun calls (bash benchmarks/collect_r04_pairs.sh 1|2|3|4); the campaign summaries of calls 2 and 3 are
# appended to those of call 1 before they go to profiles/.  Call 3 needs the previous build of the library at
# benchmarks/micro/libbisip_hip_before.so (make -C <a checkout of the parent commit>/bisip_amd/csrc OUT=...).
case "$1" in
1)
  set -o pipefail
  out=gpurun_out
  mkdir -p $out; rm -f $out/progress.log
  python3 -m pytest tests -m gpu -q > $out/r04_gpu_tests_final.txt 2>&1; echo "pytest rc=$?" >> $out/r04_gpu_tests_final.txt; tail -3 $out/r04_gpu_tests_final.txt
  bash benchmarks/collect_profiles.sh bench > $out/collect_bench.log 2>&1
  python3 benchmarks/summarize_pmc.py $out profiles r04 > /dev/null && python3 bench.py > $out/bench_final.json 2> $out/bench_final.err; echo "bench rc=$?"
  bash benchmarks/collect_profiles.sh sweep > $out/collect_sweep.log 2>&1; echo "sweep group done"
  : > $out/fuzz_parity_summary.jsonl; : > $out/fuzz_valley_summary.jsonl; : > $out/fuzz_sampler_summary.jsonl; : > $out/fuzz_batch_summary.jsonl
  python3 benchmarks/fuzz_parity.py --cases 1500 --seed 46 --widen 3 2> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl
  python3 benchmarks/fuzz_parity.py --cases 4000 --seed 64 2>> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl
  python3 benchmarks/fuzz_parity.py --cases 2000 --seed 65 --widen 1.5 2>> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl
  echo "first parity seeds done"
  python3 benchmarks/fuzz_parity.py --cases 3000 --seed 311 --valley 2>> $out/fuzz.err | tail -1 >> $out/fuzz_valley_summary.jsonl
  python3 benchmarks/fuzz_sampler.py --cases 1500 --seed 25 2>> $out/fuzz.err | tail -1 >> $out/fuzz_sampler_summary.jsonl
  python3 benchmarks/fuzz_batch.py --cases 600 --seed 26 2>> $out/fuzz.err | tail -1 >> $out/fuzz_batch_summary.jsonl
  echo "first campaign set done"
  cut -c1-150 $out/fuzz_parity_summary.jsonl $out/fuzz_valley_summary.jsonl $out/fuzz_sampler_summary.jsonl $out/fuzz_batch_summary.jsonl
  cat $out/progress.log
  ;;
2)
  set -o pipefail
  out=gpurun_out
  mkdir -p $out; rm -f $out/progress.log
  bash benchmarks/collect_profiles.sh sampler > $out/collect_sampler.log 2>&1; echo "sampler group done"
  python3 benchmarks/fuzz_parity.py --cases 10000 --seed 66 2> $out/fuzz2.err | tail -1 >> $out/fuzz_parity_summary_2.jsonl; echo "parity 66 done"
  python3 benchmarks/fuzz_parity.py --cases 10000 --seed 67 --widen 1.5 2>> $out/fuzz2.err | tail -1 >> $out/fuzz_parity_summary_2.jsonl; echo "parity 67 done"
  python3 benchmarks/fuzz_parity.py --cases 4000 --seed 68 --widen 3 2>> $out/fuzz2.err | tail -1 >> $out/fuzz_parity_summary_2.jsonl; echo "parity 68 done"
  python3 benchmarks/fuzz_parity.py --cases 6000 --seed 312 --valley 2>> $out/fuzz2.err | tail -1 >> $out/fuzz_valley_summary_2.jsonl; echo "valley 312 done"
  python3 benchmarks/fuzz_sampler.py --cases 6000 --seed 27 2>> $out/fuzz2.err | tail -1 >> $out/fuzz_sampler_summary_2.jsonl; echo "sampler 27 done"
  python3 benchmarks/fuzz_batch.py --cases 2000 --seed 28 2>> $out/fuzz2.err | tail -1 >> $out/fuzz_batch_summary_2.jsonl; echo "batch 28 done"
  cut -c1-150 $out/fuzz_parity_summary_2.jsonl $out/fuzz_valley_summary_2.jsonl $out/fuzz_sampler_summary_2.jsonl $out/fuzz_batch_summary_2.jsonl
  cat $out/progress.log
  ;;
3)
  set -o pipefail
  out=gpurun_out
  mkdir -p $out; rm -f $out/progress.log
  for lib in benchmarks/micro/libbisip_hip_before.so bisip_amd/libbisip_hip.so benchmarks/micro/libbisip_hip_before.so bisip_amd/libbisip_hip.so; do
    python3 benchmarks/micro/ab_library.py $lib 2>> $out/ab.err | grep '^{' >> $out/micro_ab_paired_reciprocals.jsonl
  done
  cat $out/micro_ab_paired_reciprocals.jsonl | cut -c1-400
  python3 benchmarks/fuzz_parity.py --cases 10000 --seed 69 2> $out/fuzz3.err | tail -1 >> $out/fuzz_parity_summary_3.jsonl; echo "parity 69 done"
  python3 benchmarks/fuzz_parity.py --cases 10000 --seed 70 --widen 1.5 2>> $out/fuzz3.err | tail -1 >> $out/fuzz_parity_summary_3.jsonl; echo "parity 70 done"
  python3 benchmarks/fuzz_parity.py --cases 5000 --seed 71 --widen 3 2>> $out/fuzz3.err | tail -1 >> $out/fuzz_parity_summary_3.jsonl; echo "parity 71 done"
  python3 benchmarks/fuzz_parity.py --cases 6000 --seed 313 --valley 2>> $out/fuzz3.err | tail -1 >> $out/fuzz_valley_summary_3.jsonl; echo "valley 313 done"
  python3 benchmarks/fuzz_sampler.py --cases 6000 --seed 29 2>> $out/fuzz3.err | tail -1 >> $out/fuzz_sampler_summary_3.jsonl; echo "sampler 29 done"
  python3 benchmarks/fuzz_batch.py --cases 2000 --seed 30 2>> $out/fuzz3.err | tail -1 >> $out/fuzz_batch_summary_3.jsonl; echo "batch 30 done"
  cut -c1-150 $out/fuzz_parity_summary_3.jsonl $out/fuzz_valley_summary_3.jsonl $out/fuzz_sampler_summary_3.jsonl $out/fuzz_batch_summary_3.jsonl
  ;;
4)
  # the micro group's entries that run a changed kernel through the library
  out=gpurun_out; mkdir -p $out
  CFG5_ENV="A=1" timeout -k 5 300 bash benchmarks/micro/cfg5_pmc.sh > "$out/micro_cfg5_pmc.txt" 2>&1; echo "cfg5_pmc rc=$?"
  timeout -k 5 200 python3 benchmarks/micro/batch_logprob_rate.py 2>/dev/null > "$out/micro_batch_logprob_rate.txt"; echo "batch_logprob_rate rc=$?"
  timeout -k 5 300 python3 benchmarks/micro/persistent_crossover.py 2>/dev/null | grep "it/s" > "$out/micro_persistent_crossover.txt"; echo "persistent_crossover rc=$?"
  timeout -k 5 200 python3 benchmarks/micro/small_call_latency.py 2>/dev/null > "$out/micro_small_call_latency.jsonl"; echo "small_call_latency rc=$?"
  ;;
*) echo "usage: $0 1|2|3|4" >&2; exit 2;;
esac
