# Round 4, final kernel sources: GPU tests, bench + rocprof + PMC (counters summarised on the box so that the
# bench line taken afterwards quotes them), then every campaign seed of the round again.
set -o pipefail
out=gpurun_out
rm -f $out/progress.log
if [ -z "$ONLY_MORE" ]; then
python3 -m pytest tests -m gpu -q > $out/r04_gpu_tests_final.txt 2>&1; echo "pytest rc=$?" >> $out/r04_gpu_tests_final.txt; tail -3 $out/r04_gpu_tests_final.txt
bash benchmarks/collect_profiles.sh bench > $out/collect_bench.log 2>&1
python3 benchmarks/summarize_pmc.py $out profiles r04 > /dev/null && python3 bench.py > $out/bench_final.json 2> $out/bench_final.err; echo "bench rc=$?"
python3 benchmarks/sweep.py > $out/sweep.jsonl 2> $out/sweep.err
: > $out/fuzz_parity_summary.jsonl; : > $out/fuzz_valley_summary.jsonl; : > $out/fuzz_sampler_summary.jsonl; : > $out/fuzz_batch_summary.jsonl
python3 benchmarks/fuzz_parity.py --cases 1500 --seed 46 --widen 3 2> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl
python3 benchmarks/fuzz_parity.py --cases 4000 --seed 64 2>> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl
python3 benchmarks/fuzz_parity.py --cases 2000 --seed 65 --widen 1.5 2>> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl
echo "first parity seeds done"
python3 benchmarks/fuzz_parity.py --cases 3000 --seed 311 --valley 2>> $out/fuzz.err | tail -1 >> $out/fuzz_valley_summary.jsonl
python3 benchmarks/fuzz_sampler.py --cases 1500 --seed 25 2>> $out/fuzz.err | tail -1 >> $out/fuzz_sampler_summary.jsonl
python3 benchmarks/fuzz_batch.py --cases 600 --seed 26 2>> $out/fuzz.err | tail -1 >> $out/fuzz_batch_summary.jsonl
echo "first campaign set done"
python3 benchmarks/fuzz_parity.py --cases 10000 --seed 66 2>> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl; echo "parity 66 done"
python3 benchmarks/fuzz_parity.py --cases 10000 --seed 67 --widen 1.5 2>> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl; echo "parity 67 done"
python3 benchmarks/fuzz_parity.py --cases 4000 --seed 68 --widen 3 2>> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl; echo "parity 68 done"
python3 benchmarks/fuzz_parity.py --cases 6000 --seed 312 --valley 2>> $out/fuzz.err | tail -1 >> $out/fuzz_valley_summary.jsonl; echo "valley 312 done"
python3 benchmarks/fuzz_sampler.py --cases 6000 --seed 27 2>> $out/fuzz.err | tail -1 >> $out/fuzz_sampler_summary.jsonl; echo "sampler 27 done"
python3 benchmarks/fuzz_batch.py --cases 2000 --seed 28 2>> $out/fuzz.err | tail -1 >> $out/fuzz_batch_summary.jsonl; echo "batch 28 done"
cut -c1-150 $out/fuzz_parity_summary.jsonl $out/fuzz_valley_summary.jsonl $out/fuzz_sampler_summary.jsonl $out/fuzz_batch_summary.jsonl
fi
# MORE=1: further seeds at the same sources, appended to the summaries (ONLY_MORE=1: nothing but these)
if [ -n "$MORE" ]; then
python3 benchmarks/fuzz_parity.py --cases 10000 --seed 69 2>> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl; echo "parity 69 done"
python3 benchmarks/fuzz_parity.py --cases 10000 --seed 70 --widen 1.5 2>> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl; echo "parity 70 done"
python3 benchmarks/fuzz_parity.py --cases 5000 --seed 71 --widen 3 2>> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl; echo "parity 71 done"
python3 benchmarks/fuzz_parity.py --cases 6000 --seed 313 --valley 2>> $out/fuzz.err | tail -1 >> $out/fuzz_valley_summary.jsonl; echo "valley 313 done"
python3 benchmarks/fuzz_sampler.py --cases 6000 --seed 29 2>> $out/fuzz.err | tail -1 >> $out/fuzz_sampler_summary.jsonl; echo "sampler 29 done"
python3 benchmarks/fuzz_batch.py --cases 2000 --seed 30 2>> $out/fuzz.err | tail -1 >> $out/fuzz_batch_summary.jsonl; echo "batch 30 done"
cut -c1-150 $out/fuzz_parity_summary.jsonl $out/fuzz_valley_summary.jsonl $out/fuzz_sampler_summary.jsonl $out/fuzz_batch_summary.jsonl
fi
