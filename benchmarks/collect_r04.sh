set -o pipefail
out=gpurun_out
rm -f $out/progress.log
bash benchmarks/collect_profiles.sh bench > $out/collect_bench.log 2>&1
echo "== survey" >> $out/progress.log
python3 benchmarks/survey.py > $out/survey.jsonl 2> $out/survey.err
python3 benchmarks/survey.py --model PeltonColeCole >> $out/survey.jsonl 2>> $out/survey.err
python3 benchmarks/batch_setup.py > $out/batch_setup.jsonl 2> $out/batch_setup.err
echo "== sweep" >> $out/progress.log
python3 benchmarks/sweep.py > $out/sweep.jsonl 2> $out/sweep.err
echo "== fuzz" >> $out/progress.log
: > $out/fuzz_parity_summary.jsonl; : > $out/fuzz_valley_summary.jsonl; : > $out/fuzz_sampler_summary.jsonl; : > $out/fuzz_batch_summary.jsonl
python3 benchmarks/fuzz_parity.py --cases 1500 --seed 46 --widen 3 2> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl
python3 benchmarks/fuzz_parity.py --cases 4000 --seed 64 2>> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl
echo "== fuzz 2" >> $out/progress.log
python3 benchmarks/fuzz_parity.py --cases 2000 --seed 65 --widen 1.5 2>> $out/fuzz.err | tail -1 >> $out/fuzz_parity_summary.jsonl
python3 benchmarks/fuzz_parity.py --cases 3000 --seed 311 --valley 2>> $out/fuzz.err | tail -1 >> $out/fuzz_valley_summary.jsonl
echo "== fuzz 3" >> $out/progress.log
python3 benchmarks/fuzz_sampler.py --cases 1500 --seed 25 2>> $out/fuzz.err | tail -1 >> $out/fuzz_sampler_summary.jsonl
python3 benchmarks/fuzz_batch.py --cases 600 --seed 26 2>> $out/fuzz.err | tail -1 >> $out/fuzz_batch_summary.jsonl
echo "== done" >> $out/progress.log
tail -3 $out/fuzz_parity_summary.jsonl | cut -c1-200
