# Closing campaigns of round 4 at the final kernel sources (GPU box; summary lines are APPENDED to the round's files).
set -o pipefail
out=gpurun_out
: > $out/closing_parity.jsonl; : > $out/closing_valley.jsonl; : > $out/closing_sampler.jsonl; : > $out/closing_batch.jsonl
python3 benchmarks/fuzz_parity.py --cases 10000 --seed 66 2> $out/closing.err | tail -1 >> $out/closing_parity.jsonl; echo "parity 66 done"
python3 benchmarks/fuzz_parity.py --cases 10000 --seed 67 --widen 1.5 2>> $out/closing.err | tail -1 >> $out/closing_parity.jsonl; echo "parity 67 done"
python3 benchmarks/fuzz_parity.py --cases 4000 --seed 68 --widen 3 2>> $out/closing.err | tail -1 >> $out/closing_parity.jsonl; echo "parity 68 done"
python3 benchmarks/fuzz_parity.py --cases 6000 --seed 312 --valley 2>> $out/closing.err | tail -1 >> $out/closing_valley.jsonl; echo "valley 312 done"
python3 benchmarks/fuzz_sampler.py --cases 6000 --seed 27 2>> $out/closing.err | tail -1 >> $out/closing_sampler.jsonl; echo "sampler 27 done"
python3 benchmarks/fuzz_batch.py --cases 2000 --seed 28 2>> $out/closing.err | tail -1 >> $out/closing_batch.jsonl; echo "batch 28 done"
cut -c1-160 $out/closing_parity.jsonl $out/closing_valley.jsonl $out/closing_sampler.jsonl $out/closing_batch.jsonl
