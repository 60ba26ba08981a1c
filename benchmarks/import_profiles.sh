#!/bin/bash
# Copies what `collect_profiles.sh all` left under gpurun_out/ into profiles/<round>_* (the tracked
# evidence), summarises the PMC passes and regenerates the tables.  Run from the repo root:
#   bash benchmarks/import_profiles.sh r03
set -e -o pipefail
tag=${1:?round tag, e.g. r02}
src=gpurun_out
dst=profiles
newest() { ls -t "$1"/*"$2" 2>/dev/null | head -1; }   # earlier calls leave their PID-named files behind
if grep -q '^!!' "$src/progress.log" 2>/dev/null; then
  echo "gpurun_out/progress.log records failed steps; look at them (and remove the log) before importing:" >&2
  grep '^!!' "$src/progress.log" >&2
  exit 1
fi
cpy() { [ -s "$1" ] && cp "$1" "$2" || echo "missing or empty: $1" >&2; }
cpy $src/bench.json                    $dst/${tag}_bench.json
cpy $src/bench_under_rocprof.json      $dst/${tag}_bench_under_rocprof.json
cpy "$(newest $src/prof_bench/runc kernel_stats.csv)"   $dst/${tag}_bench_kernel_stats.csv
cpy $src/sweep.jsonl                   $dst/${tag}_sweep.jsonl
cpy "$(newest $src/prof_sweep/runc kernel_stats.csv)"   $dst/${tag}_sweep_kernel_stats.csv
cpy $src/host_path.jsonl               $dst/${tag}_host_path.jsonl
cpy $src/ingest.json                   $dst/${tag}_ingest.json
cpy $src/batch_setup.jsonl             $dst/${tag}_batch_setup.jsonl
cpy $src/survey.jsonl                  $dst/${tag}_survey.jsonl
cpy $src/sampler_bench.jsonl           $dst/${tag}_sampler_bench.jsonl
cpy $src/batch_models.jsonl            $dst/${tag}_batch_models.jsonl
cpy $src/soak.json                     $dst/${tag}_soak.json
cpy "$(newest $src/prof_sampler/runc kernel_stats.csv)" $dst/${tag}_sampler_kernel_stats.csv
for f in cfg4_fused_device_chain cfg4_sharded_python cfg4_sharded_rccl-own cfg4_sharded_rccl \
         cfg5_device_chain cfg5_device_chain_launches cfg5_host_chain; do
  cpy $src/$f.json $dst/${tag}_$f.json
done
cpy "$(newest $src/prof_cfg5/runc kernel_stats.csv)"    $dst/${tag}_cfg5_kernel_stats.csv
cpy "$(newest $src/prof_batch_models/runc kernel_stats.csv)" $dst/${tag}_batch_models_kernel_stats.csv
for m in issue_latency row_latency half_step_phases forward_rows_variants grid_barrier xcd_barrier post_run_stall cfg5_pmc batch_pd_pmc batch_logprob_rate select_vs_sort model_percentile_single persistent_crossover collapsed_r3 exp2_variants rcp_accuracy fp64_stream_ceiling random_lines; do
  cpy $src/micro_$m.txt $dst/${tag}_micro_$m.txt
done
cpy $src/micro_small_call_latency.jsonl $dst/${tag}_micro_small_call_latency.jsonl
cpy $src/micro_group_sampler.jsonl $dst/${tag}_micro_group_sampler.jsonl
cpy $src/micro_ab_big_ensemble_packed.jsonl $dst/${tag}_micro_ab_big_ensemble_packed.jsonl
cpy $src/guard_overhead_cfg4.jsonl $dst/${tag}_guard_overhead_cfg4.jsonl
cpy $src/${tag}_gpu_tests_final.txt $dst/${tag}_gpu_tests_final.txt
for k in parity valley sampler batch group; do cpy $src/fuzz_${k}_summary.jsonl $dst/${tag}_fuzz_${k}_summary.jsonl; done
# the headline line of the round is the one taken AFTER the counters were summarised on the box (collect_final.sh bench)
[ -s $src/bench_final.json ] && cp $src/bench_final.json $dst/${tag}_bench.json
cpy $src/valley_rows.jsonl $dst/${tag}_valley_rows.jsonl
python3 benchmarks/summarize_pmc.py $src $dst $tag
python3 benchmarks/make_tables.py
