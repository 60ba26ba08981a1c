#!/bin/bash
# Collects the measurements kept under profiles/ (one MI355X).  Run from the repo root on the
# GPU box:  bash benchmarks/collect_profiles.sh [bench|sweep|sampler|micro|all]
# Everything is written under gpurun_out/ (scratch); copy what should be judged into profiles/.
# A step that fails (or a grep that matches nothing) is logged and the collection goes on: an `all` run
# must not lose the hours behind it to one optional measurement.
set -o pipefail
what=${1:-all}
out=$PWD/gpurun_out
mkdir -p "$out"
scratch=$(mktemp -d /tmp/bisip_prof.XXXXXX)   # full traces stay here: only summaries go to gpurun_out
keep() {   # keep <trace dir> <name>: copy the small CSVs of a rocprofv3 run
  mkdir -p "$out/$2/runc"
  find "$1" \( -name '*kernel_stats.csv' -o -name '*domain_stats.csv' -o -name '*agent_info.csv' -o -name '*counter_collection.csv' \) -exec cp {} "$out/$2/runc/" \;
}
export TMPDIR=/tmp
repo=$PWD
trap 'rm -rf "$scratch"' EXIT
step() {   # step <name> <command...>: run, log a failure, carry on
  local name=$1; shift
  "$@"
  local rc=$?
  # to stderr and the log only: every caller redirects stdout into the evidence file itself
  if [ $rc -ne 0 ]; then echo "!! step $name failed (status $rc)" >&2; echo "!! step $name failed (status $rc)" >> "$out/progress.log"; fi
}

if [ "$what" = bench ] || [ "$what" = all ]; then
  echo "== bench.py" | tee -a "$out/progress.log"
  step bench python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
  echo "== bench.py under rocprofv3 --kernel-trace --stats" | tee -a "$out/progress.log"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$scratch/prof_bench" -- \
      python3 "$repo/bench.py" --no-cpu-baseline --no-variants > "$out/bench_under_rocprof.json" 2> "$out/prof_bench.err") || echo "!! a rocprofv3 step failed" | tee -a "$out/progress.log"
  keep "$scratch/prof_bench" prof_bench
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== bench.py --pmc $c" | tee -a "$out/progress.log"
    (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$scratch/pmc_$c" -- \
        python3 "$repo/bench.py" --no-cpu-baseline --no-variants --steps 20 --prime-seconds 0.1 > "$out/pmc_$c.json" 2> "$out/pmc_$c.err") || echo "!! a rocprofv3 step failed" | tee -a "$out/progress.log"
    keep "$scratch/pmc_$c" pmc_$c
  done
  # one pass, eight SQ counters: the instruction counts behind roofline_valu (all VALU, and the quarter-rate fp64
  # transcendentals among them) and where the waves' cycles go (parked at a wait, stalled at issue, issuing)
  echo "== bench.py --pmc-pass --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 + wait / active cycles" | tee -a "$out/progress.log"
  (cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SMEM --output-format csv -d "$scratch/pmc_valu" -- \
      python3 "$repo/bench.py" --pmc-pass > "$out/pmc_valu.json" 2> "$out/pmc_valu.err") || echo "!! a rocprofv3 step failed" | tee -a "$out/progress.log"
  keep "$scratch/pmc_valu" pmc_valu
fi
if [ "$what" = sweep ] || [ "$what" = all ]; then
  echo "== sweep" | tee -a "$out/progress.log"
  step sweep python3 benchmarks/sweep.py > "$out/sweep.jsonl" 2> "$out/sweep.err"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$scratch/prof_sweep" -- \
      python3 "$repo/benchmarks/sweep.py" > "$out/sweep_under_rocprof.jsonl" 2> "$out/prof_sweep.err") || echo "!! a rocprofv3 step failed" | tee -a "$out/progress.log"
  keep "$scratch/prof_sweep" prof_sweep
  step host_path python3 benchmarks/host_path.py > "$out/host_path.jsonl" 2> "$out/host_path.err"
  step ingest python3 benchmarks/ingest.py > "$out/ingest.json" 2> "$out/ingest.err"
  step batch_setup python3 benchmarks/batch_setup.py > "$out/batch_setup.jsonl" 2> "$out/batch_setup.err"
  step survey python3 benchmarks/survey.py > "$out/survey.jsonl" 2> "$out/survey.err"
  step survey python3 benchmarks/survey.py --model PeltonColeCole >> "$out/survey.jsonl" 2>> "$out/survey.err"
fi
if [ "$what" = sampler ] || [ "$what" = all ]; then
  echo "== samplers" | tee -a "$out/progress.log"
  step sampler_bench python3 benchmarks/sampler_bench.py > "$out/sampler_bench.jsonl" 2> "$out/sampler.err"
  step cfg4_fused bash -c 'python3 benchmarks/cfg4_sampler.py --steps 200 --fused --chain device | grep "^{"' > "$out/cfg4_fused_device_chain.json" 2>> "$out/cfg4.err"
  for loop in rccl rccl-own python; do
    step cfg4_$loop bash -c "python3 benchmarks/cfg4_sampler.py --steps 200 --chain device --loop $loop | grep '^{'" > "$out/cfg4_sharded_$loop.json" 2>> "$out/cfg4.err"
  done
  # cfg5: 20000 iterations (0.3-0.6 s), so that a 20-30 ms start-up hiccup of the device does not decide the number
  step cfg5_batch python3 benchmarks/cfg5_batch.py --chain device --steps 500 --thin-by 40 > "$out/cfg5_device_chain.json" 2> "$out/cfg5.err"
  step cfg5_batch python3 benchmarks/cfg5_batch.py --chain device --steps 500 --thin-by 40 --no-persistent > "$out/cfg5_device_chain_launches.json" 2>> "$out/cfg5.err"
  step cfg5_batch python3 benchmarks/cfg5_batch.py --steps 100 --thin-by 10 > "$out/cfg5_host_chain.json" 2>> "$out/cfg5.err"
  step batch_models python3 benchmarks/batch_models.py > "$out/batch_models.jsonl" 2>> "$out/cfg5.err"
  step soak python3 benchmarks/soak.py > "$out/soak.json" 2>> "$out/cfg5.err"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$scratch/prof_sampler" -- \
      python3 "$repo/benchmarks/sampler_bench.py" > "$out/sampler_under_rocprof.jsonl" 2> "$out/prof_sampler.err") || echo "!! a rocprofv3 step failed" | tee -a "$out/progress.log"
  keep "$scratch/prof_sampler" prof_sampler
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$scratch/prof_cfg5" -- \
      python3 "$repo/benchmarks/cfg5_batch.py" --chain device --steps 100 --thin-by 10 > /dev/null 2> "$out/prof_cfg5.err") || echo "!! a rocprofv3 step failed" | tee -a "$out/progress.log"
  keep "$scratch/prof_cfg5" prof_cfg5
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$scratch/prof_batch_models" -- \
      python3 "$repo/benchmarks/batch_models.py" --only Polynomial > /dev/null 2> "$out/prof_batch_models.err") || echo "!! a rocprofv3 step failed" | tee -a "$out/progress.log"
  keep "$scratch/prof_batch_models" prof_batch_models
fi
if [ "$what" = micro ] || [ "$what" = all ]; then
  echo "== micro-benchmarks" | tee -a "$out/progress.log"
  for m in issue_latency row_latency half_step_phases forward_rows_variants grid_barrier xcd_barrier; do
    [ -x benchmarks/micro/$m ] && step $m timeout -k 5 200 benchmarks/micro/$m > "$out/micro_$m.txt" 2>&1
  done
  for m in collapsed_r3 exp2_variants rcp_accuracy; do
    [ -x benchmarks/micro/$m ] && step $m timeout -k 5 200 benchmarks/micro/$m > "$out/micro_$m.txt" 2>&1
  done
  step post_run_stall bash -c 'python3 benchmarks/micro/post_run_stall.py kernel 2>/dev/null | grep after' > "$out/micro_post_run_stall.txt"
  step upload_cost bash -c 'python3 benchmarks/micro/upload_cost.py plain 2>/dev/null | grep "upload ms"' >> "$out/micro_post_run_stall.txt"
  CFG5_ENV="A=1" step cfg5_pmc bash benchmarks/micro/cfg5_pmc.sh > "$out/micro_cfg5_pmc.txt" 2>&1
  step batch_pd_pmc bash benchmarks/micro/batch_pd_pmc.sh > "$out/micro_batch_pd_pmc.txt" 2>&1
  step batch_logprob_rate python3 benchmarks/micro/batch_logprob_rate.py 2>/dev/null > "$out/micro_batch_logprob_rate.txt"
  step model_percentile_single python3 benchmarks/micro/model_percentile_single.py 2>/dev/null > "$out/micro_model_percentile_single.txt"
  step select_vs_sort python3 benchmarks/micro/select_vs_sort.py 2>/dev/null > "$out/micro_select_vs_sort.txt"
  step select_long_columns python3 benchmarks/micro/select_long_columns.py 2>/dev/null >> "$out/micro_select_vs_sort.txt"
  step persistent_crossover bash -c 'python3 benchmarks/micro/persistent_crossover.py 2>/dev/null | grep "it/s"' > "$out/micro_persistent_crossover.txt"
  step small_call_latency python3 benchmarks/micro/small_call_latency.py 2>/dev/null > "$out/micro_small_call_latency.jsonl"
  step valley_rows python3 benchmarks/valley_rows.py > "$out/valley_rows.jsonl" 2> "$out/valley_rows.err"
fi
echo "== done" | tee -a "$out/progress.log"
