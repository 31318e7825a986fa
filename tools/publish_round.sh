#!/bin/bash
# Copy the summaries of one tools/collect_round.sh run (gpurun_out/<tag>/) into profiles/ (tracked): the judged evidence.
TAG=${1:-r05}
SRC=gpurun_out/$TAG
[ -d "$SRC" ] || { echo "no $SRC"; exit 1; }
for f in "$SRC"/bench_${TAG}*.json "$SRC"/rocprof_${TAG}_kernel_stats*.txt "$SRC"/pmc_${TAG}_hbm_bytes_*.txt "$SRC"/pmc_${TAG}_sq_counters_*.txt \
         "$SRC"/ab_prev_vs_${TAG}.txt "$SRC"/batch_ops_${TAG}.txt "$SRC"/kzg_timing_${TAG}.txt "$SRC"/evm_timing_${TAG}.txt "$SRC"/cached_host_scalars_${TAG}.txt \
         "$SRC"/crossover_${TAG}.jsonl "$SRC"/crossover_cpu_port_${TAG}.jsonl "$SRC"/hostptr_contexts_one_gpu_${TAG}.txt "$SRC"/sweep_sizes_${TAG}.jsonl \
         "$SRC"/hostptr_${TAG}.txt "$SRC"/table_${TAG}.jsonl "$SRC"/concurrent_callers_collection_${TAG}.txt; do
  [ -f "$f" ] && cp "$f" profiles/
done
[ -f "$SRC/batch_affine_host_collection_${TAG}.txt" ] && cp "$SRC/batch_affine_host_collection_${TAG}.txt" profiles/batch_affine_host_${TAG}.txt
[ -f "$SRC/pytest_gpu.log" ] && cp "$SRC/pytest_gpu.log" profiles/pytest_gpu_${TAG}.log
[ -f "$SRC/hbm_traffic_k_accum.json" ] && cp "$SRC/hbm_traffic_k_accum.json" profiles/hbm_traffic_k_accum.json
ls profiles | grep -c "$TAG"
