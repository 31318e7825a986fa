#!/bin/bash
# Collect one round's measured evidence on the GPU box (run through gpurun from the repo root):
#     gpurun --timeout 1500 -- 'bash tools/collect_round.sh r01'
# Writes everything under gpurun_out/<tag>/; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD

# ONLY=hbm: just the counter passes behind `roofline.traffic` (after a change to the kernel sources: bench.py refuses a figure measured on other sources)
[ "${ONLY:-}" = hbm ] || { timeout 600 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; }

# HBM traffic of every config the bench prints a roofline for: one counter per pass, csv output, kernel-trace only
hbm() {  # key, bench args...
  local key=$1; shift
  local tag=$(echo "$key" | tr '^' 'p')
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_${c}_$tag" -o p -- python "$REPO/bench.py" \
        --steps 3 --warmup 1 --no-cpu-baseline --no-latency "$@" > /dev/null 2> "$OUT/pmc_${c}_$tag.log" )
  done
  python tools/pmc_summary.py "$OUT/pmc_FETCH_SIZE_$tag" "$OUT/pmc_WRITE_SIZE_$tag" "$key" "$OUT/hbm_traffic_k_accum.json" \
      "$OUT/pmc_${TAG}_hbm_bytes_$tag.txt" > /dev/null 2>> "$OUT/prof.log"
  rm -rf "$OUT/pmc_FETCH_SIZE_$tag" "$OUT/pmc_WRITE_SIZE_$tag"
}
hbm "bls12_381_g1_2^20"
hbm "bls12_381_g1_2^22" --log2n 22
hbm "bls12_381_g1_2^24" --log2n 24
hbm "bn254_snarks_g1_2^22" --curve bn254_snarks_g1 --log2n 22
hbm "pallas_2^20" --curve pallas
hbm "bls12_381_g2_2^20" --curve bls12_381_g2
hbm "vesta_2^20" --curve vesta
for k in 16 17 18 19; do hbm "bls12_381_g1_2^$k" --log2n $k; done
# the bench lines below read their `roofline.traffic` from this run's passes
cp "$OUT/hbm_traffic_k_accum.json" profiles/hbm_traffic_k_accum.json
[ "${ONLY:-}" = hbm ] && { cat "$OUT/hbm_traffic_k_accum.json"; exit 0; }

# headline line, un-profiled
timeout 600 python bench.py > "$OUT/bench_$TAG.json" 2> "$OUT/bench.err"

# kernel trace + stats (rocpd database, summarised by tools/kernel_timeline.py)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o p -- python "$REPO/bench.py" --steps 50 --warmup 5 \
    --no-cpu-baseline --no-latency > "$OUT/bench_${TAG}_under_rocprof.json" 2>> "$OUT/prof.log" )
DB=$(find "$OUT/prof" -name "*.db" | head -1)
python tools/kernel_timeline.py "$DB" 5 > "$OUT/rocprof_${TAG}_kernel_stats.txt" 2>> "$OUT/prof.log"   # (5: the warm-up MSMs are left out of the averages; the same steps / warm-up as the un-profiled line above, so that the two k_accum averages are comparable)


# SQ counters of the accumulate kernel (own passes, kernel-trace only) for the headline and the 254/255-bit fields,
# and the kernel statistics of those configs
sq() {  # tag, bench args...   (mixed additions per launch = windows x pairs of the plan the run itself reports)
  local tag=$1; shift 1
  ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
      --kernel-trace --output-format csv -d "$OUT/sq1_$tag" -o p -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-latency "$@" > "$OUT/sq1_$tag.json" 2> "$OUT/sq1_$tag.log" )
  ( cd /tmp && timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/sq2_$tag" -o p -- \
      python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-latency "$@" > /dev/null 2> "$OUT/sq2_$tag.log" )
  local madds=$(python -c "import json,sys; d=[json.loads(l) for l in open('$OUT/sq1_$tag.json') if l.startswith('{')][-1]['config']; print(d['windows']*d['pairs_per_gpu'])")
  { echo "# SQ / GRBM counters of k_accum, bench.py $*, rocprofv3 --pmc (two passes, --kernel-trace only); $madds mixed additions per launch (windows x pairs of the run's plan)"; python tools/sq_summary.py k_accum $madds "$OUT/sq1_$tag" "$OUT/sq2_$tag"; } > "$OUT/pmc_${TAG}_sq_counters_k_accum_$tag.txt" 2>> "$OUT/prof.log"
  rm -rf "$OUT/sq1_$tag" "$OUT/sq2_$tag"
}
sq bls12_381_g1_2pow20
sq bn254_snarks_g1_2pow22 --curve bn254_snarks_g1 --log2n 22
sq pallas_2pow20 --curve pallas
sq bls12_381_g2_2pow20 --curve bls12_381_g2
sq bls12_381_g1_2pow16 --log2n 16
sq bls12_381_g1_2pow24 --log2n 24
for cfg in "bn254_snarks_g1 22" "pallas 20" "bls12_381_g2 20" "bls12_381_g1 16" "bls12_381_g1 17" "bls12_381_g1 18"; do
  set -- $cfg
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$1" -o p -- python "$REPO/bench.py" --curve $1 --log2n $2 --steps 10 --warmup 2 \
      --no-cpu-baseline --no-latency > /dev/null 2>> "$OUT/prof.log" )
  DB=$(find "$OUT/prof_$1" -name "*.db" | head -1)
  python tools/kernel_timeline.py "$DB" 2 > "$OUT/rocprof_${TAG}_kernel_stats_$1_2pow$2.txt" 2>> "$OUT/prof.log"
  find "$OUT/prof_$1" -name "*.db" -delete 2>/dev/null
done

# the other BASELINE configs and the size sweep
# (the lines of the other BASELINE configs keep their CPU / parity leg: `parity_vs_oracle_on_sample` and `parity_full_size_vs_discrete_logs`
# are true / false in every configs[] line, not null -- round-5 review)
timeout 400 python bench.py --curve bn254_snarks_g1 --log2n 22 --steps 20 --warmup 3 > "$OUT/bench_${TAG}_bn254_snarks_g1.json" 2>> "$OUT/bench.err"
timeout 400 python bench.py --curve pallas --steps 20 --warmup 3 > "$OUT/bench_${TAG}_pallas.json" 2>> "$OUT/bench.err"
timeout 400 python bench.py --curve vesta --steps 20 --warmup 3 > "$OUT/bench_${TAG}_vesta.json" 2>> "$OUT/bench.err"
timeout 400 python bench.py --curve bls12_381_g2 --steps 20 --warmup 3 > "$OUT/bench_${TAG}_bls12_381_g2.json" 2>> "$OUT/bench.err"
for k in 16 17 18 19 22 24; do
  timeout 300 python bench.py --log2n $k --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_${TAG}_bls12_381_g1_2pow$k.json" 2>> "$OUT/bench.err"
done
timeout 300 python tools/bench_batch_ops.py > "$OUT/batch_ops_$TAG.txt" 2>> "$OUT/bench.err"
timeout 300 python tools/bench_batch_ops_host.py > "$OUT/batch_affine_host_collection_$TAG.txt" 2>> "$OUT/bench.err"
timeout 300 python bench.py --cpu-only --log2n 10 > "$OUT/bench_${TAG}_cpu_only_2pow10.json" 2>> "$OUT/bench.err"     # BASELINE configs[0]
timeout 300 python tools/bench_kzg.py > "$OUT/kzg_timing_$TAG.txt" 2>> "$OUT/bench.err"
timeout 300 python tools/bench_evm.py > "$OUT/evm_timing_$TAG.txt" 2>> "$OUT/bench.err"
timeout 400 python tools/bench_cached_host_scalars.py > "$OUT/cached_host_scalars_$TAG.txt" 2>> "$OUT/bench.err"
timeout 400 python tools/bench_crossover.py > "$OUT/crossover_$TAG.jsonl" 2>> "$OUT/bench.err"
for k in 6 8 12 14; do timeout 300 python bench.py --cpu-only --log2n $k >> "$OUT/crossover_cpu_port_$TAG.jsonl" 2>> "$OUT/bench.err"; done   # the CPU port's side
timeout 300 python tools/bench_contexts.py > "$OUT/hostptr_contexts_one_gpu_$TAG.txt" 2>> "$OUT/bench.err"
timeout 300 python tools/sweep.py bls12_381_g1 6 c=0 -- bls12_381_g1 8 c=0 -- bls12_381_g1 10 c=0 -- bls12_381_g1 12 c=0 -- bls12_381_g1 14 c=0 -- bls12_381_g1 16 c=0 -- bls12_381_g1 17 c=0 -- bls12_381_g1 18 c=0 -- bls12_381_g1 19 c=0 \
    -- bls12_381_g1 20 c=0 -- bls12_381_g1 22 c=0 -- bls12_381_g1 24 c=0 -- bls12_381_g2 18 c=0 -- bls12_381_g2 20 c=0 -- bn254_snarks_g1 16 c=0 -- bn254_snarks_g1 20 c=0,16,17 -- bn254_snarks_g1 22 c=0 \
    -- pallas 20 c=0 -- vesta 20 c=0 -- bn254_snarks_g2 18 c=0 > "$OUT/sweep_sizes_$TAG.jsonl" 2>> "$OUT/bench.err"
timeout 300 python tools/bench_hostptr.py > "$OUT/hostptr_$TAG.txt" 2>> "$OUT/bench.err"
timeout 300 python tools/bench_threads.py 16 40 > "$OUT/concurrent_callers_collection_$TAG.txt" 2>> "$OUT/bench.err"
# cached bases with a window table next to the plain records (same box, same inputs): ms per pipelined step + stage times
{
  timeout 300 python tools/bench_table.py bls12_381_g1 20 0 19 21
  timeout 300 python tools/bench_table.py bls12_381_g1 18 0
  timeout 300 python tools/bench_table.py bls12_381_g1 16 0
  timeout 300 python tools/bench_table.py bls12_381_g1 12 0
  timeout 300 python tools/bench_table.py bn254_snarks_g1 22 0
  timeout 300 python tools/bench_table.py pallas 20 0
  timeout 300 python tools/bench_table.py bls12_381_g2 18 0
} 2>> "$OUT/bench.err" | grep '^{' > "$OUT/table_$TAG.jsonl"
# the N-GPU line's code path end to end on this one GPU: two ranks sharing device 0 with a gloo exchange (a plumbing check, not a scaling point)
timeout 300 python bench.py --gpus 2 --all-ranks-on-device 0 --backend gloo --steps 10 --warmup 2 > "$OUT/bench_${TAG}_2ranks_one_gpu_gloo.json" 2>> "$OUT/bench.err"
# ... and the driver's 8-rank form with every leg of the line (strong 2^20, weak, configs[3] 2^24 in total, strong_bound, hostptr_sharded_ms over 8 contexts): plumbing only
timeout 600 python bench.py --gpus 8 --all-ranks-on-device 0 --backend gloo --steps 3 --warmup 1 > "$OUT/bench_${TAG}_8ranks_one_gpu_gloo.json" 2>> "$OUT/bench.err"
# same box, the previous round's library next to this one (tools/libctt_msm_hip_prev.so, built from the previous round's commit)
if [ -f tools/libctt_msm_hip_prev.so ]; then
  { for a in "--log2n 16" "--log2n 17" "--log2n 18" "--log2n 19" "" "--curve bn254_snarks_g1 --log2n 20" "--curve bn254_snarks_g1 --log2n 22" "--curve pallas" "--curve bls12_381_g2 --log2n 18"; do
      echo "== bench.py $a (value M/s, ms per MSM with two in flight, stage times of the timed loop) =="; REPS=2 bash tools/ab_prev.sh --no-latency $a; done; } > "$OUT/ab_prev_vs_${TAG}.txt" 2>> "$OUT/bench.err"
fi
find "$OUT/prof" -name "*.db" -delete 2>/dev/null   # the rocpd databases are large; the summaries are what is kept
tail -3 "$OUT/pytest_gpu.log"; cat "$OUT/bench_$TAG.json"
