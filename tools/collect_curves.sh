#!/bin/bash
# Re-collects the lines of the curves whose device code goes through asm_pass.py (round 6, after the collection): the bench lines with their
# CPU / parity legs, the rocprofv3 kernel statistics and the SQ counters of their accumulate kernels.
#     bash tools/collect_curves.sh <tag>        -> gpurun_out/<tag>/
TAG=${1:-r06c}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export TMPDIR=/tmp
timeout 400 python bench.py --curve bn254_snarks_g1 --log2n 22 --steps 20 --warmup 3 > "$OUT/bench_r06_bn254_snarks_g1.json" 2>> "$OUT/bench.err"
timeout 400 python bench.py --curve pallas --steps 20 --warmup 3 > "$OUT/bench_r06_pallas.json" 2>> "$OUT/bench.err"
timeout 400 python bench.py --curve vesta --steps 20 --warmup 3 > "$OUT/bench_r06_vesta.json" 2>> "$OUT/bench.err"
timeout 400 python bench.py --curve bls12_381_g2 --steps 20 --warmup 3 > "$OUT/bench_r06_bls12_381_g2.json" 2>> "$OUT/bench.err"
for cfg in "bn254_snarks_g1 22" "pallas 20" "bls12_381_g2 20"; do
  set -- $cfg
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$1" -o p -- python "$REPO/bench.py" --curve $1 --log2n $2 --steps 10 --warmup 2 \
      --no-cpu-baseline --no-latency > /dev/null 2>> "$OUT/prof.log" )
  DB=$(find "$OUT/prof_$1" -name "*.db" | head -1)
  python tools/kernel_timeline.py "$DB" 2 > "$OUT/rocprof_r06_kernel_stats_$1_2pow$2.txt" 2>> "$OUT/prof.log"
  rm -rf "$OUT/prof_$1"
done
sq() {   # the SQ counters of k_accum (own passes, --kernel-trace only), exactly as tools/collect_round.sh collects them
  local tag=$1; shift 1
  ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
      --kernel-trace --output-format csv -d "$OUT/sq1_$tag" -o p -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-latency "$@" > "$OUT/sq1_$tag.json" 2> "$OUT/sq1_$tag.log" )
  ( cd /tmp && timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/sq2_$tag" -o p -- \
      python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-latency "$@" > /dev/null 2> "$OUT/sq2_$tag.log" )
  local madds=$(python -c "import json,sys; d=[json.loads(l) for l in open('$OUT/sq1_$tag.json') if l.startswith('{')][-1]['config']; print(d['windows']*d['pairs_per_gpu'])")
  { echo "# SQ / GRBM counters of k_accum, bench.py $*, rocprofv3 --pmc (two passes, --kernel-trace only); $madds mixed additions per launch (windows x pairs of the run's plan)"; python tools/sq_summary.py k_accum $madds "$OUT/sq1_$tag" "$OUT/sq2_$tag"; } > "$OUT/pmc_r06_sq_counters_k_accum_$tag.txt" 2>> "$OUT/prof.log"
  rm -rf "$OUT/sq1_$tag" "$OUT/sq2_$tag"
}
sq bn254_snarks_g1_2pow22 --curve bn254_snarks_g1 --log2n 22
sq pallas_2pow20 --curve pallas
sq bls12_381_g2_2pow20 --curve bls12_381_g2
for f in "$OUT"/bench_r06_*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["config"]["workload"], d["value"], d["unit"], "ms", d["ms_per_step"], "blocking", d.get("value_blocking"), "hostptr", d.get("value_hostptr"),
      "parity", d.get("parity_vs_oracle_on_sample"), d.get("parity_full_size_vs_discrete_logs"), "int_mad", d["roofline"].get("int_mad", {}).get("frac"))
PY
done
