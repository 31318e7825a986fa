#!/bin/bash
# Round 3, GPU call K: why one more window bit does not pay for the accumulate kernel -- SQ counters and fabric traffic of k_accum
# for BN254 2^22 and BLS12-381 2^22 at c = 16 and c = 17 (same box; the window size is forced through $CTT_HIP_MSM_C).
#     gpurun --timeout 900 -- 'bash tools/gpu_r3_k.sh'
set -u
OUT=$PWD/gpurun_out/r3k
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
RES="$OUT/accum_window_bits_counters_r03.txt"
: > "$RES"
for cfg in "bn254_snarks_g1 22 254" "bls12_381_g1 22 255"; do
  set -- $cfg
  for c in 16 17; do
    W=$(( ($3 + 1 + c - 1) / c ))
    MADDS=$(( W * (1 << $2) ))
    tag=$1_2pow$2_c$c
    ( cd /tmp && CTT_HIP_MSM_C=$c timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace \
        --output-format csv -d "$OUT/sq_$tag" -o p -- python "$REPO/bench.py" --curve $1 --log2n $2 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > /dev/null 2> "$OUT/sq_$tag.log" )
    ( cd /tmp && CTT_HIP_MSM_C=$c timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace \
        --output-format csv -d "$OUT/fs_$tag" -o p -- python "$REPO/bench.py" --curve $1 --log2n $2 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > /dev/null 2> "$OUT/fs_$tag.log" )
    CTT_HIP_MSM_C=$c timeout 300 python bench.py --curve $1 --log2n $2 --steps 10 --warmup 2 --no-cpu-baseline --no-latency > "$OUT/bench_$tag.json" 2>> "$OUT/bench.err"
    {
      echo "== $1 2^$2, c = $c: $W windows, $MADDS mixed additions per launch"
      python tools/sq_summary.py k_accum $MADDS "$OUT/sq_$tag"
      python - "$OUT/fs_$tag" $MADDS <<'PY'
import csv, glob, os, sys
vals = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_accum" in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
            vals.append(float(row["Counter_Value"]))
if vals:
    b = 2 * 1024 * sum(vals) / len(vals)   # gfx950: FETCH_SIZE counts a 128-byte request as 64 bytes
    print("FETCH bytes per launch (2 x FETCH_SIZE x 1024) = %.3e = %.0f per mixed addition" % (b, b / float(sys.argv[2])))
PY
      python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("un-profiled: ms per MSM %.3f, accumulate launch %.3f ms, plan %s" % (d["ms_per_step"], d.get("roofline", {}).get("kernel_ms", float("nan")), d.get("config", {}).get("plan")))
PY
    } >> "$RES" 2>&1
    rm -rf "$OUT/sq_$tag" "$OUT/fs_$tag"
  done
done
cat "$RES"
