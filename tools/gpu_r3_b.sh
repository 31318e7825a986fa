#!/bin/bash
# Round 3, GPU call B: A/B of the round-2 library against the current one on the SAME box (merge / reduce stage times,
# kernel timelines), and the reduction launch-shape knobs.
set -u
OUT=$PWD/gpurun_out/r3b
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
for lib in r2 cur; do
  if [ $lib = r2 ]; then export CTT_MSM_HIP_LIB=$REPO/constantine_amd/libctt_msm_hip_r2.so; else unset CTT_MSM_HIP_LIB; fi
  timeout 300 python tools/sweep.py bls12_381_g1 20 c=16 -- bls12_381_g1 18 c=16 -- bls12_381_g1 16 c=13 -- bls12_381_g2 18 c=16 > "$OUT/ab_$lib.jsonl" 2> "$OUT/ab_$lib.err"
  for k in 16 20; do
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${lib}_$k" -o p -- python "$REPO/bench.py" --log2n $k --steps 10 --warmup 2 \
        --no-cpu-baseline --no-latency > /dev/null 2> "$OUT/prof_${lib}_$k.log" )
    DB=$(find "$OUT/prof_${lib}_$k" -name "*.db" | head -1)
    python tools/kernel_timeline.py "$DB" 2 > "$OUT/rocprof_${lib}_2pow$k.txt" 2>> "$OUT/prof_${lib}_$k.log"
    find "$OUT/prof_${lib}_$k" -name "*.db" -delete 2>/dev/null
  done
done
unset CTT_MSM_HIP_LIB
for nt in 64 128 256; do
  CTT_HIP_MSM_RED_NT=$nt timeout 300 python tools/sweep.py bls12_381_g1 20 quad_ratio=3,100 -- bls12_381_g1 16 quad_ratio=3,100 -- bls12_381_g1 18 quad_ratio=3,100 reduce_block=8,10 \
     > "$OUT/nt_$nt.jsonl" 2> "$OUT/nt_$nt.err"
done
grep -h ms_per_step "$OUT"/ab_*.jsonl | cut -c1-400
