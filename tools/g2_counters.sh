#!/bin/bash
# SQ / instruction-cache counters of the BLS12-381 G2 accumulate kernel, the in-tree library beside tools/libctt_msm_hip_prev.so, same box:
#     bash tools/g2_counters.sh <outdir>
OUT=$PWD/$1; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
for which in new prev; do
  if [ $which = prev ]; then export CTT_MSM_HIP_LIB=$REPO/tools/libctt_msm_hip_prev.so CTT_MSM_HIP_ALLOW_OLD_ABI=1; else unset CTT_MSM_HIP_LIB CTT_MSM_HIP_ALLOW_OLD_ABI; fi
  ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
      --kernel-trace --output-format csv -d "$OUT/a_$which" -o p -- python "$REPO/tools/sweep.py" bls12_381_g2 20 c=0 > "$OUT/a_$which.json" 2> "$OUT/a_$which.log" )
  ( cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_IFETCH SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD \
      --kernel-trace --output-format csv -d "$OUT/b_$which" -o p -- python "$REPO/tools/sweep.py" bls12_381_g2 20 c=0 > /dev/null 2> "$OUT/b_$which.log" )
  ( cd /tmp && timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE \
      --kernel-trace --output-format csv -d "$OUT/c_$which" -o p -- python "$REPO/tools/sweep.py" bls12_381_g2 20 c=0 > /dev/null 2> "$OUT/c_$which.log" )
  { echo "# k_accum<Fp2<FpU<BLS12-381>>> at 2^20 pairs (16 windows), library: $which; rocprofv3 --pmc, three passes, --kernel-trace only"; python tools/sq_summary.py k_accum 16777216 "$OUT/a_$which" "$OUT/b_$which" "$OUT/c_$which"; } > "$OUT/g2_counters_$which.txt" 2>&1
  rm -rf "$OUT/a_$which" "$OUT/b_$which" "$OUT/c_$which"
done
unset CTT_MSM_HIP_LIB CTT_MSM_HIP_ALLOW_OLD_ABI
tail -30 "$OUT/g2_counters_new.txt" "$OUT/g2_counters_prev.txt"
