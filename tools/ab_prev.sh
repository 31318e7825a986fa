#!/bin/bash
# same-box A/B of the in-tree library against tools/libctt_msm_hip_prev.so (built from an earlier commit)
ARGS="$@"
for rep in $(seq 1 ${REPS:-3}); do
  for which in prev new; do
    # (the previous round's library has two in-flight slots: its loop keeps two MSMs in flight; the current one runs bench.py's own depth -- three up to 2^16 pairs)
    if [ $which = prev ]; then export CTT_MSM_HIP_LIB=$PWD/tools/libctt_msm_hip_prev.so CTT_MSM_HIP_ALLOW_OLD_ABI=1 CTT_BENCH_DEPTH=2; else unset CTT_MSM_HIP_LIB CTT_MSM_HIP_ALLOW_OLD_ABI CTT_BENCH_DEPTH; fi
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline $ARGS 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()})"
  done
done
