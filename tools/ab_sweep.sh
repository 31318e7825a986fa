#!/bin/bash
# same box, tools/libctt_msm_hip_prev.so beside the in-tree library through tools/sweep.py (no stage events in the timed loop):
#     bash tools/ab_sweep.sh <curve> <log2n> [<curve> <log2n> ...]        REPS=2 by default; OTHER_LIB=<path> compares another library (an experiment build) instead
args=()
while [ $# -ge 2 ]; do args+=("$1" "$2" "c=0" "--"); shift 2; done
for rep in $(seq 1 ${REPS:-2}); do
  for which in prev new; do
    if [ $which = prev ]; then export CTT_MSM_HIP_LIB=${OTHER_LIB:-$PWD/tools/libctt_msm_hip_prev.so} CTT_MSM_HIP_ALLOW_OLD_ABI=1; else unset CTT_MSM_HIP_LIB CTT_MSM_HIP_ALLOW_OLD_ABI; fi
    python tools/sweep.py "${args[@]}" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); st=d['stage_ms_blocking']
    print('$which' if not '${OTHER_LIB:-}' or '$which' == 'new' else 'other', d['curve'], '2^%d' % d['log2n'], 'ms/MSM two in flight', d['ms_per_step'], 'blocking', d['blocking_ms'], 'accumulate', st.get('accumulate'), 'merge', st.get('merge'), 'reduce', st.get('reduce'), 'crc', d['crc'])"
  done
done
