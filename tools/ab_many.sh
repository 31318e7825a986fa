#!/bin/bash
# same box: several experiment libraries beside the in-tree one through tools/sweep.py
#     LIBS="tag1 tag2" bash tools/ab_many.sh <curve> <log2n> [...]      (tools/libctt_msm_hip_<tag>.so; "shipped" = the in-tree library)
args=()
while [ $# -ge 2 ]; do args+=("$1" "$2" "c=0" "--"); shift 2; done
for rep in $(seq 1 ${REPS:-2}); do
  for tag in shipped $LIBS; do
    if [ $tag = shipped ]; then unset CTT_MSM_HIP_LIB; else export CTT_MSM_HIP_LIB=$PWD/tools/libctt_msm_hip_$tag.so; fi
    python tools/sweep.py "${args[@]}" 2>/dev/null | TAG=$tag python -c "
import json,sys,os
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); st=d['stage_ms_blocking']
    print(os.environ['TAG'], d['curve'], '2^%d' % d['log2n'], 'ms/MSM two in flight', d['ms_per_step'], 'blocking', d['blocking_ms'], 'accumulate', st.get('accumulate'), 'merge', st.get('merge'), 'reduce', st.get('reduce'), 'crc', d['crc'])"
  done
done
