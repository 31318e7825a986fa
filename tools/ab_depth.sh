#!/bin/bash
# same box: experiment libraries beside the in-tree one through tools/cu_mask_sweep.py with two and three MSMs in flight
#     LIBS="tag1 tag2" bash tools/ab_depth.sh <curve> <log2n,log2n,...>
for rep in $(seq 1 ${REPS:-2}); do
  for tag in shipped $LIBS; do
    if [ $tag = shipped ]; then unset CTT_MSM_HIP_LIB; else export CTT_MSM_HIP_LIB=$PWD/tools/libctt_msm_hip_$tag.so; fi
    python tools/cu_mask_sweep.py $1 $2 3 -- "d2:" "d3:;depth=3" 2>/dev/null | TAG=$tag python -c "
import json,sys,os
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print(os.environ['TAG'], d['curve'], '2^%d' % d['log2n'], 'in flight', d['in_flight'], 'ms/MSM', d['ms_per_msm_two_in_flight'], 'blocking', d['blocking_ms'], 'same', d['same_result'])"
  done
done
