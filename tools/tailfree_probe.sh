#!/bin/bash
# how many wave slots the accumulate grid leaves free for the previous MSM's tail (CTT_HIP_MSM_TAIL_MIN_FREE, default 16) and how much of an accumulation
# submit() may spend on that (CTT_HIP_MSM_TAIL_FREE_RATIO, default 0.5): ms per MSM with two in flight
#     bash tools/tailfree_probe.sh <curve> <log2n,...>
for rep in 1 2; do
  for cfg in "default:" "free0:CTT_HIP_MSM_TAIL_MIN_FREE=0" "free4:CTT_HIP_MSM_TAIL_MIN_FREE=4" "free64:CTT_HIP_MSM_TAIL_MIN_FREE=64" "ratio0:CTT_HIP_MSM_TAIL_FREE_RATIO=0" "ratio2:CTT_HIP_MSM_TAIL_FREE_RATIO=2"; do
    name=${cfg%%:*}; kv=${cfg#*:}
    ( [ -n "$kv" ] && export $kv; python tools/cu_mask_sweep.py $1 $2 3 -- "d2:" 2>/dev/null ) | MODE=$name python -c "
import json,sys,os
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print(os.environ['MODE'], d['curve'], '2^%d' % d['log2n'], 'ms/MSM', d['ms_per_msm_two_in_flight'], 'blocking', d['blocking_ms'], 'same', d['same_result'])"
  done
done
