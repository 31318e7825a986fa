#!/bin/bash
# accumulate(i+1) waiting for the WHOLE tail of MSM i (CTT_HIP_MSM_TAIL_MIN_FREE huge) against the default (the narrow passes run beside it in the wave slots its grid leaves free)
#     bash tools/tailwait_probe.sh <curve> <log2n,...>
for rep in 1 2; do
  for mode in default full_wait; do
    if [ $mode = full_wait ]; then export CTT_HIP_MSM_TAIL_MIN_FREE=1000000; else unset CTT_HIP_MSM_TAIL_MIN_FREE; fi
    python tools/cu_mask_sweep.py $1 $2 3 -- "d2:" 2>/dev/null | MODE=$mode python -c "
import json,sys,os
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print(os.environ['MODE'], d['curve'], '2^%d' % d['log2n'], 'in flight', d['in_flight'], 'ms/MSM', d['ms_per_msm_two_in_flight'], 'blocking', d['blocking_ms'], 'same', d['same_result'])"
  done
done
