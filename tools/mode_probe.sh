#!/bin/bash
# the two modes of the 9-limb accumulate kernel at 2^22 pairs (EXPERIMENTS R6.8): ms per MSM with two in flight under perturbations of what runs beside / before it
#     bash tools/mode_probe.sh <curve> [log2n=22]
CURVE=$1; K=${2:-22}
run() {   # label, env..., -- config
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python tools/cu_mask_sweep.py $CURVE $K 3 -- "$@" 2>/dev/null | LABEL="$label" python -c "
import json,sys,os
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print(os.environ['LABEL'], d['curve'], '2^%d' % d['log2n'], d['config'], 'in flight', d['in_flight'], 'ms/MSM', d['ms_per_msm_two_in_flight'], d['runs'], 'blocking', d['blocking_ms'], 'accumulate(blocking)', d['stage_ms_blocking'].get('accumulate'))"
}
for rep in 1 2; do
  run default X=1 -- "d2:" "d3:;depth=3" "d1:;depth=1"
  run wide_early_off CTT_HIP_MSM_WIDE_EARLY=0 -- "d2:"
  run early_tail_off X=1 -- "et0:;early_tail=0"
  run early_tail_always X=1 -- "et2:;early_tail=2"
  run hw_queues_4 GPU_MAX_HW_QUEUES=4 -- "d2:"
done
