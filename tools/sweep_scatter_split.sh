#!/bin/bash
# sort-stage time vs number of scatter sweeps at large N (tuning experiment; run through gpurun)
for cfg in "bls12_381_g1 22" "bn254_snarks_g1 22" "bls12_381_g1 24" "bls12_381_g1 21"; do
  set -- $cfg
  for sp in 1 2 4 8 16; do
    CTT_HIP_MSM_SCATTER_SPLIT=$sp python bench.py --curve $1 --log2n $2 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 2^$2 split=$sp', round(d['value']/1e6,1),'M/s sort', round(d['stage_ms']['sort'],3),'ms total', round(d['ms_per_step'],3))"
  done
done
