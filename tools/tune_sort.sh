#!/bin/bash
# tuning sweep of the two-pass sort parameters (group size, LDS tile, partition slice)
for lg in 20 22 18; do
for cfg in "4096 5120" "8192 10240" "16384 20480"; do
  set -- $cfg
  for sl in 1024 2048 4096; do
  CTT_SORT_GROUP=$1 CTT_SORT_CAP=$2 CTT_SORT_SLICE=$sl python bench.py --log2n $lg --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | \
   python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2^$lg group=$1 cap=$2 slice=$sl', 'sort', round(d['stage_ms']['sort'],3), 'acc', round(d['stage_ms']['accumulate'],3), 'total', round(d['ms_per_step'],3))"
  done
done
done
