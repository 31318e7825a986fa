#!/bin/bash
# kernel timeline of the pipelined loop of one cu_mask_sweep configuration:  [CURVE=<curve>] bash tools/cu_mask_timeline.sh <outdir> <log2n> <config...>
OUT=$PWD/$1; LOG2N=$2; shift 2
mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
for cfg in "$@"; do
  name=${cfg%%:*}
  rm -rf "$OUT/prof_$name"
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$OUT/prof_$name" -o p -- python "$REPO/tools/cu_mask_sweep.py" ${CURVE:-bls12_381_g1} $LOG2N 1 -- "$cfg" > "$OUT/prof_$name.json" 2> "$OUT/prof_$name.log" )
  DB=$(find "$OUT/prof_$name" -name "*.db" | head -1)
  python tools/timeline_window.py "$DB" 0.4 80 > "$OUT/timeline_${name}_2pow$LOG2N.txt" 2>> "$OUT/prof_$name.log"
  rm -rf "$OUT/prof_$name"
done
