#!/bin/bash
# kernel timeline of one bench configuration: bash tools/prof_once.sh <tag> [bench args...]
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT" -o p -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/log.txt" )
DB=$(find "$OUT" -name "*.db" | head -1)
python tools/kernel_timeline.py "$DB" > "$OUT/timeline.txt" 2>> "$OUT/log.txt"
find "$OUT" -name "*.db" -delete
cut -c1-150 "$OUT/timeline.txt" | grep -v "k_pyr"
